#!/usr/bin/env python3
"""Placement searches (findBestParentForNewSample) on the bench tree: queries/s of maple_amd.search (GPU box)."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.search import PlacementParams, PlacementSearcher
from maple_amd.synth import make_dataset, perturb_diffs
from maple_amd.tree_host import HostTree
from maple_amd.tree_mirror import TreeMirror
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200
data = make_dataset(n_samples=n + nq, l_ref=29903, seed=1, mean_diffs=30.0)
ref_idx, rf = reference_tables(data.ref)
dev = Device(ref_idx, rf, arena_bytes=4 << 30)
dev.set_model(bench.UNREST_Q)
tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
nn = m.n_nodes
ht = HostTree.from_mirror(m, dev)
l_ref = dev.lRef; ll = math.log(l_ref)
ps = PlacementSearcher(dev, ht, PlacementParams(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref),
                                                  thresholdLogLK=18.0 * ll, thresholdLogLKoptimization=ll,
                                                  thresholdLogLKconsecutivePlacement=1.0))
prng = np.random.default_rng(11)
queries = [tip_genome_list(perturb_diffs(dl, data.ref, prng), ref_idx) for dl in data.diffs[:nq]]   # not in the tree
ps.find_best_parent_host_replay(queries[0])
t0 = time.perf_counter(); tot = 0; scored = 0
for q in queries:
    node, score, bl, diffs, info = ps.find_best_parent_host_replay(q)
    tot += info["n_append"]; scored += info.get("candidates_scored", 0)
dt = time.perf_counter() - t0
print(f"Python replay: {nq} placement searches on a {n + nq}-tip tree: {dt / nq * 1e3:.2f} ms/query, {nq / dt:.0f} queries/s; "
      f"reference-equivalent placements {tot} ({tot / dt:.3g}/s), branches scored {scored} ({scored / dt:.3g}/s)")
# the same queries as ONE native batch (maple_placement_search_batch): device-side traversal, one lane per query
for reps in (1, 8):
    qs = queries * reps
    ps.find_best_parent_batch(qs[:4])
    t0 = time.perf_counter()
    res = ps.find_best_parent_batch(qs)
    dt = time.perf_counter() - t0
    tot_b = sum(r[4]["n_append"] for r in res)
    print(f"batched: {len(qs)} queries in {dt * 1e3:.1f} ms = {len(qs) / dt:.0f} queries/s "
          f"(reference-equivalent placements {tot_b / dt:.3g}/s, branches scored {len(qs) * (len(ps.cand) + 1) / dt:.3g}/s)")
    if reps == 1:
        single = [ps.find_best_parent_host_replay(q) for q in queries[:50]]
        assert all(a[0] == b[0] and a[1] == b[1] and a[4]["n_append"] == b[4]["n_append"] for a, b in zip(single, res[:50]))
        print("batched == single-query results on the first 50 queries")
# one query per native call (the sequential placement loop's shape)
ps.find_best_parent_batch(queries[:1])
t0 = time.perf_counter()
for q in queries[:100]:
    ps.find_best_parent_batch([q])
dt = time.perf_counter() - t0
print(f"native call per query: {dt / 100 * 1e3:.2f} ms/query")
