#!/usr/bin/env python3
"""Placement searches (findBestParentForNewSample) on the bench tree: queries/s of maple_amd.search (GPU box)."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.search import PlacementParams, PlacementSearcher
from maple_amd.synth import make_dataset
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
up = [None if p < 0 else int(p) for p in m.parent]
children = [[] if m.children[v, 0] < 0 else [int(m.children[v, 0]), int(m.children[v, 1])] for v in range(nn)]
ht = HostTree(m.root, up, children, m.dist, [[] for _ in range(nn)], [0] * nn, None, None, None, None)
ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp = m.lower, m.up_right, m.up_left, m.tot_up
ht.id_mut = -np.ones(nn, dtype=np.int32)
l_ref = dev.lRef; ll = math.log(l_ref)
ps = PlacementSearcher(dev, ht, PlacementParams(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref),
                                                  thresholdLogLK=18.0 * ll, thresholdLogLKoptimization=ll,
                                                  thresholdLogLKconsecutivePlacement=1.0))
queries = [tips[int(v)] for v in data.tip_node[:nq]]
ps.find_best_parent_for_new_sample(queries[0])
t0 = time.perf_counter(); tot = 0; scored = 0
for q in queries:
    node, score, bl, diffs, info = ps.find_best_parent_for_new_sample(q)
    tot += info["n_append"]; scored += info.get("candidates_scored", 0)
dt = time.perf_counter() - t0
print(f"{nq} placement searches on a {n + nq}-tip tree: {dt / nq * 1e3:.2f} ms/query, {nq / dt:.0f} queries/s; "
      f"reference-equivalent placements {tot} ({tot / dt:.3g}/s), branches scored {scored} ({scored / dt:.3g}/s)")
