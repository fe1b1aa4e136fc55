#!/usr/bin/env python3
"""update_genome_lists (level-synchronous updatePartials) and rebuild_genome_lists on the bench tree (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_host import HostTree, rebuild_genome_lists, update_genome_lists
from maple_amd.tree_mirror import TreeMirror
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
data = make_dataset(n_samples=n, l_ref=29903, seed=1, mean_diffs=30.0)
ref_idx, rf = reference_tables(data.ref)
dev = Device(ref_idx, rf, arena_bytes=6 << 30)
dev.set_model(bench.UNREST_Q)
tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
nn = m.n_nodes
ht = HostTree.from_mirror(m, dev)
t0 = time.perf_counter()
lo, ur, ul, tu = rebuild_genome_lists(dev, ht)
dt = time.perf_counter() - t0
print(f"full rebuild of all 4 lists of {nn} nodes: {dt * 1e3:.1f} ms")
rng = np.random.default_rng(3)
cand = np.nonzero((m.parent >= 0) & (m.dist > 1e-5))[0]
for k in (1, 10, 100, 1000):
    pick = rng.choice(cand, size=k, replace=False)
    for v in pick:
        ht.dist[v] = ht.dist[v] * 3.0
    mark = dev.mark()
    t0 = time.perf_counter()
    rep = update_genome_lists(dev, ht, pick.tolist())
    dt = time.perf_counter() - t0
    print(f"{k} simultaneous branch-length changes: {rep} lists replaced in {dt * 1e3:.2f} ms ({dt / k * 1e3:.3f} ms per change)")
    if k == 1:                                          # the same kind of change again, both level loops, warm
        for native in (True, False):
            ts = []
            for v in rng.choice(cand, size=20, replace=False):
                ht.dist[v] = ht.dist[v] * 1.5
                t0 = time.perf_counter()
                update_genome_lists(dev, ht, [int(v)], native=native)
                ts.append(time.perf_counter() - t0)
            print(f"  single change, {'library' if native else 'Python'} level loop: median {np.median(ts) * 1e3:.2f} ms, "
                  f"min {min(ts) * 1e3:.2f} ms over 20 changes")
# findBestRoot's search + the fast branch-length pass on the same tree
import math
from maple_amd.tree_host import find_best_root, optimize_branch_lengths_fast_pass
ht2 = HostTree.from_mirror(m)
ll = math.log(dev.lRef)
t0 = time.perf_counter()
node, best, best_nodes, visited = find_best_root(dev, ht2, strictTopologyStopRules=True, allowedFailsTopology=2,
                                                 thresholdLogLKtopology=6.0 * ll, thresholdLogLKoptimizationTopology=ll,
                                                 thresholdLogLKconsecutivePlacement=1.0)
dt = time.perf_counter() - t0
print(f"findBestRoot search: all {nn - 1} branches scored in {dt * 1e3:.1f} ms; best node {node} (root {ht2.root}), "
      f"score {best:.4f}, {visited} nodes visited by the reference's traversal, {len(best_nodes)} kept")
t0 = time.perf_counter()
upd, _ = optimize_branch_lengths_fast_pass(dev, ht2, 1.0 / (10 * dev.lRef))
print(f"fast branch-length pass: {upd} of {nn - 1} lengths replaced in {(time.perf_counter() - t0) * 1e3:.1f} ms")
# consistency at scale, with MAT local references: the repaired lists against a full rebuild of the changed tree
from maple_amd.mat import add_local_references
from maple_amd.tree_host import tree_log_likelihood
ht3 = HostTree.from_mirror(m)
print("reference nodes", add_local_references(dev, ht3, 50))
pick = rng.choice(cand, size=200, replace=False)
for v in pick:
    ht3.dist[v] = ht3.dist[v] * 2.5
rep = update_genome_lists(dev, ht3, pick.tolist())
lk_upd, _ = tree_log_likelihood(dev, ht3)
lo, ur, ul, tu = rebuild_genome_lists(dev, ht3)
bad = 0
for a, b in ((ht3.id_lower, lo), (ht3.id_upRight, ur), (ht3.id_upLeft, ul), (ht3.id_totUp, tu)):
    ok = (a >= 0) & (b >= 0)
    bad += int((dev.differ_batch(a[ok], b[ok]) | dev.differ_batch(b[ok], a[ok])).sum()) + int(((a >= 0) != (b >= 0)).sum())
ht3.id_lower, ht3.id_upRight, ht3.id_upLeft, ht3.id_totUp = lo, ur, ul, tu
lk_reb, _ = tree_log_likelihood(dev, ht3)
print(f"200 changes on the tree with local references: {rep} lists replaced; lists that differ from a full rebuild by the "
      f"reference's own thresholds: {bad}; tree log-LK {lk_upd:.6f} vs {lk_reb:.6f} (rel {abs(lk_upd - lk_reb) / abs(lk_reb):.1e})")
