#!/usr/bin/env python3
"""The bench tree with MAT local references (as real MAPLE trees have): list sizes and search timings (GPU box)."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.mat import add_local_references
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_host import HostTree, tree_log_likelihood
from maple_amd.tree_mirror import TreeMirror
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
data = make_dataset(n_samples=n, l_ref=29903, seed=1, mean_diffs=30.0)
ref_idx, rf = reference_tables(data.ref)
dev = Device(ref_idx, rf, arena_bytes=8 << 30)
dev.set_model(bench.UNREST_Q)
tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
ht = HostTree.from_mirror(m)
lk0, _ = tree_log_likelihood(dev, ht)
ne0, na0 = dev.sizes(ht.id_totUp[ht.id_totUp >= 0])
t0 = time.perf_counter()
nref = add_local_references(dev, ht, 50)
print(f"{nref} reference nodes added in {time.perf_counter() - t0:.2f} s")
lk1, _ = tree_log_likelihood(dev, ht)
ne1, na1 = dev.sizes(ht.id_totUp[ht.id_totUp >= 0])
print(f"tree log-LK without / with local references: {lk0:.6f} / {lk1:.6f} (rel diff {abs(lk0 - lk1) / abs(lk0):.2e})")
print(f"probVectTotUp entries mean {ne0.mean():.1f} -> {ne1.mean():.1f}; aux doubles mean {na0.mean():.1f} -> {na1.mean():.1f}")
up = np.asarray([-1 if u is None else u for u in ht.up], dtype=np.int32)
c0 = np.asarray([c[0] if c else -1 for c in ht.children], dtype=np.int32)
c1 = np.asarray([c[1] if c else -1 for c in ht.children], dtype=np.int32)
is_tip = np.asarray([not c for c in ht.children], dtype=np.uint8)
dist = np.asarray([float(x or 0.0) for x in ht.dist])
dev.upload_tree(ht.root, up, c0, c1, dist, is_tip, ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp, ht.id_mut)
l_ref = dev.lRef; ll = math.log(l_ref)
for label, kw in (("fast round", dict(strict=True, allowedFails=2, thresholdLogLKtopology=6.0 * ll)),
                  ("deep round", dict(strict=False, allowedFails=4, thresholdLogLKtopology=14.0 * ll))):
    kw.update(thresholdTopologyPlacement=-0.1, thresholdLogLKoptimizationTopology=ll, thresholdLogLKconsecutivePlacement=1.0,
              effectivelyNon0BLen=1.0 / (10 * l_ref))
    nodes = np.arange(ht.n)
    dev.spr_search_batch(nodes, **kw)
    dev.timing_reset(); t0 = time.perf_counter(); r = dev.spr_search_batch(nodes, **kw); w = time.perf_counter() - t0
    nl, ms = dev.timing_read()
    print("   per launch ms:", [round(x, 1) for x in dev.timing_read_each()])
    print(f"{label} with local references: {r['nAppend'].sum()} placements, kernel {ms:.1f} ms, wall {1e3 * w:.1f} ms, "
          f"{r['nAppend'].sum() / (ms * 1e-3):.3g}/s, status<0: {(r['status'] < 0).sum()}, launches {nl}")
