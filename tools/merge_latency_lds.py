#!/usr/bin/env python3
"""One lane per mergeVectors, input lists walked in global memory (mode 0) or staged in LDS first (mode 1): HIP-event time of a
launch of 64 different pairs (one wavefront) and of 16 384 pairs (one wavefront per compute unit), by list length.
tools/merge_latency_lds.py  (GPU box)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror
data = make_dataset(n_samples=20000, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=True)
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=4 << 30)
dev.set_model(**bench.model_kwargs("ratevar", len(ref_idx)))
tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
inner = np.nonzero(m.children[:, 0] >= 0)[0]
ne, na = dev.sizes(m.lower)
print("lower lists: entries mean %.1f p50 %d p90 %d p99 %d | aux mean %.1f p90 %d p99 %d" % (
    ne.mean(), np.percentile(ne, 50), np.percentile(ne, 90), np.percentile(ne, 99), na.mean(), np.percentile(na, 90), np.percentile(na, 99)))
neu, nau = dev.sizes(m.tot_up[m.tot_up >= 0])
print("totUp lists: entries mean %.1f p50 %d p90 %d p99 %d | aux mean %.1f p90 %d p99 %d" % (
    neu.mean(), np.percentile(neu, 50), np.percentile(neu, 90), np.percentile(neu, 99), nau.mean(), np.percentile(nau, 90), np.percentile(nau, 99)))
c0, c1 = m.children[inner, 0], m.children[inner, 1]
tot = ne[c0] + ne[c1]
tota = na[c0] + na[c1]
for lo, hi in ((0, 60), (60, 100), (100, 128), (128, 192)):
    ok = (tot >= lo) & (tot < hi) & (tota <= 96)
    sel = inner[ok][:64]
    if len(sel) < 64:
        print("bucket", lo, hi, "only", len(sel)); continue
    a, b = m.children[sel, 0], m.children[sel, 1]
    steps = ne[a] + ne[b]
    for reps_of_64 in (1, 256):
        l1 = np.tile(m.lower[a], reps_of_64); l2 = np.tile(m.lower[b], reps_of_64)
        b1 = np.tile(m.dist[a], reps_of_64); b2 = np.tile(m.dist[b], reps_of_64)
        t1 = np.tile(m.is_tip[a], reps_of_64); t2 = np.tile(m.is_tip[b], reps_of_64)
        res = []
        for mode, sw, sa in ((0, 0, 0), (1, 192, 96), (3, 192, 96), (2, 192, 96)):
            if mode and (steps.max() > sw or (na[a] + na[b]).max() > sa):
                res.append("(slab %d/%d: does not fit)" % (sw, sa)); continue
            grid = 0
            ms, out = dev.debug_merge_lds(l1, b1, t1, l2, b2, t2, 0, mode, sw, sa, reps=10, grid=grid)
            res.append("mode %d slab %d/%d: %.4f ms (%.3f us per step of the longest; checksum %d)" % (
                mode, sw, sa, ms, 1e3 * ms / steps.max() / (1 if reps_of_64 <= 256 else max(1, reps_of_64 // max(1, grid if grid else reps_of_64))), int(out.astype(np.int64).sum())))
        print(f"{64 * reps_of_64} pairs, entries of both lists {steps.min()}-{steps.max()} (mean {steps.mean():.0f}):\n   " + "\n   ".join(res), flush=True)
