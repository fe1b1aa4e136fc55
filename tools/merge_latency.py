#!/usr/bin/env python3
"""Latency of ONE wavefront of one-lane mergeVectors walks (k_merge, no spills) by list length: merge_latency.py"""
import sys, os, time
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
dev.set_tuning(wave_per_item_max=-1)                                   # one lane per pair
inner = np.nonzero(m.children[:, 0] >= 0)[0]
ne, _ = dev.sizes(m.lower)
for lo, hi in ((0, 60), (60, 100), (100, 160), (160, 400)):
    sel = [v for v in inner if lo <= ne[m.children[v, 0]] + ne[m.children[v, 1]] < hi][:64]
    if len(sel) < 64:
        continue
    l1 = m.lower[m.children[sel, 0]]; l2 = m.lower[m.children[sel, 1]]
    b1 = m.dist[m.children[sel, 0]]; b2 = m.dist[m.children[sel, 1]]
    steps = (ne[m.children[sel, 0]] + ne[m.children[sel, 1]])
    mark = dev.mark()
    ts = []
    for rep in range(30):
        mk2 = dev.mark()
        t0 = time.perf_counter()
        dev.merge_batch(l1, b1, m.is_tip[m.children[sel, 0]], l2, b2, m.is_tip[m.children[sel, 1]], False)
        ts.append(1e3 * (time.perf_counter() - t0))
        dev.release(mk2)
    # the same with 64 copies of the longest pair: no divergence between the lanes, every load a broadcast
    k = int(np.argmax(steps))
    ts2 = []
    for rep in range(30):
        mk2 = dev.mark()
        t0 = time.perf_counter()
        dev.merge_batch(np.repeat(l1[k], 64), np.repeat(b1[k], 64), np.repeat(m.is_tip[m.children[sel, 0]][k], 64), np.repeat(l2[k], 64),
                        np.repeat(b2[k], 64), np.repeat(m.is_tip[m.children[sel, 1]][k], 64), False)
        ts2.append(1e3 * (time.perf_counter() - t0))
        dev.release(mk2)
    ts1 = []
    for rep in range(30):
        mk2 = dev.mark()
        t0 = time.perf_counter()
        dev.merge_batch(l1[k:k + 1], b1[k:k + 1], m.is_tip[m.children[sel, 0]][k:k + 1], l2[k:k + 1], b2[k:k + 1], m.is_tip[m.children[sel, 1]][k:k + 1], False)
        ts1.append(1e3 * (time.perf_counter() - t0))
        dev.release(mk2)
    print(f"   64 copies of the longest pair: {np.median(ts2):.3f} ms; that pair alone (one lane): {np.median(ts1):.3f} ms")
    dev.release(mark)
    print(f"64 merges, entries of both lists {steps.min()}-{steps.max()} (mean {steps.mean():.0f}): wall median {np.median(ts):.3f} ms, min {min(ts):.3f} ms per call", flush=True)
