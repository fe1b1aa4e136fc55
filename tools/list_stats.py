#!/usr/bin/env python3
"""Print the size distribution of the bench workload's genome lists (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
data = make_dataset(n_samples=n, l_ref=29903, seed=1, mean_diffs=30.0)
ref_idx, rf = reference_tables(data.ref)
dev = Device(ref_idx, rf, arena_bytes=4 << 30)
dev.set_model(bench.UNREST_Q)
tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
cand = m.candidate_nodes(1.0 / (10 * dev.lRef))
for name, ids in (("totUp[cand]", m.tot_up[cand]), ("lower[tips]", m.lower[data.tip_node])):
    ne, na = dev.sizes(ids)
    print(name, "n", len(ids), "entries mean %.1f p50 %d p90 %d p99 %d max %d | aux mean %.1f p99 %d" % (
        ne.mean(), np.percentile(ne, 50), np.percentile(ne, 90), np.percentile(ne, 99), ne.max(), na.mean(), np.percentile(na, 99)))
    w = ne[: len(ne) // 64 * 64].reshape(-1, 64)
    print("   per-wave max/mean of entries: %.2f" % (w.max(axis=1).mean() / ne.mean()))
lst = dev.download(m.tot_up[cand][:3])
from collections import Counter
print(Counter(e[0] for l in lst for e in l), Counter(len(e) for l in lst for e in l))
print(lst[0][:12])
