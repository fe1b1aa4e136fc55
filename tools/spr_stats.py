#!/usr/bin/env python3
"""SPR search round on the bench tree: distribution of per-query work and timing of subsets (GPU box)."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
strict = len(sys.argv) > 2 and sys.argv[2] == "fast"
MODEL = os.environ.get("MODEL", "unrest")
data = make_dataset(n_samples=n, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(MODEL != "unrest"))
ref_idx, rf = reference_tables(data.ref)
dev = Device(ref_idx, rf, arena_bytes=(4 << 30) if n <= 20000 else (24 << 30))
mkw = bench.model_kwargs(MODEL, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if MODEL == "siteerr" else {}
tips = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
t0 = time.time(); m = TreeMirror(dev, data.parent, data.blen, tips).build(); print("mirror build s", time.time() - t0, dev.stats())
l_ref = dev.lRef; ll = math.log(l_ref)
dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, -np.ones(m.n_nodes, dtype=np.int32))
if strict:
    kw = dict(strict=True, allowedFails=2, thresholdLogLKtopology=6.0 * ll, thresholdTopologyPlacement=-0.1, thresholdLogLKoptimizationTopology=ll, thresholdLogLKconsecutivePlacement=1.0, effectivelyNon0BLen=1.0 / (10 * l_ref))
else:
    kw = dict(strict=False, allowedFails=4, thresholdLogLKtopology=14.0 * ll, thresholdTopologyPlacement=-0.1, thresholdLogLKoptimizationTopology=ll, thresholdLogLKconsecutivePlacement=1.0, effectivelyNon0BLen=1.0 / (10 * l_ref))
nodes = np.arange(m.n_nodes)
dev.spr_search_batch(nodes[:64], **kw)
BUDGET = int(os.environ.get("WIDE_BUDGET", "0"))
def run(sel, label):
    dev.timing_reset(); t0 = time.perf_counter(); r = dev.spr_search_batch(sel, wide_search_budget=BUDGET, **kw); w = time.perf_counter() - t0
    nl, ms = dev.timing_read()
    print("   per launch ms:", [round(x, 1) for x in dev.timing_read_each()])
    na = r["nAppend"]
    print(f"{label}: queries {len(sel)} placements {na.sum()} kernel_ms {ms:.1f} wall_ms {1e3*w:.1f} -> {na.sum()/(ms*1e-3):.3g}/s | nAppend mean {na.mean():.0f} p50 {np.percentile(na,50):.0f} p99 {np.percentile(na,99):.0f} max {na.max()} | status<0: {(r['status']<0).sum()} launches {nl}")
    return r
r = run(nodes, "all")
order = np.argsort(-r["nAppend"])
run(nodes[order[:64]], "64 longest")
run(nodes[order[:1]], "the longest")
run(nodes[order[len(order)//2:len(order)//2+2048]], "2048 median")
