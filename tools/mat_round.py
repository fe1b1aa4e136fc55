#!/usr/bin/env python3
"""Deep SPR rounds on the bench tree WITH MAT local references (as real MAPLE trees have): mat_round.py [samples] [model] [rounds] [tier]
MAPLE_DEBUG=1 prints the tier's own accounting (searches handed back, levels, pool sizes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_lists_packed
from maple_amd.mat import add_local_references
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset_native
from maple_amd.tree_host import HostTree
from maple_amd.tree_mirror import TreeMirror

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tier = int(sys.argv[4]) if len(sys.argv) > 4 else 0
data = make_dataset_native(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(8 << 30, samples * (320 << 10))))
mkw = bench.model_kwargs(model, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
c = data.diffs
m = TreeMirror(dev, data.parent, data.blen, tip_packed=(data.tip_node, tip_lists_packed(c.off, c.code, c.pos, c.length, ref_idx, **tip_kw))).build()
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
no_mut = -np.ones(m.n_nodes, dtype=np.int32)
names = ("SPR_SCORE", "SPR_SEARCH", "SPR_REPLAY", "FR_UPDATING", "FR_CACHED", "FR_REPLAY", "FR_WIDE")


def rounds_on(label, verbose_round=1):
    for i in range(rounds + 1):
        if i == 1:
            dev.timing_reset()
        if i == verbose_round and os.environ.get("MAPLE_DEBUG"):
            dev.set_tuning(verbose=1)
        t0 = time.perf_counter()
        r = dev.spr_search_batch(order, **kw, search_tier=tier)
        wall = time.perf_counter() - t0
        dev.set_tuning(verbose=0)
        print(f"{label} round {i}: {1e3 * wall:.1f} ms, placements {int(r['nAppend'][r['status'] >= -1].sum()):.4e}, moves {(r['placement'] >= 0).sum()}, "
              f"failed {(r['status'] < -1).sum()}", flush=True)
    ks = {k: dev.timing_read_kind(getattr(Device, "KIND_" + k)) for k in names}
    print(f"{label} kernel ms per round:", {k: round(ks[k][1] / rounds, 1) for k in names}, flush=True)
    return r


dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, no_mut)
plain = rounds_on("plain tree")
ht = HostTree.from_mirror(m)
t0 = time.perf_counter()
n_ref = add_local_references(dev, ht, 50)
print(f"{n_ref} reference nodes added in {time.perf_counter() - t0:.1f} s", flush=True)
dist = np.asarray([float(x or 0.0) for x in ht.dist])
dev.upload_tree(ht.root, m.parent, m.children[:, 0], m.children[:, 1], dist, m.is_tip, ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp, ht.id_mut)
mat = rounds_on("with local references")
same = {k: int((plain[k] != mat[k]).sum()) for k in ("status", "bestNode", "placement", "nAppend")}
print("searches that differ between the two forms of the tree (node ids, moves, candidate counts):", same)
