"""One deep SPR search round on the bench tree, a few times (for rocprofv3): spr_round.py [samples] [model] [rounds] [tier]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tier = int(sys.argv[4]) if len(sys.argv) > 4 else 0
data = make_dataset(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(4 << 30, samples * (64 << 10))))
mkw = bench.model_kwargs(model, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
tip_lists = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tip_lists).build()
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
no_mut = -np.ones(m.n_nodes, dtype=np.int32)
dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, no_mut)
for i in range(rounds + 1):
    if i == 1:
        dev.timing_reset()
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order, **kw, search_tier=tier, wide_search_budget=int(os.environ.get("WIDE_BUDGET", "0")))
    wall = time.perf_counter() - t0
    print(f"round {i}: {1e3 * wall:.1f} ms, placements {int(r['nAppend'][r['status'] >= -1].sum()):.4e}, moves {(r['placement'] >= 0).sum()}", flush=True)
names = ("SPR_SCORE", "SPR_SEARCH", "SPR_REPLAY", "FR_UPDATING", "FR_CACHED", "FR_REPLAY", "FR_WIDE")
ks = {k: dev.timing_read_kind(getattr(Device, "KIND_" + k)) for k in names}
print("kernel ms per round:", {k: round(ks[k][1] / rounds, 1) for k in names}, flush=True)
print("launches per round:", {k: round(ks[k][0] / rounds, 1) for k in names}, flush=True)
