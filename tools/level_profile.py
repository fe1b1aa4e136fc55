"""Per level of the frontier tier's expansion on the bench tree: items and kernel times.
level_profile.py [samples] [model] [v1|v2] [searches per round: evenly spread over the pre-order; default all]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list, tip_lists_packed
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset, make_dataset_native
from maple_amd.tree_mirror import TreeMirror

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
synth = sys.argv[3] if len(sys.argv) > 3 else "v1"
data = (make_dataset if synth == "v1" else make_dataset_native)(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(4 << 30, samples * (64 << 10))))
mkw = bench.model_kwargs(model, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
if synth == "v1":
    m = TreeMirror(dev, data.parent, data.blen, {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)})
else:
    c = data.diffs
    m = TreeMirror(dev, data.parent, data.blen, tip_packed=(data.tip_node, tip_lists_packed(c.off, c.code, c.pos, c.length, ref_idx, **tip_kw)))
tip_ids = m.lower.copy()
mark = dev.mark()
m.build()
bench.optimise_branch_lengths(dev, m, tip_ids, mark, 1.0 / (10 * dev.lRef))
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
if len(sys.argv) > 4:
    order = order[:: max(1, len(order) // int(sys.argv[4]))][: int(sys.argv[4])]
dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, -np.ones(m.n_nodes, dtype=np.int32))
for i in range(3):
    dev.timing_reset()
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order, **kw)
    print(f"round {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
iu, ic, mu, mc = dev.frontier_levels()
ws, wb = dev.last_wave_items
print("level  updating_items  (by wavefronts: small class, 512 class)  cached_items  ms_updating  ms_cached")
for l in range(len(iu)):
    if l > 60 and l % 10 and l < len(iu) - 3:
        continue
    print(f"{l:5d} {iu[l]:12d} {ws[l]:10d} {wb[l]:8d} {ic[l]:12d} {mu[l]:10.3f} {mc[l]:10.3f}")
print("total", iu.sum(), ic.sum(), round(float(mu.sum()), 1), round(float(mc.sum()), 1))
