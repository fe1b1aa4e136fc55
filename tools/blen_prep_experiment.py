"""Experiment (round 3): Jacobi-style branch-length passes (the reference's fastPass form, M:8727-8893, every branch from
the same frozen lists) + full rebuild on the bench tree, and what one deep SPR round looks like after each pass."""
import sys, os, time, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 5
data = make_dataset(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(4 << 30, samples * (64 << 10))))
mkw = bench.model_kwargs(model, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
tip_lists = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tip_lists)
tip_ids = m.lower.copy()
mark0 = dev.mark()
t0 = time.time(); m.build(); print(f"build {time.time()-t0:.2f}s", flush=True)
l_ref = dev.lRef
kw = bench.search_kwargs(l_ref)
order = bench.preorder_nodes(m)
no_mut = -np.ones(m.n_nodes, dtype=np.int32)
nodes = np.nonzero(m.parent >= 0)[0]
nodes = nodes[m.parent[nodes] != m.root]

def spr_round(tag):
    dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, no_mut)
    dev.spr_search_batch(order, **kw)
    dev.timing_reset()
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order, **kw)
    wall = time.perf_counter() - t0
    ks = {k: dev.timing_read_kind(k) for k in (Device.KIND_SPR_SCORE, Device.KIND_SPR_SEARCH, Device.KIND_SPR_REPLAY)}
    st = dict(zip(*[x.tolist() for x in np.unique(r["status"], return_counts=True)]))
    na = r["nAppend"][r["status"] == 0]
    pc = np.percentile(na, [50, 75, 90, 99, 99.9, 100]).astype(int).tolist() if len(na) else []
    print(f"[{tag}] round {1e3*wall:.0f} ms, placements {int(r['nAppend'][r['status']>=-1].sum()):.3e}, status {st}, moves {(r['placement']>=0).sum()}, "
          f"nAppend pct {pc}, n>4264: {(na>4264).sum()}, kernels ms score/lane/replay: "
          f"{ks[Device.KIND_SPR_SCORE][1]:.0f}/{ks[Device.KIND_SPR_SEARCH][1]:.0f}/{ks[Device.KIND_SPR_REPLAY][1]:.0f} "
          f"launches {ks[Device.KIND_SPR_SCORE][0]}/{ks[Device.KIND_SPR_SEARCH][0]}/{ks[Device.KIND_SPR_REPLAY][0]}", flush=True)
    return r

spr_round("truth tree")
for it in range(passes):
    p = m.parent[nodes]
    first = m.children[p, 0] == nodes
    upv = np.where(first, m.up_right[p], m.up_left[p]).astype(np.int32)
    t0 = time.perf_counter()
    t, isf = dev.blen_batch(upv, m.lower[nodes], m.is_tip[nodes])
    t_bl = time.perf_counter() - t0
    best = np.where(isf.astype(bool), 0.0, t)
    d = m.dist[nodes]
    both0 = (best == 0) & (d == 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = d / best
    upd = ~both0 & ((best == 0) | (d == 0) | (ratio > 1.01) | (ratio < 0.99))
    cur = dev.append_batch(upv, m.lower[nodes], m.is_tip[nodes], d)
    print(f"pass {it}: blen_batch {1e3*t_bl:.1f} ms for {len(nodes)} branches; updates {upd.sum()}; zero-length now {(d==0).sum()} "
          f"-> {(best==0).sum()}; currentLK -inf: {np.isinf(cur).sum()}; sum dist {d.sum():.6f} -> {np.where(upd,best,d).sum():.6f}", flush=True)
    m.dist[nodes[upd]] = best[upd]
    dev.release(mark0)
    m.lower = tip_ids.copy(); m.up_right[:] = -1; m.up_left[:] = -1; m.tot_up[:] = -1
    t0 = time.time(); m.build(); print(f"  rebuild {time.time()-t0:.2f}s", flush=True)
    if it in (0, 1, 2, passes - 1):
        spr_round(f"after pass {it}")
