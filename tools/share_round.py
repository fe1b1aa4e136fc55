#!/usr/bin/env python3
"""One rank's share of a deep round when N GPUs deal the nodes round-robin: share_round.py [samples] [model] [N...]"""
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
worlds = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
data = make_dataset(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(4 << 30, samples * (64 << 10))))
mkw = bench.model_kwargs(model, len(ref_idx)); dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
tips = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
no_mut = -np.ones(m.n_nodes, dtype=np.int32)
dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, no_mut)
for w in worlds:
    mine = order[0::w]
    for rep in range(3):
        dev.timing_reset()
        t0 = time.perf_counter()
        r = dev.spr_search_batch(mine, **kw)
        wall = 1e3 * (time.perf_counter() - t0)
    ks = {k: round(dev.timing_read_kind(getattr(Device, "KIND_" + k))[1], 1) for k in ("SPR_SCORE", "SPR_SEARCH", "FR_UPDATING", "FR_CACHED", "FR_REPLAY", "FR_WIDE", "SPR_REPLAY")}
    print(f"world {w}: rank 0 searches {len(mine)}: {wall:.1f} ms, placements {int(r['nAppend'][r['status'] >= -1].sum()):.3e}, {ks}", flush=True)
