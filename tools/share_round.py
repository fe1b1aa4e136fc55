#!/usr/bin/env python3
"""One rank's share of a deep round on the bench's own tree when N GPUs deal the nodes round-robin: share_round.py [samples] [model] [N...]
(env WAVE_ALL_BELOW: the tuning of that name)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.runtime import Device
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
worlds = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
bt = bench.build_bench_tree(samples, model)
dev, m = bt.dev, bt.mirror
if os.environ.get("WAVE_ALL_BELOW"):
    dev.set_tuning(wave_all_below=int(os.environ["WAVE_ALL_BELOW"]))
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
for w in worlds:
    mine = order[0::w]
    for rep in range(3):
        dev.timing_reset()
        t0 = time.perf_counter()
        r = dev.spr_search_batch(mine, **kw)
        wall = 1e3 * (time.perf_counter() - t0)
    ks = {k: round(dev.timing_read_kind(getattr(Device, "KIND_" + k))[1], 1) for k in ("SPR_SCORE", "SPR_SEARCH", "FR_UPDATING", "FR_CACHED", "FR_REPLAY", "FR_WIDE", "SPR_REPLAY")}
    print(f"world {w}: rank 0 searches {len(mine)}: {wall:.1f} ms, placements {int(r['nAppend'][r['status'] >= -1].sum()):.3e}, {ks}", flush=True)
