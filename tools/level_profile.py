"""Per level of the frontier tier's expansion on the bench's own tree (bench.build_bench_tree): items and kernel times.
level_profile.py [samples] [model] [searches per round: evenly spread over the pre-order; default all] [refs: local|none]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
bt = bench.build_bench_tree(samples, model, refs=sys.argv[4] if len(sys.argv) > 4 else "local", debug_library=True)
dev, m = bt.dev, bt.mirror
kw = bench.search_kwargs(dev.lRef)
if os.environ.get("MAPLE_VERBOSE"):
    dev.set_tuning(verbose=int(os.environ["MAPLE_VERBOSE"]), wave_all_below=int(os.environ.get("WAVE_ALL_BELOW", "0")))
elif os.environ.get("WAVE_ALL_BELOW"):
    dev.set_tuning(wave_all_below=int(os.environ["WAVE_ALL_BELOW"]))
order = bench.preorder_nodes(m)
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    order = order[:: max(1, len(order) // int(sys.argv[3]))][: int(sys.argv[3])]
for i in range(3):
    dev.timing_reset()
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order, **kw)
    print(f"round {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
if os.environ.get("REUPLOAD"):                                   # (what bench.py reports as first_step_after_upload_ms)
    for i in range(2):
        t0 = time.perf_counter()
        bt.upload_headline_tree()
        t1 = time.perf_counter()
        r = dev.spr_search_batch(order, **kw)
        print(f"upload {1e3 * (t1 - t0):.1f} ms, round after it: {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order, **kw)
    print(f"round after that: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
iu, ic, mu, mc = dev.frontier_levels()
ws, wb = dev.last_wave_items
print("level  updating_items  (by wavefronts: small class, 512 class)  cached_items  ms_updating  ms_cached")
for l in range(len(iu)):
    if l > 60 and l % 10 and l < len(iu) - 3:
        continue
    print(f"{l:5d} {iu[l]:12d} {ws[l]:10d} {wb[l]:8d} {ic[l]:12d} {mu[l]:10.3f} {mc[l]:10.3f}")
print("total", iu.sum(), ic.sum(), round(float(mu.sum()), 1), round(float(mc.sum()), 1))
K = {name: dev.timing_read_kind(getattr(type(dev), "KIND_" + name)) for name in ("SPR_SCORE", "SPR_SEARCH", "SPR_REPLAY", "FR_UPDATING", "FR_CACHED", "FR_REPLAY", "FR_WIDE")}
for k, v in K.items():
    print(k, "launches %d  ms %.2f  units %.4g  bytes %.4g" % v)
