#!/usr/bin/env python3
"""BASELINE configs[4] on its own: tools/online_probe.py [samples=1000000] [n_add=4096] [ahead=512] [model=siteerr]
-- bench.online_update_leg on the bench tree (new samples added one after the other, then a round); prints its block."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_add = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ahead = int(sys.argv[3]) if len(sys.argv) > 3 else 512
model = sys.argv[4] if len(sys.argv) > 4 else "siteerr"
t0 = time.time()
bt = bench.build_bench_tree(samples, model, refs="none")
print(f"tree built in {time.time() - t0:.0f} s", flush=True)
if os.environ.get("MAPLE_VERBOSE"):
    bt.dev.set_tuning(verbose=int(os.environ["MAPLE_VERBOSE"]))
kw = bench.search_kwargs(bt.dev.lRef)
out = bench.online_update_leg(bt.dev, bt.mirror, bt.data, bt.ref_idx, bt.tip_kw, kw, n_add, min(131072, bt.mirror.n_nodes), ahead=ahead)
print(json.dumps(out, indent=1))
bt.dev.close()
