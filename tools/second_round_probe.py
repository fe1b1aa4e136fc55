"""The same batch of nodes searched several times on the bench's tree (the library remembers which searches ran over the whole-tree
budget: maple_ctx::h_over_hint): wall time per call and that every call returns the first call's answers.
second_round_probe.py [samples] [model] [searches] [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
model = sys.argv[2] if len(sys.argv) > 2 else "siteerr"
nsearch = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
bt = bench.build_bench_tree(samples, model, synth="v2")
dev = bt.dev
dev.set_tuning(verbose=int(os.environ.get("MAPLE_VERBOSE", "0")), no_over_hint=bool(os.environ.get("NOHINT")))
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(bt.mirror)[:nsearch]
first = None
for i in range(calls):
    t0 = time.perf_counter()
    r = dev.spr_search_batch(order, **kw)
    ms = 1e3 * (time.perf_counter() - t0)
    same = True
    if first is None:
        first = r
    else:
        same = all(np.array_equal(r[k], first[k]) for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"))
    print(f"call {i}: {ms:.1f} ms, {int((r['status'] == 0).sum())} searches done, nAppend {int(r['nAppend'].sum()):.4g}, identical to the first call: {same}", flush=True)
