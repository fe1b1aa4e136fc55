import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
bt = bench.build_bench_tree(100000, "ratevar", refs="none")
dev, m = bt.dev, bt.mirror
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
zero = [int(v) for v in order if m.dist[v] == 0.0 and m.parent[v] >= 0][:40]
r = dev.spr_search_batch(order, **kw)     # warm: pools
cols = (m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up)
touched = np.asarray(order[:8], dtype=np.int32)
for k in range(4):
    dev.tree_patch(m.n_nodes, touched, *[np.asarray(c)[touched] for c in cols])
    if k == 3: dev.set_tuning(verbose=1)
    t0 = time.perf_counter()
    r = dev.spr_search_batch(np.asarray([zero[k]], dtype=np.int32), wide_search_budget=0, **kw)
    print(f"whole-tree re-search after a patch: {1e3*(time.perf_counter()-t0):.1f} ms, status {r['status'][0]} nAppend {r['nAppend'][0]}", flush=True)
dev.set_tuning(verbose=0)
