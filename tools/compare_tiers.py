#!/usr/bin/env python3
"""The bench tree (branch lengths optimised), one deep round by the frontier tier and by the one-lane-per-search kernels; the
searches that differ, checked with the C oracle: compare_tiers.py [samples] [model] [truth|optimised]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "siteerr"
tree = sys.argv[3] if len(sys.argv) > 3 else "optimised"
data = make_dataset(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(4 << 30, samples * (64 << 10))))
mkw = bench.model_kwargs(model, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
tips = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips)
tip_ids = m.lower.copy()
mark = dev.mark()
m.build()
if tree == "optimised":
    print(bench.optimise_branch_lengths(dev, m, tip_ids, mark, 1.0 / (10 * dev.lRef)), flush=True)
kw = bench.search_kwargs(dev.lRef)
order = bench.preorder_nodes(m)
no_mut = -np.ones(m.n_nodes, dtype=np.int32)
dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, no_mut)
a = dev.spr_search_batch(order, **kw)
b = dev.spr_search_batch(order, search_tier=1, **kw)
bad = np.zeros(len(order), bool)
for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "blen"):
    d = a[k] != b[k]
    if d.ndim > 1: d = d.any(axis=1)
    print(k, "differs in", int(d.sum()), flush=True)
    bad |= d
idx = np.nonzero(bad)[0][:20]
if not len(idx) and len(sys.argv) > 4:                                   # no difference between the tiers: a sample against the oracle
    idx = np.arange(0, len(order), int(sys.argv[4]))
if os.environ.get("NODES"):
    want = [int(x) for x in os.environ["NODES"].split(",")]
    idx = np.asarray([int(np.nonzero(order == v)[0][0]) for v in want])
for i in idx:
    print(f"search {i} node {order[i]} dist {m.dist[order[i]]:.3g}: frontier status {a['status'][i]} best {a['bestNode'][i]} {a['bestScore'][i]:.6f} n {a['nAppend'][i]}"
          f" | lane status {b['status'][i]} best {b['bestNode'][i]} {b['bestScore'][i]:.6f} n {b['nAppend'][i]}")
if len(idx):
    from oracle.oracle_py import Oracle, OracleTree
    orc = Oracle(ref_idx, root_freqs); orc.set_model(**mkw)
    n = m.n_nodes
    lists4 = []
    for ids in (m.lower, m.up_right, m.up_left, m.tot_up):
        have = np.nonzero(ids >= 0)[0]
        have = have[np.argsort(ids[have], kind="stable")]
        lists4.append((have, dev.download_packed(ids[have])))
    up = [None if p < 0 else int(p) for p in m.parent]
    children = [[] if m.children[v, 0] < 0 else [int(m.children[v, 0]), int(m.children[v, 1])] for v in range(n)]
    ot = OracleTree(orc, m.root, up, children, m.dist, [[] for _ in range(n)], [0] * n, lists4)
    o = orc.spr_worker(ot, order[idx], **kw)
    for j, i in enumerate(idx):
        same = all(o[k][j] == a[k][i] for k in ("status", "bestNode", "placement", "nAppend"))
        if not same or len(idx) <= 20:
            print(f"oracle search {i} node {order[i]} dist {m.dist[order[i]]:.3g} tip {m.is_tip[order[i]]}: status {o['status'][j]} best {o['bestNode'][j]} {o['bestScore'][j]:.9f} "
                  f"placement {o['placement'][j]} n {o['nAppend'][j]} blen {o['blen'][j]} | gpu status {a['status'][i]} best {a['bestNode'][i]} {a['bestScore'][i]:.9f} "
                  f"placement {a['placement'][i]} n {a['nAppend'][i]} blen {a['blen'][i]}")
    print("oracle sample", len(idx), "done")
