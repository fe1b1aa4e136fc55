#!/usr/bin/env python3
"""Which SPR searches of a deep round are whole-tree ones?  (GPU box; experiments behind DESIGN.md section 3.)

usage: python tools/wide_stats.py [samples] [model]
Prints, for the bench tree, the distribution of nAppend over all searches and how the whole-tree searches correlate with
properties of the pruned node (tip / inner, clade size, length of its lower list, branch length).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from maple_amd.host import reference_tables, tip_genome_list  # noqa: E402
from maple_amd.runtime import Device  # noqa: E402
from maple_amd.synth import make_dataset  # noqa: E402
from maple_amd.tree_mirror import TreeMirror  # noqa: E402


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    model = sys.argv[2] if len(sys.argv) > 2 else "unrest"
    data = make_dataset(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
    ref_idx, root_freqs = reference_tables(data.ref)
    dev = Device(ref_idx, root_freqs, device=0, arena_bytes=max(4 << 30, samples * (640 << 10)))
    mkw = bench.model_kwargs(model, len(ref_idx))
    dev.set_model(**mkw)
    tip_lists = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
    mirror = TreeMirror(dev, data.parent, data.blen, tip_lists).build()
    no_mut = -np.ones(mirror.n_nodes, dtype=np.int32)
    dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist, mirror.is_tip,
                    mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up, no_mut)
    kw = bench.search_kwargs(dev.lRef, False)
    order = bench.preorder_nodes(mirror)
    res = dev.spr_search_batch(order, **kw)
    na = res["nAppend"].astype(np.int64)
    st = res["status"]
    ok = st == 0
    print("searches", len(order), "status counts", dict(zip(*np.unique(st, return_counts=True))))
    qs = [1, 5, 10, 25, 50, 75, 90, 95, 99]
    print("nAppend percentiles", dict(zip(qs, np.percentile(na[ok], qs).astype(int))))
    n_scored = int((mirror.tot_up >= 0).sum())
    wide = ok & (na > 0.5 * n_scored)
    print("scored branches", n_scored, "; whole-tree searches (> half of them):", int(wide.sum()))
    edges = [0, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 1 << 30]
    h, _ = np.histogram(na[ok & ~wide], bins=edges)
    print("nAppend histogram of the others:", list(zip(edges[:-1], h.tolist())))
    # properties of the pruned node
    n = mirror.n_nodes
    size = np.ones(n, dtype=np.int64)
    for v in order[::-1]:
        p = mirror.parent[v]
        if p >= 0:
            size[p] += size[v]
    n_ent, n_aux = dev.sizes(mirror.lower[order])
    tip = mirror.is_tip[order].astype(bool)
    dist = mirror.dist[order]
    for name, m in (("whole-tree", wide), ("others", ok & ~wide)):
        print(f"{name}: n {int(m.sum())}, tips {tip[m].mean():.3f}, clade size median {np.median(size[order][m]):.0f} p90 "
              f"{np.percentile(size[order][m], 90):.0f}, lower-list entries median {np.median(n_ent[m]):.0f} p10 "
              f"{np.percentile(n_ent[m], 10):.0f} p90 {np.percentile(n_ent[m], 90):.0f}, aux median {np.median(n_aux[m]):.0f}, "
              f"dist median {np.median(dist[m]):.3g} zero-dist {np.mean(dist[m] == 0):.3f}, currentLK median "
              f"{np.median(res['currentLK'][m]):.2f}")
    np.savez(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"wide_stats_{samples}.npz"), nAppend=na,
             status=st, size=size[order], n_ent=n_ent, n_aux=n_aux, tip=tip, dist=dist, currentLK=res["currentLK"],
             bestScore=res["bestScore"])


if __name__ == "__main__":
    main()
