#!/usr/bin/env python3
"""The serial placement phase on the bench tree (GPU box): one sample after the other -- placement search, tree edit,
updatePartials, maple_tree_patch -- and what each step costs per sample.

    python tools/online_stats.py [tips in the starting tree] [samples to add]

The tree edit here is a STAND-IN for MAPLE's placeSampleOnTree (M:8300-8722), which stays host code of the reference: a new
internal node on the branch above the best node, with the three branch lengths the search returned (a sample the search
calls a minor sequence of a tip, or a placement at the root, is skipped).  It is good for timing -- the edits have the shape
and the locality of the reference's -- not for parity: tests/test_hip_search.py::test_online_sample_additions_through_tree_patch
applies the reference's own recorded edits.  The tree lives in plain numpy columns with room to grow; nothing in the loop
touches all nodes (which nodes updatePartials touched comes from maple_update_partials_touched).
"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from maple_amd.host import reference_tables, tip_genome_list  # noqa: E402
from maple_amd.runtime import Device  # noqa: E402
from maple_amd.synth import make_dataset, perturb_diffs  # noqa: E402
from maple_amd.tree_mirror import TreeMirror  # noqa: E402

DEPTH_STEP = 1 << 12        # depths are kept in units of 1/4096 of a level: a node put on a branch gets a depth in between


def main():
    n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_add = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    data = make_dataset(n_samples=n_tips, l_ref=29903, seed=1, mean_diffs=30.0)
    ref_idx, rf = reference_tables(data.ref)
    dev = Device(ref_idx, rf, arena_bytes=max(4 << 30, n_tips * (96 << 10)))
    dev.set_model(bench.UNREST_Q)
    tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
    m = TreeMirror(dev, data.parent, data.blen, tips).build()
    n0, cap = m.n_nodes, m.n_nodes + 2 * n_add

    def grown(a, fill, dtype):
        out = np.full(cap, fill, dtype=dtype)
        out[:n0] = a
        return out
    up = grown(m.parent, -1, np.int32)
    c0, c1 = grown(m.children[:, 0], -1, np.int32), grown(m.children[:, 1], -1, np.int32)
    tip = grown(m.is_tip, 0, np.uint8)
    dist = grown(m.dist, 0.0, np.float64)
    mut = np.full(cap, -1, dtype=np.int32)
    lower, up_right = grown(m.lower, -1, np.int32), grown(m.up_right, -1, np.int32)
    up_left, tot_up = grown(m.up_left, -1, np.int32), grown(m.tot_up, -1, np.int32)
    depth = np.zeros(cap, dtype=np.int32)                # (maple_update_partials only compares depths)
    for v in bench.preorder_nodes(m):
        if up[v] >= 0:
            depth[v] = depth[up[v]] + DEPTH_STEP
    n = n0
    dev.upload_tree(m.root, up[:n], c0[:n], c1[:n], dist[:n], tip[:n], lower[:n], up_right[:n], up_left[:n], tot_up[:n], mut[:n])
    l_ref = dev.lRef
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=ll, thresholdLogLKconsecutivePlacement=1.0)
    prng = np.random.default_rng(11)
    new = [tip_genome_list(perturb_diffs(data.diffs[i % n_tips], data.ref, prng), ref_idx) for i in range(n_add)]
    t_search, t_update, t_patch, t_upload, placed, skipped, patched = [], [], [], [], 0, 0, []
    for k, lst in enumerate(new):
        dev.placement_prepare(**pkw)
        t0 = time.perf_counter()
        qid = int(dev.upload([lst])[0])                  # the sample's list stays: it becomes the new tip's lower list
        t_upload.append(time.perf_counter() - t0)
        mark = dev.mark()
        t0 = time.perf_counter()
        out = dev.placement_search_batch(np.asarray([qid], dtype=np.int32), **pkw)
        t_search.append(time.perf_counter() - t0)
        dev.release(mark)
        b = int(out["bestNode"][0])
        if out["status"][0] != 0 or up[b] < 0:
            skipped += 1
            continue
        top, bottom, app = (float(x) for x in out["blen"][0])
        g, p, s = int(up[b]), n, n + 1
        # ---- the stand-in tree edit: p on the branch above b, the sample s as p's other child
        if c0[g] == b:
            c0[g] = p
        else:
            c1[g] = p
        up[p], c0[p], c1[p], dist[p], tip[p] = g, b, s, top, 0
        up[b], dist[b] = p, bottom
        up[s], dist[s], tip[s], lower[s] = p, app, 1, qid
        depth[p] = (depth[g] + depth[b]) // 2
        depth[s] = depth[p] + 1
        if not (depth[g] < depth[p] < depth[b]):
            raise SystemExit("ran out of depth resolution on one branch (raise DEPTH_STEP)")
        n += 2
        # ---- updatePartials around the new nodes, inside the library, on these very columns
        t0 = time.perf_counter()
        dev.update_partials(m.root, up[:n], c0[:n], c1[:n], tip[:n], mut[:n], depth[:n], dist[:n], lower[:n], up_right[:n],
                            up_left[:n], tot_up[:n], [b, s, p])
        t_update.append(time.perf_counter() - t0)
        # ---- the library's copy of the tree: only the nodes that changed
        t0 = time.perf_counter()
        touched = np.unique(np.concatenate([dev.update_partials_touched(), [g, b, p, s]])).astype(np.int32)
        dev.tree_patch(n, touched, up[touched], c0[touched], c1[touched], dist[touched], tip[touched], lower[touched],
                       up_right[touched], up_left[touched], tot_up[touched])
        t_patch.append(time.perf_counter() - t0)
        patched.append(len(touched))
        placed += 1

    def med(x):
        return 1e3 * float(np.median(x[len(x) // 10:])) if len(x) else float("nan")
    total = med(t_upload) + med(t_search) + med(t_update) + med(t_patch)
    print(f"{n_tips}-tip tree, {n_add} samples one after the other: {placed} placed, {skipped} skipped (minor sequence / at the root)")
    print(f"  per sample (median, ms): upload of its list {med(t_upload):.2f}, placement search {med(t_search):.2f}, "
          f"updatePartials {med(t_update):.2f}, tree patch {med(t_patch):.2f} "
          f"-> {total:.2f} ms = {1e3 / total:.0f} samples/s; nodes patched per sample: median {int(np.median(patched))}, max {max(patched)}")
    # for comparison: what a full re-upload of the tree costs at this size
    t0 = time.perf_counter()
    dev.upload_tree(m.root, up[:n], c0[:n], c1[:n], dist[:n], tip[:n], lower[:n], up_right[:n], up_left[:n], tot_up[:n], mut[:n])
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev.placement_prepare(**pkw)
    print(f"  a full maple_tree_upload of the {n}-node tree: {1e3 * t_full:.1f} ms, + {1e3 * (time.perf_counter() - t0):.1f} ms for the "
          f"placement tables")
    # the final tree is consistent: a batch of searches on it (tables rebuilt from the patched copy) agrees with single queries
    mark = dev.mark()
    qs = dev.upload(new[:8])
    one = [int(dev.placement_search_batch(np.asarray([q], dtype=np.int32), **pkw)["bestNode"][0]) for q in qs]
    many = dev.placement_search_batch(np.concatenate([qs, qs]), **pkw)["bestNode"][:8].tolist()
    print(f"  single-query and batched searches on the final tree agree: {one == many}")
    dev.release(mark)


if __name__ == "__main__":
    main()
