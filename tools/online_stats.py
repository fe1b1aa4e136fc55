#!/usr/bin/env python3
"""The serial placement phase on the bench tree (GPU box): one sample after the other -- placement search, tree edit,
updatePartials, maple_tree_patch -- and what each step costs per sample.

    python tools/online_stats.py [tips in the starting tree] [samples to add]

The loop is bench.serial_phase (its tree edit is a STAND-IN for MAPLE's placeSampleOnTree, M:8300-8722 -- good for timing,
not for parity: tests/test_hip_search.py::test_online_sample_additions_through_tree_patch applies the reference's own
recorded edits).
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


def main():
    n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_add = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    data = make_dataset(n_samples=n_tips, l_ref=29903, seed=1, mean_diffs=30.0)
    ref_idx, rf = reference_tables(data.ref)
    dev = Device(ref_idx, rf, arena_bytes=max(4 << 30, n_tips * (96 << 10)))
    dev.set_model(bench.UNREST_Q)
    tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
    m = TreeMirror(dev, data.parent, data.blen, tips).build()
    l_ref = dev.lRef
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=ll, thresholdLogLKconsecutivePlacement=1.0)
    prng = np.random.default_rng(11)
    new = [tip_genome_list(perturb_diffs(data.diffs[i % n_tips], data.ref, prng), ref_idx) for i in range(n_add)]
    sp = bench.serial_phase(dev, m, new, pkw)
    t_upload, t_search, t_update, t_patch = (sp["times"][k] for k in ("upload", "search", "update", "patch"))
    placed, skipped, patched, cols = sp["placed"], sp["skipped"], sp["patched"], sp["cols"]
    n = cols["n"]
    up, c0, c1, dist, tip, mut = (cols[k] for k in ("up", "c0", "c1", "dist", "tip", "mut"))
    lower, up_right, up_left, tot_up = (cols[k] for k in ("lower", "up_right", "up_left", "tot_up"))

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
