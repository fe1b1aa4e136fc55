#!/usr/bin/env python3
"""The serial placement loop with its samples announced (rows made ahead, traversals made ahead) against the plain loop, at scale:
tools/online_equal.py [samples=1000000] [n_add=2000] [ahead=100] [model=siteerr] -- every search's status, node, score, branch
lengths and candidate count, and the final trees, must be identical.  (The GPU test does this on a 6 000-tip tree.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, math
from maple_amd.host import tip_genome_list
from maple_amd.synth import perturb_diffs
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_add = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ahead = int(sys.argv[3]) if len(sys.argv) > 3 else 100
model = sys.argv[4] if len(sys.argv) > 4 else "siteerr"
bt = bench.build_bench_tree(samples, model, refs="none")
dev, m = bt.dev, bt.mirror
l_ref = dev.lRef
ll = math.log(l_ref)
pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
           thresholdLogLKoptimization=1.0 * ll, thresholdLogLKconsecutivePlacement=1.0)
prng = np.random.default_rng(33)
src = prng.choice(len(bt.data.diffs), size=n_add, replace=True)
new_lists = [tip_genome_list(bt.data.diffs[int(i)] if k % 40 == 0 else perturb_diffs(bt.data.diffs[int(i)], bt.data.ref, prng), bt.ref_idx, **bt.tip_kw)
             for k, i in enumerate(src)]
runs = []
for a in (0, ahead):
    mark = dev.mark()
    t0 = time.time()
    s0 = dev.placement_ahead_stats()
    sp = bench.serial_phase(dev, m, new_lists, pkw, ahead=a)
    s1 = dev.placement_ahead_stats()
    print(f"ahead={a}: {time.time() - t0:.1f} s, placed {sp['placed']}, skipped {sp['skipped']}, stats {({k: s1[k] - s0[k] for k in s1})}", flush=True)
    runs.append(sp)
    dev.release(mark)
a, b = runs
assert a["results"] == b["results"], [i for i, (x, y) in enumerate(zip(a["results"], b["results"])) if x != y][:5]
for name in ("up", "c0", "c1", "dist", "tip"):
    assert np.array_equal(a["cols"][name][: a["cols"]["n"]], b["cols"][name][: b["cols"]["n"]]), name
print("identical:", len(a["results"]), "searches,", a["cols"]["n"], "nodes")
dev.close()
