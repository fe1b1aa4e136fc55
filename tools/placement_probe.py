"""One sample at a time through maple_placement_search_batch on the bench's tree (plain form), with the library's own account of
each call (MAPLE_VERBOSE=2 style prints on stderr): placement_probe.py [samples] [model] [queries]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import tip_genome_list
from maple_amd.synth import perturb_diffs

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 12
bt = bench.build_bench_tree(samples, model, refs="none", tree="truth")
dev = bt.dev
l_ref = dev.lRef
ll = math.log(l_ref)
pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
           thresholdLogLKoptimization=1.0 * ll, thresholdLogLKconsecutivePlacement=1.0)
prng = np.random.default_rng(11)
new = [tip_genome_list(perturb_diffs(bt.data.diffs[i], bt.data.ref, prng), bt.ref_idx, **bt.tip_kw) for i in range(nq)]
ids = dev.upload(new)
dev.placement_prepare(**pkw)
dev.placement_search_batch(ids[:1], **pkw)
ts = []
for k, q in enumerate(ids):
    if k == nq - 3:
        dev.set_tuning(verbose=2)
    mark = dev.mark()
    t0 = time.perf_counter()
    out = dev.placement_search_batch(np.asarray([q], dtype=np.int32), **pkw)
    ts.append(1e3 * (time.perf_counter() - t0))
    dev.release(mark)
    if k >= nq - 3:
        print(f"query {k}: {ts[-1]:.3f} ms, nAppend {int(out['nAppend'][0])}, status {int(out['status'][0])}", flush=True)
print("median ms per single-query placement search:", float(np.median(ts)), " candidates:", bt.mirror.n_nodes)
