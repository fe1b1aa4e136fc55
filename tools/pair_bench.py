#!/usr/bin/env python3
"""The list walk of the frontier tier's kernels on its own: appendProbNode over millions of UNRELATED (candidate list, removed list)
pairs, one lane per pair (k_append) -- the pairs a round's cached-regime items hold: every node's lower list against the
probVectTotUp of branches near it in the tree -- and the dense kernel (k_append_queries_lds: 512 queries x every branch).
Prints ns per pair and a digest of the results, so that two builds of the library (MAPLE_HIP_LIB) can be compared for speed
and for bit-identity.   tools/pair_bench.py [samples] [model]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
bt = bench.build_bench_tree(samples, model, refs="none")
dev, m = bt.dev, bt.mirror
order = bench.preorder_nodes(m)
offs = np.asarray([-60, -41, -29, -17, -11, -7, -4, -2, -1, 1, 2, 3, 5, 8, 13, 19, 27, 38, 52, 70])
idx = np.arange(100, len(order) - 100)
qi = np.repeat(idx, len(offs))
ci = qi + np.tile(offs, len(idx))
q, cnd = order[qi], order[ci]
ok = (m.tot_up[cnd] >= 0) & (m.lower[q] >= 0)
q, cnd = q[ok], cnd[ok]
cu = torch.device("cuda", 0)
t_p = torch.from_numpy(m.tot_up[cnd].astype(np.int32)).to(cu)
t_c = torch.from_numpy(m.lower[q].astype(np.int32)).to(cu)
t_tip = torch.from_numpy(m.is_tip[q].astype(np.uint8)).to(cu)
t_bl = torch.from_numpy(np.ascontiguousarray(m.dist[q], dtype=np.float64)).to(cu)
out = torch.empty(len(q), dtype=torch.float64, device=cu)
st = torch.cuda.current_stream().cuda_stream
def run_pairs():
    dev.append_batch_dev(len(q), t_p.data_ptr(), t_c.data_ptr(), t_tip.data_ptr(), t_bl.data_ptr(), out.data_ptr(), st)
run_pairs(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run_pairs(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
res = out.cpu().numpy()
print(f"k_append: {len(q)} pairs, {1e6 * min(ts) / len(q):.2f} ns per pair (best of 5: {min(ts):.2f} ms; all {[round(t, 2) for t in ts]}), "
      f"-inf {int(np.isinf(res).sum())}, digest {hashlib.sha1(res.tobytes()).hexdigest()[:16]}")
# the dense kernel: 512 queries (every 1/512th node's lower list) against every scored branch
cand = m.candidate_nodes(1.0 / (10 * dev.lRef))
qn = order[:: max(1, len(order) // 512)][:512]
t_q = torch.from_numpy(m.lower[qn].astype(np.int32)).to(cu)
t_k = torch.from_numpy(m.tot_up[cand].astype(np.int32)).to(cu)
out2 = torch.empty(len(qn) * len(cand), dtype=torch.float64, device=cu)
def run_dense():
    dev.append_queries_dev(len(qn), t_q.data_ptr(), len(cand), t_k.data_ptr(), True, 1.0 / dev.lRef, out2.data_ptr(), st)
run_dense(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run_dense(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
res2 = out2.cpu().numpy()
print(f"dense: {len(qn)} x {len(cand)} pairs, {1e6 * min(ts) / len(res2):.3f} ns per pair (best of 3: {min(ts):.2f} ms), "
      f"-inf {int(np.isinf(res2).sum())}, digest {hashlib.sha1(res2.tobytes()).hexdigest()[:16]}")
dev.close()
