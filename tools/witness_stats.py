#!/usr/bin/env python3
"""How many (whole-tree search, candidate branch) pairs survive a one-witness filter?  witness_stats.py [samples] [model]

A search from a zero-length branch without an error model scores -inf on every branch whose probVectTotUp holds a nucleotide X
without a stored length at a site where the removed list holds another nucleotide without one (M:6663).  Give every candidate
ONE such entry (its rarest) as a witness; a query then only needs the candidates whose witness it is compatible with."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_mirror import TreeMirror

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = sys.argv[2] if len(sys.argv) > 2 else "ratevar"
data = make_dataset(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
ref_idx, root_freqs = reference_tables(data.ref)
dev = Device(ref_idx, root_freqs, arena_bytes=min(128 << 30, max(4 << 30, samples * (64 << 10))))
dev.set_model(**bench.model_kwargs(model, len(ref_idx)))
tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
L = dev.lRef
eff = 1.0 / (10 * L)
par = m.parent
isroot_child = np.zeros(m.n_nodes, bool)
isroot_child[par >= 0] = par[par[par >= 0]] < 0
cand = np.where((par >= 0) & ((m.dist > eff) | isroot_child) & (m.tot_up >= 0))[0]
qs = np.where((par >= 0) & (m.dist == 0.0) & (m.lower >= 0))[0]
print(f"{len(cand)} candidates, {len(qs)} zero-length nodes", flush=True)
pk = dev.download_packed(m.tot_up[cand])
owner = np.repeat(np.arange(len(cand)), np.diff(pk.ent_off))
typ = (pk.meta & 7).astype(np.int64); refb = ((pk.meta >> 3) & 3).astype(np.int64)
b5 = (pk.meta >> 5) & 1; b6 = (pk.meta >> 6) & 1
n_ent = len(owner)
pos = pk.pos[:n_ent].astype(np.int64); typ = typ[:n_ent]; refb = refb[:n_ent]; b5 = b5[:n_ent]; b6 = b6[:n_ent]
elig = (typ < 4) & (b5 == 0) & (b6 == 0) & (typ != refb)
key = pos * 4 + typ
nK = 4 * (L + 2)
hist = np.bincount(key[elig], minlength=nK)
print(f"candidate entries {n_ent}, eligible {elig.sum()}, distinct keys {np.count_nonzero(hist)}", flush=True)
# rarest eligible entry per candidate
e_idx = np.where(elig)[0]
order = np.lexsort((hist[key[e_idx]], owner[e_idx]))
e_sorted = e_idx[order]
first = np.ones(len(e_sorted), bool); first[1:] = owner[e_sorted][1:] != owner[e_sorted][:-1]
wit_ent = e_sorted[first]
wit_key = -np.ones(len(cand), np.int64); wit_key[owner[wit_ent]] = key[wit_ent]
none = int((wit_key < 0).sum())
bsize = np.bincount(wit_key[wit_key >= 0], minlength=nK)
cum = np.concatenate([[0], np.cumsum(bsize)])
print(f"candidates without a witness {none}; largest bucket {bsize.max()}, buckets used {np.count_nonzero(bsize)}", flush=True)
# queries
qk = dev.download_packed(m.lower[qs])
qo = np.repeat(np.arange(len(qs)), np.diff(qk.ent_off))
nq = len(qo)
qpos = qk.pos[:nq].astype(np.int64); qmeta = qk.meta[:nq]
qt = (qmeta & 7).astype(np.int64); q5 = (qmeta >> 5) & 1; q6 = (qmeta >> 6) & 1
start = np.ones(nq, np.int64); start[1:] = qpos[:-1] + 1
firstq = np.ones(nq, bool); firstq[1:] = qo[1:] != qo[:-1]
start[firstq] = 1
anyk = (qt == 5) | (qt == 6) | (q5 == 1) | (q6 == 1)                   # every key of the entry's sites is compatible
lo = np.where(anyk, start * 4, qpos * 4 + qt)
hi = np.where(anyk, qpos * 4 + 3, qpos * 4 + qt)
use = anyk | (qt < 4)
surv = np.zeros(len(qs), np.int64)
np.add.at(surv, qo[use], cum[hi[use] + 1] - cum[lo[use]])
surv += none
print(f"queries {len(qs)}: survivors per query mean {surv.mean():.0f}, median {np.median(surv):.0f}, p90 {np.percentile(surv, 90):.0f}, "
      f"p99 {np.percentile(surv, 99):.0f}, max {surv.max()}; total pairs {surv.sum():.3e} of {len(qs) * len(cand):.3e} "
      f"({surv.sum() / (len(qs) * len(cand)):.4f})", flush=True)
frac_any = np.bincount(qo[anyk], weights=(qpos - start + 1)[anyk], minlength=len(qs))
print(f"sites of a query where anything is compatible (N, O, entries with lengths): mean {frac_any.mean():.0f}, p99 {np.percentile(frac_any, 99):.0f}")
