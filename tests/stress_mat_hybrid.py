#!/usr/bin/env python3
"""Stress check at scale (GPU box, run by hand: python tests/stress_mat_hybrid.py N; not collected by pytest): on a tree with MAT local references the hybrid (batch-scored, replayed) deep round
must equal the lane-only search query by query."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root (bench, maple_amd, oracle)
import bench
from maple_amd.host import reference_tables, tip_genome_list
from maple_amd.mat import add_local_references
from maple_amd.runtime import Device
from maple_amd.synth import make_dataset
from maple_amd.tree_host import HostTree
from maple_amd.tree_mirror import TreeMirror
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
MODEL = os.environ.get("MODEL", "unrest")
SEED = int(os.environ.get("SEED", "1"))
data = make_dataset(n_samples=n, l_ref=29903, seed=SEED, mean_diffs=30.0, rate_variation=(MODEL != "unrest"), frac_with_n=0.05, frac_ambig=0.05)
ref_idx, rf = reference_tables(data.ref)
dev = Device(ref_idx, rf, arena_bytes=max(4 << 30, n * (640 << 10)))
mkw = bench.model_kwargs(MODEL, len(ref_idx))
dev.set_model(**mkw)
tip_kw = dict(error_rates=mkw["errorRates"]) if MODEL == "siteerr" else {}
tips = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
m = TreeMirror(dev, data.parent, data.blen, tips).build()
ht = HostTree.from_mirror(m)
print("reference nodes", add_local_references(dev, ht, 50))
dev.upload_tree(ht.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, ht.id_lower, ht.id_upRight,
                ht.id_upLeft, ht.id_totUp, ht.id_mut)
ll = math.log(dev.lRef)
kw = dict(strict=False, allowedFails=4, thresholdLogLKtopology=14.0 * ll, thresholdTopologyPlacement=-0.1,
          thresholdLogLKoptimizationTopology=ll, thresholdLogLKconsecutivePlacement=1.0, effectivelyNon0BLen=1.0 / (10 * dev.lRef))
nodes = np.arange(ht.n)
t0 = time.time(); hyb = dev.spr_search_batch(nodes, want_removed_partials=True, **kw); t1 = time.time()
pla = dev.spr_search_batch(nodes, wide_search_budget=-1, want_removed_partials=True, **kw); t2 = time.time()
print(f"hybrid {t1 - t0:.1f} s, lane-only {t2 - t1:.1f} s, placements {pla['nAppend'].sum()}")
for k in ("status", "bestNode", "placement", "nAppend"):
    print(k, "equal:", np.array_equal(hyb[k], pla[k]), "mismatches:", int((hyb[k] != pla[k]).sum()))
for k in ("bestScore", "improvement", "currentLK"):
    print(k, "max abs diff:", float(np.abs(hyb[k] - pla[k]).max()))
ok = (hyb["status"] == 0)
la = dev.download(hyb["removedPartials"][ok]); lb = dev.download(pla["removedPartials"][ok])
print("bestRemovedPartials lists equal:", sum(1 for a, b in zip(la, lb) if a == b), "of", len(la))
# batched placement (device traversal, frames) against the host replay on new samples
from maple_amd.search import PlacementParams, PlacementSearcher
from maple_amd.synth import perturb_diffs
ps = PlacementSearcher(dev, ht, PlacementParams(oneMutBLen=1.0 / dev.lRef, effectivelyNon0BLen=1.0 / (10 * dev.lRef), thresholdLogLK=18.0 * ll,
                                                  thresholdLogLKoptimization=ll, thresholdLogLKconsecutivePlacement=1.0,
                                                  onlyFindIdentical=(MODEL == "siteerr")))
prng = np.random.default_rng(5)
qs = [tip_genome_list(perturb_diffs(dl, data.ref, prng, n_extra=k % 3), ref_idx, **tip_kw) for k, dl in enumerate(data.diffs[:64])]
batch = ps.find_best_parent_batch(qs)
same = sum(1 for q, g in zip(qs, batch) if (lambda w: g[0] == w[0] and g[1] == w[1] and g[3] == w[3] and g[4]["n_append"] == w[4]["n_append"])(ps.find_best_parent_host_replay(q)))
print("batched placement == host replay on the tree with local references:", same, "of", len(qs))
# the C oracle's search (pinned to the reference's records) on a sample of the nodes of the same tree
if os.environ.get("ORACLE"):
    from oracle.oracle_py import Oracle, OracleTree
    orc = Oracle(ref_idx, rf)
    orc.set_model(**mkw)
    lists4 = [dev.download(ids) for ids in (ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp)]
    otree = OracleTree(orc, ht.root, ht.up, ht.children, np.asarray([float(x or 0.0) for x in ht.dist]), ht.mutations, [0] * ht.n, lists4)
    sel = nodes[::int(os.environ["ORACLE"])]
    o = orc.spr_worker(otree, sel, **kw)
    for k in ("status", "bestNode", "placement", "nAppend"):
        print("oracle", k, "equal:", np.array_equal(hyb[k][sel], o[k]), "mismatches:", int((hyb[k][sel] != o[k]).sum()))
    fin = np.isfinite(o["bestScore"]) & np.isfinite(hyb["bestScore"][sel])
    print("oracle bestScore max rel diff:", float((np.abs(hyb["bestScore"][sel][fin] - o["bestScore"][fin]) / np.maximum(1.0, np.abs(o["bestScore"][fin]))).max()))
