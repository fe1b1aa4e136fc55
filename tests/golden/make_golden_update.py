#!/usr/bin/env python3
"""Golden vectors for updatePartials (M:5479-5815) -- BUILD CONTAINER ONLY.

Re-runs the unmodified reference exactly like make_golden_search.py (same flags, hence the same final tree: the
snapshot stored in search_<name>.json.gz is the BASE here and is not stored again), then, on a fresh deep copy of
the frozen tree per case, applies one local change and lets the reference's own ``updatePartials`` repair the genome
lists:

* kind "dist":  dist[v] is multiplied or divided by 3; nodeList = [(v,2,True,False),(up[v],childNum,True,False)]
  (the pair ``updateBLen(..., addToList=True)`` pushes, M:5412-5414);
* kind "tip":   the lower list of tip a is replaced by a copy of that of tip b (same MAT reference frame);
  same nodeList.

Per case the file keeps the change, the lists that differ from the base afterwards (node -> four lists), and the
tree log-likelihood after the repair.  Data only: tests/golden/update_<name>.json.gz.
"""
import contextlib
import copy
import gzip
import io
import json
import os
import random
import runpy
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, ser_list  # noqa: E402
from make_golden_search import RUNS, frozen_reference_tree, snapshot_tree  # noqa: E402

KEYS = ("probVect", "probVectUpRight", "probVectUpLeft", "probVectTotUp")


def run(name, n_cases=14):
    g, tree, t1, _, _ = frozen_reference_tree(name, RUNS[name])
    base = json.loads(json.dumps(snapshot_tree(tree, t1)))
    with gzip.open(os.path.join(HERE, f"search_{name}.json.gz"), "rt") as fh:
        stored = json.load(fh)["tree"]
    assert all(base[k] == stored[k] for k in KEYS + ("up", "children", "dist", "mutations")), "base tree differs from search fixture"

    n = len(tree.up)
    reach = []
    st = [t1]
    while st:
        v = st.pop()
        reach.append(v)
        st.extend(tree.children[v])
    frame = {}
    for v in reach:
        u = tree.up[v]
        frame[v] = v if tree.mutations[v] else (frame[u] if u is not None else -1)
    rng = random.Random(7)
    tips = [v for v in reach if not tree.children[v] and not tree.minorSequences[v]]
    inner = [v for v in reach if v != t1 and tree.dist[v] and tree.dist[v] > 1e-5]
    cases = []
    while len(cases) < n_cases:
        tc = copy.deepcopy(tree)
        kind = "dist" if len(cases) % 2 == 0 else "tip"
        if kind == "dist":
            v = rng.choice(inner)
            factor = rng.choice([3.0, 1.0 / 3.0])
            tc.dist[v] = tree.dist[v] * factor
            change = dict(kind=kind, node=v, dist=tc.dist[v])
        else:
            a = rng.choice(tips)
            same = [b for b in tips if b != a and frame[b] == frame[a] and tree.probVect[b] != tree.probVect[a]]
            if not same:
                continue
            b = rng.choice(same)
            tc.probVect[a] = copy.deepcopy(tree.probVect[b])
            v = a
            change = dict(kind=kind, node=a, source=b, probVect=ser_list(tc.probVect[a]))
        u = tc.up[v]
        cnum = 0 if tc.children[u][0] == v else 1
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                g["updatePartials"](tc, [(v, 2, True, False), (u, cnum, True, False)])
                lk = g["calculateTreeLikelihood"](tc, t1)
        except Exception as e:                      # the reference bails out on an inconsistent change: not a usable case
            print("skipped", change["kind"], v, repr(e)[:80])
            continue
        after = json.loads(json.dumps(snapshot_tree(tc, t1)))
        delta = {}
        for key in KEYS:
            for w in reach:
                if after[key][w] != base[key][w]:
                    delta.setdefault(str(w), {})[key] = after[key][w]
        dist_delta = {str(w): after["dist"][w] for w in reach if after["dist"][w] != base["dist"][w]}
        cases.append(dict(change=change, lists=delta, dist=dist_delta, treeLK=lk))
        print(f"[{name}] case {len(cases)} {change['kind']} node {v}: {len(delta)} nodes touched, "
              f"{len(dist_delta)} branch lengths changed, treeLK {lk:.6f}", flush=True)
    # ---- traverseTreeToOptimizeBranchLengths(fastPass=True), M:8727-8893: every dirty branch re-estimated from the
    # SAME (frozen) genome lists, no updatePartials in between -- first on the converged tree, then with every
    # branch length multiplied by a random factor in [0.4, 2.5] (lists left as they are)
    sweeps = []
    for perturbed in (False, True):
        tc = copy.deepcopy(tree)
        if perturbed:
            for v in reach:
                if v != t1 and tc.dist[v]:
                    tc.dist[v] = tc.dist[v] * (0.4 + 2.1 * rng.random())
        dist_in = [float(x or 0.0) for x in tc.dist]
        with contextlib.redirect_stdout(io.StringIO()):
            g["setAllDirty"](tc, t1)
            updates = g["traverseTreeToOptimizeBranchLengths"](tc, t1, fastPass=True)
        sweeps.append(dict(dist_in=dist_in, dist_out=[float(x or 0.0) for x in tc.dist], updates=updates,
                           dirty_out=[bool(x) for x in tc.dirty]))
        print(f"[{name}] fast branch-length pass (perturbed={perturbed}): {updates} updates", flush=True)
    # ---- the sweep the reference really runs: traverseTreeToOptimizeBranchLengths(tree, t1) with its default arguments
    # (fastPass=False: one updatePartials after every changed branch, M:8875-8878; call sites M:11059 ... 12256), on the
    # converged tree and on the tree with every length perturbed (lists recomputed for the perturbed lengths first)
    full_sweeps = []
    rng2 = random.Random(11)
    for perturbed in (False, True):
        tc = copy.deepcopy(tree)
        with contextlib.redirect_stdout(io.StringIO()):
            if perturbed:
                for v in reach:
                    if v != t1 and tc.dist[v]:
                        tc.dist[v] = tc.dist[v] * (0.4 + 2.1 * rng2.random())
                g["setAllDirty"](tc, t1)
                g["reCalculateAllGenomeLists"](tc, t1)
            g["setAllDirty"](tc, t1)
            lk_in = g["calculateTreeLikelihood"](tc, t1)
            dist_in = [float(x or 0.0) for x in tc.dist]
            order = []

            def prof3(frame, event, arg):
                if event == "call" and frame.f_code.co_name == "updatePartials":
                    order.append(frame.f_locals["nodeList"][0][0])
            sys.setprofile(prof3)
            try:
                updates = g["traverseTreeToOptimizeBranchLengths"](tc, t1)
            finally:
                sys.setprofile(None)
            lk_out = g["calculateTreeLikelihood"](tc, t1)
        full_sweeps.append(dict(dist_in=dist_in, dist_out=[float(x or 0.0) for x in tc.dist], updates=updates,
                                update_order=order, dirty_out=[bool(x) for x in tc.dirty], treeLK_in=lk_in, treeLK_out=lk_out))
        print(f"[{name}] full branch-length sweep (perturbed={perturbed}): {updates} updates, LK {lk_in:.6f} -> {lk_out:.6f}",
              flush=True)
    # ---- findBestRoot (M:7730-7905) on the frozen tree: the search part (the re-rooting itself is tree surgery).
    # Two parameter sets; locals are read when the function returns (bestNodes is only re-keyed when it re-roots).
    roots = []
    for strict, fails, thr in ((True, g["allowedFailsTopology"], g["thresholdLogLKtopology"]),
                               (False, 4, 14.0 * g["thresholdLogLKtopology"] / max(1e-300, g["thresholdLogLKtopology"]) * 1.0)):
        if not strict:
            thr = g["thresholdLogLKtopology"] * 2.0
        tc = copy.deepcopy(tree)
        got = {}

        def prof(frame, event, arg):
            if event == "return" and frame.f_code.co_name == "findBestRoot":
                loc = frame.f_locals
                got.update(bestNode=loc["bestNode"], bestLKdiff=loc["bestLKdiff"], visited=loc["nodesVisitedRoot"],
                           bestNodes={str(k): v for k, v in loc["bestNodes"].items()}, newRoot=arg)
        sys.setprofile(prof)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                g["findBestRoot"](tc, t1, strictTopologyStopRules=strict, allowedFailsTopology=fails, thresholdLogLKtopology=thr,
                                  aBayesPlusOn=False)
        finally:
            sys.setprofile(None)
        got.update(strict=strict, fails=fails, thr=thr, rerooted=got["bestNode"] != t1)
        roots.append(got)
        print(f"[{name}] findBestRoot strict={strict}: bestNode {got['bestNode']} (root {t1}), bestLKdiff {got['bestLKdiff']:.6f}, "
              f"{got['visited']} nodes visited, {len(got['bestNodes'])} within the threshold", flush=True)
    path = os.path.join(HERE, f"update_{name}.json.gz")
    with gzip.open(path, "wt") as fh:
        json.dump(dict(name=name, base=f"search_{name}.json.gz", cases=cases, blen_sweeps=sweeps, blen_full_sweeps=full_sweeps,
                       find_best_root=roots,
                       effectivelyNon0BLen=g["effectivelyNon0BLen"]), fh)
    print(f"[{name}] -> {path} {os.path.getsize(path)/1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["synth_unrest", "synth_siteerr"]):
        run(nm)
