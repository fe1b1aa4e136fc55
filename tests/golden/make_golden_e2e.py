#!/usr/bin/env python3
"""End-to-end records for the metric's second half ("final tree log-LK") -- BUILD CONTAINER ONLY.

Two sequences of tree states, both produced by the unmodified reference on tests/golden/synth_small.maple.txt with
--noLocalRef (no MAT reference frames, so that a tip's genome list never changes and a tree state is fully described
by its topology, branch lengths and tip lists):

* ``spr_moves``: the reference's own SPR rounds.  Every call of traverseTreeForTopologyUpdate (M:9287-9486) that changes
  the tree is recorded: the pruned node, the tree's log-likelihood before (calculateTreeLikelihood, M:9721-9779), the
  predicted improvement the call returns, the tree after the move (up / children / dist) and its log-likelihood, both as
  the reference keeps it (incrementally repaired lists) and recomputed from scratch on a copy
  (reCalculateAllGenomeLists) -- the same comparison the reference's --debugging mode makes (M:9508-9565).
* ``online``: BASELINE configs[4] in small.  On the frozen final tree, with the model frozen, new samples are added one
  after the other exactly as the main loop does (M:11744-11752): findBestParentForNewSample -> placeSampleOnTree.  Per
  sample: the query list, the search result, the tree after the placement and its log-likelihood.

Data only: tests/golden/e2e_synth_unrest.json.gz.
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
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from make_golden import REF, ser_list  # noqa: E402
from make_golden_search import input_path, perturb  # noqa: E402

# no MAT reference frames: --noLocalRef only covers input trees (M:6219); new clades become references once they hold
# --maxNumDescendantsForMATClade branches (M:8543), so that is set out of reach as well
FLAGS = ["--model", "UNREST", "--noLocalRef", "--maxNumDescendantsForMATClade", "1000000"]
# variants: name -> (input of make_golden_search.input_path, flags, record the SPR moves?)
#   *_fullmodel_mat: BASELINE configs[4]'s model (per-site rates + per-site error rates) on a tree WITH MAT local references
#   (the reference's default); only the online additions are recorded (the model changes between SPR rounds there)
VARIANTS = {
    "synth_unrest": ("synth_unrest", FLAGS, True),
    "b1429_unrest": ("b1429_unrest", FLAGS, True),
    "synth_fullmodel_mat": ("synth_unrest", ["--model", "UNREST", "--rateVariation", "--estimateSiteSpecificErrorRate",
                                             "--maxNumDescendantsForMATClade", "40"], False),
}
MAX_MOVES = 40
SCRAMBLE = 30          # tips whose names are permuted in the starting tree of the SPR-move run
N_ONLINE = 30


def topo(tree, root):
    return dict(root=root, up=list(tree.up), children=[list(c) if c else [] for c in tree.children],
                dist=[float(x or 0.0) for x in tree.dist], nMinor=[len(m) for m in tree.minorSequences],
                mutations=[[list(m) for m in ml] if ml else [] for ml in tree.mutations])


def main(name="synth_unrest"):
    out_dir = tempfile.mkdtemp(prefix="maple_golden_e2e_")
    in_name, FLAGS, with_moves = VARIANTS[name]
    inp = input_path(in_name, out_dir)
    argv = ["MAPLE", "--input", inp, "--output", os.path.join(out_dir, "out"), "--overwrite"] + FLAGS
    st = {"g": None, "busy": False, "pending": None}
    moves = []
    tips0 = {}

    def root_of(tree, v):
        while tree.up[v] is not None:
            v = tree.up[v]
        return v

    def prof(frame, event, arg):
        co = frame.f_code
        if not co.co_filename.endswith("MAPLEv0.7.5.4.py"):
            return
        if st["g"] is None:
            st["g"] = frame.f_globals
        if st["busy"] or co.co_name != "traverseTreeForTopologyUpdate" or len(moves) >= MAX_MOVES or not with_moves:
            return
        g = st["g"]
        if event == "call":
            tree, node = frame.f_locals["tree"], frame.f_locals["node"]
            st["busy"] = True
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    r = root_of(tree, node)
                    lk = g["calculateTreeLikelihood"](tree, r)
                st["pending"] = dict(node=node, lk_before=lk, before=topo(tree, r))
            finally:
                st["busy"] = False
        elif event == "return" and st["pending"] is not None:
            p, st["pending"] = st["pending"], None
            if arg is None or not arg[1]:
                return
            tree = frame.f_locals["tree"]
            st["busy"] = True
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    r = root_of(tree, p["node"]) if arg[0] is None else arg[0]
                    r = root_of(tree, r)
                    lk_inc = g["calculateTreeLikelihood"](tree, r)
                    tc = copy.deepcopy(tree)
                    g["setAllDirty"](tc, r)
                    g["reCalculateAllGenomeLists"](tc, r)
                    lk_fresh = g["calculateTreeLikelihood"](tc, r)
                if not tips0:
                    for v in range(len(tree.up)):
                        if tree.probVect[v] is not None and not tree.children[v]:
                            tips0[v] = ser_list(tree.probVect[v])
                moves.append(dict(node=p["node"], improvement=arg[1], lk_before=p["lk_before"], lk_after_incremental=lk_inc,
                                  lk_after_fresh=lk_fresh, before=p["before"] if not moves else None, after=topo(tree, r),
                                  Q=[list(x) for x in g["mutMatrixGlobal"]]))
                print(f"move {len(moves)}: node {p['node']} predicted {arg[1]:.4f} realised {lk_inc - p['lk_before']:.4f} "
                      f"(fresh lists {lk_fresh:.6f})", file=sys.stderr, flush=True)
            finally:
                st["busy"] = False

    def reference_run(args, hook):
        old = sys.argv
        sys.argv = args
        sys.setprofile(hook)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                runpy.run_path(REF, run_name="__main__")
        except SystemExit:
            pass
        finally:
            sys.setprofile(None)
            sys.argv = old

    # Run 1 builds the reference's own tree.  A tree the reference built is close to a local optimum (its SPR rounds
    # then apply a handful of moves at most), so for a long sequence of APPLIED moves run 2 starts from that tree with
    # the names of SCRAMBLE tips permuted among themselves (--inputTree ... --largeUpdate: every node dirty,
    # M:3648 / 12247): the SPR rounds have to carry each of them back.
    reference_run(argv, None)
    import re
    newick = open(os.path.join(out_dir, "out_tree.tree")).read()
    tip_names = sorted(set(re.findall(r"[(,]([A-Za-z][^:,()]*):", newick)))
    rng0 = random.Random(99)
    chosen = rng0.sample(tip_names, min(SCRAMBLE, len(tip_names)))
    perm = chosen[1:] + chosen[:1]
    swap = dict(zip(chosen, perm))
    scrambled = re.sub(r"([(,])([A-Za-z][^:,()]*):", lambda m: m.group(1) + swap.get(m.group(2), m.group(2)) + ":", newick)
    tree_path = os.path.join(out_dir, "scrambled.tree")
    open(tree_path, "w").write(scrambled)
    argv2 = ["MAPLE", "--input", inp, "--output", os.path.join(out_dir, "out2"), "--overwrite", "--inputTree", tree_path,
             "--largeUpdate"] + FLAGS
    st["g"] = None
    reference_run(argv2, prof)
    g = st["g"]
    tree, t1 = g["tree"], g["t1"]
    if "--noLocalRef" in FLAGS:
        assert not any(tree.mutations[v] for v in range(len(tree.up))), "--noLocalRef run has MAT mutations"
    # every tip list of the final tree (tips never change without local references; minor-sequence bookkeeping aside)
    tips = {}
    for v in range(len(tree.up)):
        if tree.probVect[v] is not None and not tree.children[v]:
            tips[str(v)] = ser_list(tree.probVect[v])
    keys = ["lRef", "thresholdProb", "minBLenSensitivity", "thresholdDiffForUpdate", "thresholdFoldChangeUpdate",
            "oneMutBLen", "effectivelyNon0BLen", "thresholdLogLK", "thresholdLogLKoptimization",
            "thresholdLogLKoptimizationTopology", "thresholdLogLKtopology", "thresholdLogLKconsecutivePlacement",
            "allowedFails", "allowedFailsTopology", "defaultBLen", "strictStopRules", "thresholdTopologyPlacement"]
    ctx = {k: g[k] for k in keys}
    ctx["rootFreqs"] = list(g["rootFreqs"])
    ctx["ref"] = g["ref"]
    model = dict(useRateVariation=bool(g["useRateVariation"]), usingErrorRate=bool(g["usingErrorRate"]),
                 errorRateSiteSpecific=bool(g["errorRateSiteSpecific"]), Q=[list(r) for r in g["mutMatrixGlobal"]],
                 siteRates=list(g["siteRates"]) if g["useRateVariation"] else None,
                 errorRateGlobal=g["errorRateGlobal"] if g["usingErrorRate"] else 0.0,
                 errorRates=list(g["errorRates"]) if (g["usingErrorRate"] and g["errorRateSiteSpecific"]) else None)
    # (with an error model the reference's tips share one ambiguity vector object per IUPAC code and rewrite it in place,
    # M:3966: de-alias them before the lists are recomputed, as make_golden_search.py does)
    for v in range(len(tree.up)):
        if tree.probVect[v] is not None and not tree.children[v]:
            tree.probVect[v] = copy.deepcopy(tree.probVect[v])

    # ---- online additions on the frozen final tree (model frozen) ----
    with contextlib.redirect_stdout(io.StringIO()):
        g["setAllDirty"](tree, t1)
        g["reCalculateAllGenomeLists"](tree, t1)
        lk0 = g["calculateTreeLikelihood"](tree, t1)
    start = topo(tree, t1)
    tips = {}                                   # (the lists of the de-aliased tips, as the online phase starts)
    for v in range(len(tree.up)):
        if tree.probVect[v] is not None and not tree.children[v]:
            tips[str(v)] = ser_list(tree.probVect[v])
    from maple_amd.host import read_maple_file
    _, data = read_maple_file(inp)
    names = sorted(data)
    rng = random.Random(321)
    online = []
    names_in_tree = g["namesInTree"]
    for k in range(N_ONLINE):
        diffs = perturb(data[names[rng.randrange(len(names))]], g["ref"], rng)
        sample_name = f"online{k}"
        names_in_tree.append(sample_name)
        idx = len(names_in_tree) - 1
        n_before = len(tree.up)
        tips_before = {v: ser_list(tree.probVect[v]) for v in range(n_before) if tree.probVect[v] is not None and not tree.children[v]}
        muts_before = [[list(m) for m in ml] if ml else [] for ml in tree.mutations]
        with contextlib.redirect_stdout(io.StringIO()):
            q = g["probVectTerminalNode"](diffs, None, None)
            q_ser = ser_list(q)
            ret = g["findBestParentForNewSample"](tree, t1, q, idx, False)
            best_node, best_score, blens, passed = ret
            new_root = None
            if blens is not None:
                new_root = g["placeSampleOnTree"](tree, best_node, passed, idx, best_score, blens[0], blens[1], blens[2],
                                                  g["pseudoMutCounts"])
            if new_root is not None:
                t1 = new_root
            lk = g["calculateTreeLikelihood"](tree, t1)
        new_nodes = list(range(n_before, len(tree.up)))
        new_tips = {str(v): ser_list(tree.probVect[v]) for v in new_nodes if not tree.children[v]}
        # existing tips whose list the placement rewrote (a minor sequence with an error model, M:7951 / 3966)
        for v, before in tips_before.items():
            if not tree.children[v] and ser_list(tree.probVect[v]) != before:
                new_tips[str(v)] = ser_list(tree.probVect[v])
        # (a placement above a MAT reference node moves that node's mutation list to the new internal node, M:8300-8722:
        # `after` carries the mutation lists of every node)
        online.append(dict(diffs=[list(e) for e in diffs], query=q_ser,
                           ret=dict(bestNode=best_node, bestScore=best_score,
                                    bestBranchLengths=None if blens is None else [0.0 if b is False else b for b in blens]),
                           after=topo(tree, t1), new_nodes=new_nodes, new_tips=new_tips, treeLK=lk))
        print(f"online {k}: placed at {best_node} score {best_score:.4f} -> {len(new_nodes)} new nodes, LK {lk:.6f}", flush=True)
    out = dict(flags=FLAGS, context=ctx, model=model, tips=tips, spr_moves=moves, online_start=start, online_start_LK=lk0,
               online=online)
    path = os.path.join(HERE, f"e2e_{name}.json.gz")
    with gzip.open(path, "wt") as fh:
        json.dump(out, fh)
    print(f"-> {path} {os.path.getsize(path) / 1e6:.2f} MB; {len(moves)} SPR moves, {len(online)} online placements", flush=True)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["synth_unrest"]):
        main(nm)
