#!/usr/bin/env python3
"""Golden vectors for the SEARCH rows of SURVEY.md section 8a (a11-a13) -- BUILD CONTAINER ONLY.

Runs the unmodified reference to completion on tests/golden/synth_small.maple.txt, then -- on the
frozen final tree, with the model the run ended with -- calls the reference's own
``startTopologyUpdatesParallel`` (M:9580-9716; it calls ``findBestParentTopology`` M:6817 for every
node) and ``findBestParentForNewSample`` (M:7912) and records inputs and outputs, plus ONE snapshot
of the tree (topology, branch lengths, MAT mutations and the four genome lists of every node).

Data only is written: tests/golden/search_<name>.json.gz.
"""
import contextlib
import copy
import gzip
import io
import json
import os
import random
import re
import runpy
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from make_golden import REF, ser_list  # noqa: E402

RUNS = {
    # no SPR rounds in the reference run itself, so the frozen tree still has improvable placements
    "synth_unrest": ["--model", "UNREST", "--maxNumDescendantsForMATClade", "12", "--numTopologyImprovements", "0",
                     "--noFastTopologyInitialSearch"],
    "synth_siteerr": ["--model", "UNREST", "--rateVariation", "--estimateSiteSpecificErrorRate",
                      "--maxNumDescendantsForMATClade", "12"],
    # the reference's own example alignment (lRef 29 903, 112 samples): SURVEY.md section 8c
    "example_unrest": ["--model", "UNREST", "--numTopologyImprovements", "0", "--noFastTopologyInitialSearch"],
    "example_siteerr": ["--model", "UNREST", "--rateVariation", "--estimateSiteSpecificErrorRate"],
    # BASELINE configs[0]: --model JC; the reference raises after SPR round 1 (EM is not implemented for JC, M:10877-10879)
    # with the round-1 tree in place, log-LK -44775.7276008509
    "example_jc": ["--model", "JC"],
    # real SARS-CoV-2 data: the first 600 samples of example_files/sameRef_B.1.429.maple.gz
    "b1429_unrest": ["--model", "UNREST", "--numTopologyImprovements", "0", "--noFastTopologyInitialSearch"],
    # Trees with MANY improvable placements (the accept rule and the four vetoes of M:9681-9700 need proposed moves to
    # bite on; a tree the reference built itself proposes 0-3): the reference builds its tree, the names of SCRAMBLE tips
    # are permuted among themselves in its Newick output, and the reference reads that tree back (--inputTree
    # --largeUpdate, M:3648 / 12247) without running any SPR round -- the tree frozen here is the one its first parallel
    # SPR round would search.
    "synth_scrambled": ["--model", "UNREST", "--maxNumDescendantsForMATClade", "12", "--numTopologyImprovements", "0",
                        "--noFastTopologyInitialSearch"],
    "b1429_scrambled": ["--model", "UNREST", "--numTopologyImprovements", "0", "--noFastTopologyInitialSearch"],
}
SCRAMBLE = {"synth_scrambled": 30, "b1429_scrambled": 40}
EXAMPLE_DIR = "/root/reference/example_files"
INPUTS = {                                   # default: tests/golden/synth_small.maple.txt
    "example_unrest": os.path.join(EXAMPLE_DIR, "MAPLE_alignment_example.txt"),
    "example_siteerr": os.path.join(EXAMPLE_DIR, "MAPLE_alignment_example.txt"),
    "example_jc": os.path.join(EXAMPLE_DIR, "MAPLE_alignment_example.txt"),
    "b1429_unrest": ("prefix", os.path.join(EXAMPLE_DIR, "sameRef_B.1.429.maple.gz"), 600),
    "b1429_scrambled": ("prefix", os.path.join(EXAMPLE_DIR, "sameRef_B.1.429.maple.gz"), 600),
}


def input_path(name, tmp_dir):
    """The alignment a run reads: a committed / reference file, or the first N samples of one (written to tmp_dir)."""
    spec = INPUTS.get(name, os.path.join(HERE, "synth_small.maple.txt"))
    if isinstance(spec, str):
        return spec
    _, src, n_keep = spec
    op = gzip.open if src.endswith(".gz") else open
    out = os.path.join(tmp_dir, f"{name}_input.maple.txt")
    n_rec = 0
    with op(src, "rt") as fi, open(out, "w") as fo:
        for line in fi:
            if line.startswith(">"):
                n_rec += 1
                if n_rec > n_keep + 1:          # record 1 is the reference
                    break
            fo.write(line)
    return out


def snapshot_tree(tree, root):
    n = len(tree.up)
    return dict(
        root=root, up=list(tree.up), children=[list(c) if c else [] for c in tree.children], dist=list(tree.dist),
        mutations=[[list(m) for m in ml] if ml else [] for ml in tree.mutations],
        nMinor=[len(m) if m else 0 for m in tree.minorSequences],
        probVect=[ser_list(x) for x in tree.probVect],
        probVectUpRight=[ser_list(x) for x in tree.probVectUpRight],
        probVectUpLeft=[ser_list(x) for x in tree.probVectUpLeft],
        probVectTotUp=[ser_list(x) for x in tree.probVectTotUp],
        n=n)


def perturb(diffs, ref, rng):
    """A new sample close to an existing one: drop / add a substitution, maybe an N run."""
    out = [tuple(e) for e in diffs]
    if out and rng.random() < 0.5:
        out.pop(rng.randrange(len(out)))
    for _ in range(rng.randrange(0, 3)):
        p = rng.randrange(1, len(ref) + 1)
        if any((e[1] <= p <= e[1] + (e[2] - 1 if len(e) > 2 else 0)) for e in out):
            continue
        ch = rng.choice([c for c in "acgt" if c != ref[p - 1]])
        out.append((ch, p))
    if rng.random() < 0.3:
        p = rng.randrange(1, len(ref) - 40)
        ln = rng.randrange(3, 30)
        out = [e for e in out if not (p <= e[1] < p + ln) and not (len(e) > 2 and e[1] <= p + ln and p <= e[1] + e[2] - 1)]
        out.append(("n", p, ln))
    out.sort(key=lambda e: e[1])
    return out


def frozen_reference_tree(name, flags):
    """Run the unmodified reference as __main__ on the run's input and return (its globals, the tree it ended with, root,
    printed log-likelihoods, input path) with every genome list recomputed from the tips.  With an error model the
    reference's tips share ambiguity vectors with one another (one table object per IUPAC code) and
    reCalculateAllGenomeLists rewrites them in place tip by tip (updateProbVectTerminalNode, M:3966 / 6130), so that an
    internal list depends on which tip was visited last; the tips are de-aliased first, which makes every list of the
    frozen tree a function of its tips' lists."""
    out_dir = tempfile.mkdtemp(prefix="maple_golden_search_")
    inp = input_path(name, out_dir)
    argv = ["MAPLE", "--input", inp, "--output", os.path.join(out_dir, "out"), "--overwrite"] + flags
    if name in SCRAMBLE:
        old = sys.argv
        sys.argv = argv
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                runpy.run_path(REF, run_name="__main__")
        except SystemExit:
            pass
        finally:
            sys.argv = old
        newick = open(os.path.join(out_dir, "out_tree.tree")).read()
        tip_names = sorted(set(re.findall(r"[(,]([A-Za-z][^:,()]*):", newick)))
        rng0 = random.Random(99)
        chosen = rng0.sample(tip_names, min(SCRAMBLE[name], len(tip_names)))
        swap = dict(zip(chosen, chosen[1:] + chosen[:1]))
        scrambled = re.sub(r"([(,])([A-Za-z][^:,()]*):", lambda m: m.group(1) + swap.get(m.group(2), m.group(2)) + ":", newick)
        tree_path = os.path.join(out_dir, "scrambled.tree")
        open(tree_path, "w").write(scrambled)
        argv = ["MAPLE", "--input", inp, "--output", os.path.join(out_dir, "out2"), "--overwrite", "--inputTree", tree_path,
                "--largeUpdate"] + flags
    holder = {}

    def grab(frame, event, arg):
        if "g" not in holder and frame.f_code.co_filename.endswith("MAPLEv0.7.5.4.py"):
            holder["g"] = frame.f_globals
        return None

    old = sys.argv
    sys.argv = argv
    log = io.StringIO()
    sys.setprofile(grab)
    try:
        with contextlib.redirect_stdout(log):
            runpy.run_path(REF, run_name="__main__")
    except SystemExit:
        pass
    except Exception as e:                  # --model JC: "EM for given model JC not implemented" after round 1
        print(f"[{name}] the reference raised: {e!r}; continuing with the tree it had built", flush=True)
    finally:
        sys.setprofile(None)
        sys.argv = old
    g = holder["g"]
    tree, t1 = g["tree"], g["t1"]
    run_log = log.getvalue()
    printed_lk = [float(x) for x in re.findall(r"ikelihood[^\n]*?(-\d+\.\d+)", run_log)]
    for v in range(len(tree.up)):
        if tree.probVect[v] is not None and not tree.children[v]:
            tree.probVect[v] = copy.deepcopy(tree.probVect[v])
    with contextlib.redirect_stdout(io.StringIO()):
        g["setAllDirty"](tree, t1)
        g["reCalculateAllGenomeLists"](tree, t1)
        g["assignCoreNumbers"](tree, t1, 1)
    for i in range(len(tree.replacements)):
        tree.replacements[i] = 0
    return g, tree, t1, printed_lk, inp


def run(name, flags):
    g, tree, t1, printed_lk, inp = frozen_reference_tree(name, flags)
    snap = snapshot_tree(tree, t1)
    with contextlib.redirect_stdout(io.StringIO()):
        tree_lk = g["calculateTreeLikelihood"](tree, t1)          # the parity metric "tree log-LK" (M:9721-9779)
        root_pv = tree.probVect[t1]
        if tree.mutations[t1]:
            root_pv = g["passGenomeListThroughBranch"](root_pv, tree.mutations[t1], dirIsUp=True)
        root_lk = g["findProbRoot"](root_pv)
    lRef = g["lRef"]
    model = dict(useRateVariation=bool(g["useRateVariation"]), usingErrorRate=bool(g["usingErrorRate"]),
                 errorRateSiteSpecific=bool(g["errorRateSiteSpecific"]), Q=[list(r) for r in g["mutMatrixGlobal"]],
                 siteRates=list(g["siteRates"]) if g["useRateVariation"] else None,
                 errorRateGlobal=g["errorRateGlobal"], totError=g["totError"],
                 errorRates=list(g["errorRates"]) if (g["usingErrorRate"] and g["errorRateSiteSpecific"]) else None)
    keys = ["lRef", "thresholdProb", "minBLenSensitivity", "thresholdDiffForUpdate", "thresholdFoldChangeUpdate",
            "oneMutBLen", "effectivelyNon0BLen", "thresholdLogLK", "thresholdLogLKoptimization",
            "thresholdLogLKoptimizationTopology", "thresholdLogLKtopology", "thresholdLogLKconsecutivePlacement",
            "allowedFails", "allowedFailsTopology", "defaultBLen", "maxReplacements", "strictStopRules",
            "thresholdTopologyPlacement", "thresholdLogLKtopologyInitial", "allowedFailsTopologyInitial",
            "minBranchSupport"]
    ctx = {k: g[k] for k in keys}
    ctx["rootFreqs"] = list(g["rootFreqs"])
    ctx["ref"] = g["ref"]

    # ---- the --numCores shard lists: coreNum[node] of assignCoreNumbers (M:12164-12195) for 2 and 3 cores ----
    core_numbers = {}
    for nc in (2, 3):
        with contextlib.redirect_stdout(io.StringIO()):
            g["assignCoreNumbers"](tree, t1, nc)
        core_numbers[str(nc)] = list(tree.coreNum)
    with contextlib.redirect_stdout(io.StringIO()):
        g["assignCoreNumbers"](tree, t1, 1)

    # ---- a11 / a13: SPR searches on the frozen tree, two parameter sets (fast initial round, deep round) ----
    param_sets = [
        dict(strict=True, fails=g["allowedFailsTopologyInitial"], thr=g["thresholdLogLKtopologyInitial"], place=-0.1),
        dict(strict=bool(g["strictTopologyStopRules"]), fails=g["allowedFailsTopology"], thr=g["thresholdLogLKtopology"],
             place=g["thresholdTopologyPlacement"]),
    ]
    spr = []
    for ps in param_sets:
        calls = []
        moves = []
        state = {"cur": None, "n_append": 0}

        def prof(frame, event, arg):
            co = frame.f_code
            if not co.co_filename.endswith("MAPLEv0.7.5.4.py"):
                return
            if co.co_name == "findBestParentTopology":
                if event == "call":
                    L = frame.f_locals
                    state["cur"] = dict(node=L["node"], child=L["child"], bestLKdiff=L["bestLKdiff"],
                                        removedBLen=L["removedBLen"])
                    state["n_append"] = 0
                elif event == "return" and state["cur"] is not None:
                    r = state["cur"]
                    if arg is not None:
                        r["ret"] = dict(bestNode=arg[0], bestScore=arg[1], bestBranchLengths=list(arg[2]),
                                        bestRemovedPartials=ser_list(arg[5]))
                    else:
                        r["ret"] = None
                    r["n_append"] = state["n_append"]
                    calls.append(r)
                    state["cur"] = None
            elif co.co_name == "appendProbNode" and event == "call" and state["cur"] is not None:
                state["n_append"] += 1

        # The reference shortens genome lists of the tree in place while it searches (M:7087), which makes a
        # query's float results depend (at the 1e-9 level, enough to flip exact ties) on the queries before it.
        # Frozen-tree semantics: every pruned node is searched on its own pristine copy of the tree.
        for v in range(len(tree.up)):
            tcopy = copy.deepcopy(tree)
            for i in range(len(tcopy.dirty)):
                tcopy.dirty[i] = (i == v)
            tup = (tcopy, t1, 0, ps["strict"], ps["fails"], ps["thr"], ps["place"], None, g["errorRateGlobal"],
                   g["mutMatrixGlobal"], g["errorRates"], g["mutMatrices"], g["cumulativeRate"],
                   g["cumulativeErrorRate"])
            sys.setprofile(prof)
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    moves.extend(g["startTopologyUpdatesParallel"](tup))
            finally:
                sys.setprofile(None)
        spr.append(dict(params=ps, calls=calls, proposedMoves=[list(m) for m in moves]))
        print(f"[{name}] SPR params {ps}: {len(calls)} searches, {len(moves)} proposed moves, "
              f"{sum(c['n_append'] for c in calls)} appendProbNode calls", flush=True)

    # ---- a12: placement searches for new samples on the frozen tree ----
    # (the reference imports math.exp only under --findSamplePlacements / --lineageRefs / --aBayesPlus (M:289-290), the
    # modes whose call site passes computePlacementSupportOnly=True; the runs here do not set them)
    import math
    g.setdefault("exp", math.exp)
    rng = random.Random(123)
    from maple_amd.host import read_maple_file
    _, data = read_maple_file(inp)
    n_place = 60 if len(tree.up) < 400 else 40
    names = sorted(data)
    placements = []
    for k in range(n_place):
        base = data[names[rng.randrange(len(names))]]
        diffs = perturb(base, g["ref"], rng)
        tcopy = copy.deepcopy(tree)
        state = {"n_append": 0}

        def prof2(frame, event, arg):
            if event == "call" and frame.f_code.co_name == "appendProbNode":
                state["n_append"] += 1

        with contextlib.redirect_stdout(io.StringIO()):
            q = g["probVectTerminalNode"](diffs, None, None)
            q_ser = ser_list(q)
            sys.setprofile(prof2)
            try:
                ret = g["findBestParentForNewSample"](tcopy, t1, q, f"new{k}", False)
            finally:
                sys.setprofile(None)
        # the same query through the computePlacementSupportOnly=True exit (M:8101-8290), the form process_chunk
        # consumes (M:11200): (possiblePlacements, bestPlacementTotalLh)
        tcopy2 = copy.deepcopy(tree)
        with contextlib.redirect_stdout(io.StringIO()):
            q2 = g["probVectTerminalNode"](diffs, None, None)
            sup = g["findBestParentForNewSample"](tcopy2, t1, q2, f"new{k}", True)
        placements.append(dict(diffs=[list(e) for e in diffs], query=q_ser, n_append=state["n_append"],
                               ret=dict(bestNode=ret[0], bestScore=ret[1],
                                        bestBranchLengths=None if ret[2] is None else list(ret[2]),
                                        bestDiffs=ser_list(ret[3])),
                               supports=dict(possiblePlacements=[[p[0], p[1], [0.0 if b is False else b for b in p[2]]]
                                                                 for p in sup[0]],
                                             bestPlacementTotalLh=ser_list(sup[1]))))
    print(f"[{name}] {len(placements)} placement searches, "
          f"{sum(p['n_append'] for p in placements)} appendProbNode calls", flush=True)

    fixture = dict(name=name, flags=flags, context=ctx, model=model, tree=snap, spr=spr, placements=placements,
                   treeLK=tree_lk, rootLK=root_lk, printedLKs=printed_lk, coreNum=core_numbers,
                   input=os.path.basename(inp))
    path = os.path.join(HERE, f"search_{name}.json.gz")
    with gzip.open(path, "wt") as fh:
        json.dump(fixture, fh)
    print(f"[{name}] -> {path} {os.path.getsize(path)/1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(RUNS)):
        run(nm, RUNS[nm])
