"""GPU parity of the device-resident SPR search against records of the
reference's own startTopologyUpdatesParallel / findBestParentTopology run on a frozen tree
(tests/golden/search_*.json.gz, made by tests/golden/make_golden_search.py).

Bar: node ids, proposed moves and the number of candidate placements scored (appendProbNode calls)
bit-exact; scores / branch lengths within 1e-8 relative (the reference shortens shared lists in place
while it searches, which moves later results at the 1e-12 level).
"""
import gzip
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, close, lists_match, model_args, ref_indices, tup

pytestmark = pytest.mark.gpu
NAMES = sorted(f[len("search_"):-len(".json.gz")] for f in os.listdir(GOLDEN) if f.startswith("search_"))


def load(name):
    with gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt") as fh:
        return json.load(fh)


@pytest.fixture(scope="module", params=NAMES)
def env(request):
    from maple_amd.runtime import Device
    from maple_amd.tree_host import HostTree
    f = load(request.param)
    ctx, t = f["context"], f["tree"]
    dev = Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"],
                 minBLenSensitivity=ctx["minBLenSensitivity"], thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"],
                 thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"], defaultBLen=ctx["defaultBLen"],
                 arena_bytes=256 << 20)
    dev.set_model(**model_args(f["model"]))
    tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"],
                    t["probVectUpRight"], t["probVectUpLeft"], t["probVectTotUp"]).upload(dev)
    yield f, dev, tree
    dev.close()


def test_fixture_present():
    assert NAMES


def test_spr_search_matches_reference(env):
    f, dev, tree = env
    ctx = f["context"]
    for rnd in f["spr"]:
        ps = rnd["params"]
        calls = rnd["calls"]
        # the pruned node of each recorded findBestParentTopology(tree, node, child, ...) call
        nodes = [tree.children[c["node"]][c["child"]] for c in calls]
        out = dev.spr_search_batch(nodes, strict=ps["strict"], allowedFails=ps["fails"],
                                   thresholdLogLKtopology=ps["thr"], thresholdTopologyPlacement=ps["place"],
                                   thresholdLogLKoptimizationTopology=ctx["thresholdLogLKoptimizationTopology"],
                                   thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"],
                                   effectivelyNon0BLen=ctx["effectivelyNon0BLen"], want_removed_partials=True)
        rpr = dev.download(out["removedPartials"])
        n_total = 0
        for k, c in enumerate(calls):
            assert out["status"][k] == 0, (k, out["status"][k])
            assert close(float(out["currentLK"][k]), c["bestLKdiff"], 1e-9)
            want = c["ret"]
            assert want is not None
            assert int(out["bestNode"][k]) == want["bestNode"], (k, out["bestNode"][k], want["bestNode"])
            assert close(float(out["bestScore"][k]), want["bestScore"], 1e-8), (out["bestScore"][k], want["bestScore"])
            wb = [0.0 if b is False else b for b in want["bestBranchLengths"]]
            assert all(close(float(g), w, 1e-7, 1e-15) for g, w in zip(out["blen"][k], wb)), (out["blen"][k], wb)
            assert int(out["nAppend"][k]) == c["n_append"], (k, out["nAppend"][k], c["n_append"])
            assert lists_match(rpr[k], tup(want["bestRemovedPartials"]), 1e-7), (rpr[k], want["bestRemovedPartials"])
            n_total += c["n_append"]
        assert n_total > 1000
        # a13: the proposed moves (node, placement, improvement)
        got_moves = sorted((nodes[k], int(out["placement"][k]), float(out["improvement"][k]))
                           for k in range(len(nodes)) if out["placement"][k] >= 0)
        want_moves = sorted(tuple(m) for m in rnd["proposedMoves"])
        assert [(m[0], m[1]) for m in got_moves] == [(m[0], m[1]) for m in want_moves]
        assert all(close(g[2], w[2], 1e-7) for g, w in zip(got_moves, want_moves))


def test_nodes_not_searched(env):
    """Nodes the reference's worker skips (root; zero-length branch with a good current cost) report why."""
    f, dev, tree = env
    ctx = f["context"]
    ps = f["spr"][0]["params"]
    searched = {tree.children[c["node"]][c["child"]] for c in f["spr"][0]["calls"]}
    others = [v for v in tree.preorder() if v not in searched]
    out = dev.spr_search_batch(others, strict=ps["strict"], allowedFails=ps["fails"], thresholdLogLKtopology=ps["thr"],
                               thresholdTopologyPlacement=ps["place"],
                               thresholdLogLKoptimizationTopology=ctx["thresholdLogLKoptimizationTopology"],
                               thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"],
                               effectivelyNon0BLen=ctx["effectivelyNon0BLen"])
    for v, st in zip(others, out["status"]):
        assert st == (1 if tree.up[v] is None else 2), (v, st)
    assert (out["placement"] == -1).all()


def test_placement_search_matches_reference(env):
    """a12: findBestParentForNewSample on the frozen tree for 60 new samples."""
    from maple_amd.search import PlacementParams, PlacementSearcher
    f, dev, tree = env
    ctx = f["context"]
    any_err = f["model"]["usingErrorRate"]
    flags = f["flags"]
    only_identical = any(x in flags for x in ("--estimateErrorRate", "--estimateSiteSpecificErrorRate"))
    ps = PlacementSearcher(dev, tree, PlacementParams(
        oneMutBLen=ctx["oneMutBLen"], effectivelyNon0BLen=ctx["effectivelyNon0BLen"],
        thresholdLogLK=ctx["thresholdLogLK"], thresholdLogLKoptimization=ctx["thresholdLogLKoptimization"],
        thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"],
        allowedFails=ctx["allowedFails"], strictStopRules=ctx["strictStopRules"], onlyFindIdentical=only_identical))
    n_real = 0
    for rec in f["placements"]:
        node, score, blens, best_diffs, info = ps.find_best_parent_for_new_sample(tup(rec["query"]))
        want = rec["ret"]
        assert node == want["bestNode"], (node, want["bestNode"])
        assert close(score, want["bestScore"], 1e-9), (score, want["bestScore"])
        if want["bestBranchLengths"] is None:
            assert blens is None
        else:
            wb = [0.0 if b is False else b for b in want["bestBranchLengths"]]
            gb = [0.0 if b is False else b for b in blens]
            assert all(close(g, w, 1e-8, 1e-15) for g, w in zip(gb, wb)), (gb, wb)
            assert info["n_append"] == rec["n_append"], (info["n_append"], rec["n_append"])
            n_real += 1
        assert lists_match(best_diffs, tup(want["bestDiffs"]), 0.0), (best_diffs, want["bestDiffs"])
    assert n_real > 20
