"""Pin the oracle's SPR search (oracle/maple_oracle_search.c) to the reference's own records of
startTopologyUpdatesParallel / findBestParentTopology on a frozen tree (tests/golden/search_*.json.gz)."""
import gzip
import json
import os

import pytest

from golden_util import GOLDEN, close, lists_match, model_args, ref_indices, tup
from oracle.oracle_py import Oracle, OracleTree

NAMES = sorted(f[len("search_"):-len(".json.gz")] for f in os.listdir(GOLDEN) if f.startswith("search_"))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_spr_search_matches_reference(name):
    with gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt") as fh:
        f = json.load(fh)
    ctx, t = f["context"], f["tree"]
    o = Oracle(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"],
               minBLenSensitivity=ctx["minBLenSensitivity"], thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"],
               thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"], defaultBLen=ctx["defaultBLen"])
    o.set_model(**model_args(f["model"]))
    tree = OracleTree(o, t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"],
                      [t["probVect"], t["probVectUpRight"], t["probVectUpLeft"], t["probVectTotUp"]])
    for rnd in f["spr"]:
        ps, calls = rnd["params"], rnd["calls"]
        nodes = [t["children"][c["node"]][c["child"]] for c in calls]
        out = o.spr_worker(tree, nodes, strict=ps["strict"], allowedFails=ps["fails"], thresholdLogLKtopology=ps["thr"],
                           thresholdTopologyPlacement=ps["place"],
                           thresholdLogLKoptimizationTopology=ctx["thresholdLogLKoptimizationTopology"],
                           thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"],
                           effectivelyNon0BLen=ctx["effectivelyNon0BLen"], want_removed_partials=True)
        for k, c in enumerate(calls):
            want = c["ret"]
            assert out["status"][k] == 0
            assert close(float(out["currentLK"][k]), c["bestLKdiff"], 1e-12)
            assert int(out["bestNode"][k]) == want["bestNode"], (k, out["bestNode"][k], want["bestNode"])
            assert close(float(out["bestScore"][k]), want["bestScore"], 1e-9)
            wb = [0.0 if b is False else b for b in want["bestBranchLengths"]]
            assert all(close(float(g), w, 1e-8, 1e-15) for g, w in zip(out["blen"][k], wb))
            assert int(out["nAppend"][k]) == c["n_append"], (k, out["nAppend"][k], c["n_append"])
            assert lists_match(out["removedPartials"][k], tup(want["bestRemovedPartials"]), 1e-9)
        got = sorted((nodes[k], int(out["placement"][k])) for k in range(len(nodes)) if out["placement"][k] >= 0)
        assert got == sorted((m[0], m[1]) for m in rnd["proposedMoves"])


@pytest.mark.parametrize("name", NAMES[:3])
def test_packed_lists_to_oracle_entries_c_equals_numpy(name):
    """The oracle library's converter of packed lists (omo_entries_from_packed, used for whole trees of 10^8 entries) against
    the numpy conversion and against to_entries() of the tuple form, on every list of a reference tree."""
    import numpy as np
    from maple_amd.genome_list import pack_lists
    from oracle.oracle_py import packed_to_entries, packed_to_entries_numpy, to_entries
    with gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt") as fh:
        f = json.load(fh)
    t, u = f["tree"], bool(f["model"]["usingErrorRate"])
    lists = [tup(gl) for kind in ("probVect", "probVectUpRight", "probVectUpLeft", "probVectTotUp") for gl in t[kind] if gl]
    pk = pack_lists(lists, u)
    a, off = packed_to_entries(pk, u, threads=3)
    b, off2 = packed_to_entries_numpy(pk, u)
    assert np.array_equal(off, off2) and a.tobytes() == b.tobytes()
    want = np.concatenate([to_entries(gl, u) for gl in lists[:400]])
    assert a[: len(want)].tobytes() == want.tobytes()
