"""GPU checks of the genome-list arena's bookkeeping: stack discipline with lists that maple_lists_update moved, and
compaction after a run of incremental repairs."""
import gzip
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, model_args, ref_indices, tup

pytestmark = pytest.mark.gpu


def _env(name="synth_unrest"):
    from maple_amd.runtime import Device
    from maple_amd.tree_host import HostTree
    with gzip.open(os.path.join(GOLDEN, f"search_{name}.json.gz"), "rt") as fh:
        f = json.load(fh)
    ctx, t = f["context"], f["tree"]
    dev = Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"], minBLenSensitivity=ctx["minBLenSensitivity"],
                 thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"], thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"],
                 defaultBLen=ctx["defaultBLen"], arena_bytes=256 << 20)
    dev.set_model(**model_args(f["model"]))
    tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"], t["probVectUpRight"],
                    t["probVectUpLeft"], t["probVectTotUp"]).upload(dev)
    return f, dev, tree


def test_release_keeps_the_room_of_a_list_that_an_update_moved():
    """mark -> maple_lists_update with a LONGER list (it moves to the end of the arena) -> release -> upload: the updated
    list must survive the upload that follows the release."""
    f, dev, tree = _env()
    lists = [tup(x) for x in f["tree"]["probVect"] if x]
    short = min(lists, key=len)
    long_ = max(lists, key=len)
    assert len(long_) > len(short)
    ids = dev.upload([short, short])
    mark = dev.mark()
    tmp = dev.upload([long_] * 3)                       # temporaries above the mark
    dev.update_lists(ids[:1], [long_])                  # does not fit the old room: moved above the temporaries
    assert dev.download(ids[:1])[0] == long_
    dev.release(mark)
    again = dev.upload([short] * 50)                    # would overwrite the moved list if the release had freed its room
    assert dev.download(ids[:1])[0] == long_
    assert dev.download(ids[1:2])[0] == short
    assert all(x == short for x in dev.download(again))
    dev.close()


def test_compaction_after_many_repairs_keeps_every_list_of_the_tree():
    from maple_amd.tree_host import compact_arena, tree_log_likelihood, update_genome_lists
    f, dev, tree = _env()
    lk0, _ = tree_log_likelihood(dev, tree)
    rng = np.random.default_rng(3)
    cand = [v for v in tree.preorder() if tree.up[v] is not None and tree.dist[v] > 1e-5]
    for v in rng.choice(cand, size=40, replace=False):
        tree.dist[v] = tree.dist[v] * 1.3
        update_genome_lists(dev, tree, [int(v)])
    lk1, _ = tree_log_likelihood(dev, tree)
    before = dev.stats()
    cols = ("id_lower", "id_upRight", "id_upLeft", "id_totUp")
    want = {a: dev.download(getattr(tree, a)[getattr(tree, a) >= 0]) for a in cols}
    compact_arena(dev, tree)
    after = dev.stats()
    assert after["n_lists"] < before["n_lists"] and after["n_entries"] < before["n_entries"]
    for a in cols:
        ids = getattr(tree, a)
        assert dev.download(ids[ids >= 0]) == want[a], a
    lk2, _ = tree_log_likelihood(dev, tree)
    assert lk2 == lk1 and lk1 != lk0
    # the searches run on the re-uploaded tree
    ctx = f["context"]
    ps = f["spr"][1]["params"]
    out = dev.spr_search_batch(list(range(10)), strict=ps["strict"], allowedFails=ps["fails"], thresholdLogLKtopology=ps["thr"],
                               thresholdTopologyPlacement=ps["place"],
                               thresholdLogLKoptimizationTopology=ctx["thresholdLogLKoptimizationTopology"],
                               thresholdLogLKconsecutivePlacement=ctx["thresholdLogLKconsecutivePlacement"],
                               effectivelyNon0BLen=ctx["effectivelyNon0BLen"])
    assert (out["status"] >= -1).all()
    dev.close()


def test_rebuild_lists_inside_the_library_edge_cases():
    """maple_tree_rebuild_lists (reCalculateAllGenomeLists, M:6013-6347): two tips that differ at a site and hang on zero-length
    branches cannot be merged (mergeVectors returns None, M:4753-4758): with bumpLen = 0 the call fails like the reference would
    (it calls updateBLen there), with bumpLen > 0 the two branches are lengthened and the build goes through; malformed
    trees are refused; downloads of lists that are interleaved in the arena come back intact whatever the order."""
    from maple_amd.runtime import MapleError
    f, dev, tree = _env()
    l_ref = dev.lRef
    ref = ref_indices(f["context"])
    a = [(4, l_ref)]                                                          # the reference itself
    other = (int(ref[9]) + 1) % 4
    b = [(4, 9), (other, int(ref[9])), (4, l_ref)]                            # one difference at site 10
    ids = dev.upload([a, b])
    up = np.asarray([-1, 0, 0], np.int32)
    c0 = np.asarray([1, -1, -1], np.int32)
    c1 = np.asarray([2, -1, -1], np.int32)
    tip = np.asarray([0, 1, 1], np.uint8)
    lower = np.asarray([-1, ids[0], ids[1]], np.int32)
    dist = np.zeros(3)
    with pytest.raises(MapleError) as e:
        dev.tree_rebuild_lists(0, up, c0, c1, tip, None, dist, lower)
    assert e.value.code == MapleError.ERR_FATAL
    dist = np.zeros(3)
    lo, ur, ul, tu, bad = dev.tree_rebuild_lists(0, up, c0, c1, tip, None, dist, lower, bump_len=0.1 / l_ref)
    assert len(bad) == 0 and dist[1] > 0 and dist[2] > 0 and dist[0] == 0
    assert lo[0] >= 0 and ur[0] >= 0 and ul[0] >= 0 and tu[1] >= 0 and tu[2] >= 0 and tu[0] == -1
    root_list = dev.download(lo[:1])[0]
    assert root_list[0][0] == 4 and root_list[-1][1] == l_ref and any(e[0] == 6 for e in root_list)   # an O entry at the site that differs
    # a node with one child, a child index out of range
    with pytest.raises(MapleError) as e:
        dev.tree_rebuild_lists(0, up, np.asarray([1, -1, -1], np.int32), np.asarray([-1, -1, -1], np.int32), tip, None, np.zeros(3), lower)
    assert e.value.code == MapleError.ERR_ARG
    with pytest.raises(MapleError) as e:
        dev.tree_rebuild_lists(0, up, np.asarray([7, -1, -1], np.int32), c1, tip, None, np.zeros(3), lower)
    assert e.value.code == MapleError.ERR_ARG
    # lists of different kinds interleaved in the arena, downloaded in an order of their own
    mix = np.asarray([tu[2], lo[0], ids[1], ur[0], ids[0], tu[1], ul[0]], np.int32)
    got = dev.download(mix)
    for i, l in zip(mix, got):
        assert l == dev.download([i])[0]
    assert got[2] == b and got[4] == a
    dev.close()
