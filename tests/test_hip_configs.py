"""BASELINE.json's configs[2], [3] and [4] at their stated sizes, through the C ABI, on one MI355X.

The reference cannot run at these sizes in minutes (CPython: ~0.1 s per placement); the checker is the C oracle -- itself
pinned to the reference's recorded searches (tests/test_oracle_search_golden.py) -- run on evenly spread samples of the
searches over the WHOLE downloaded tree, plus properties that do not depend on the size:

* configs[2]: 100 000 samples, UNREST + per-site rates: one whole deep SPR round (every node of the tree), every 400th search
  against the oracle (status, best node, move, candidate count exact; scores and lengths 1e-9), a second call identical to
  the first, the lane tier (another implementation of the same search) identical on 4 096 nodes;
* configs[3]: 1 000 000 samples, UNREST + per-site rates + per-site error rates: 16 384 evenly spread searches, every 48th
  against the oracle, the same properties;
* configs[4]: the online update at its stated size: 50 000 new samples added one after the other to the 1 000 000-tip tree
  (placement search -- the samples announced 512 at a time, maple_placement_ahead --, tree edit, maple_update_partials,
  maple_tree_patch), then a deep round over an even sample of the nodes the additions touched and 8 192 others -- see the test.
"""
import math
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS_EXACT = ("status", "bestNode", "placement", "nAppend")


def build_bench_tree(samples, model, arena_gb):
    """The bench's tree (synth v2, bench.model_kwargs), genome lists built on the GPU, uploaded for the searches."""
    import bench
    from maple_amd.host import reference_tables, tip_lists_packed
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset_native
    from maple_amd.tree_mirror import TreeMirror
    data = make_dataset_native(n_samples=samples, l_ref=29903, seed=1, mean_diffs=30.0, rate_variation=(model != "unrest"))
    ref_idx, root_freqs = reference_tables(data.ref)
    dev = Device(ref_idx, root_freqs, arena_bytes=arena_gb << 30)
    mkw = bench.model_kwargs(model, len(ref_idx))
    dev.set_model(**mkw)
    tip_kw = dict(error_rates=mkw["errorRates"]) if model == "siteerr" else {}
    c = data.diffs
    mirror = TreeMirror(dev, data.parent, data.blen, tip_packed=(data.tip_node, tip_lists_packed(c.off, c.code, c.pos, c.length, ref_idx,
                                                                                                **tip_kw))).build()
    dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist, mirror.is_tip, mirror.lower,
                    mirror.up_right, mirror.up_left, mirror.tot_up, -np.ones(mirror.n_nodes, dtype=np.int32))
    return data, dev, mirror, ref_idx, root_freqs, mkw, tip_kw


def oracle_tree(dev, ref_idx, root_freqs, mkw, root, parent, children, dist, lower, up_right, up_left, tot_up):
    """The whole tree, lists and all, downloaded into the oracle's layout."""
    from oracle.oracle_py import Oracle, OracleTree
    orc = Oracle(ref_idx, root_freqs)
    orc.set_model(**mkw)
    lists4 = []
    for ids in (lower, up_right, up_left, tot_up):
        have = np.nonzero(ids >= 0)[0]
        have = have[np.argsort(ids[have], kind="stable")]
        lists4.append((have, dev.download_packed(ids[have])))
    n = len(parent)
    return orc, OracleTree(orc, int(root), np.asarray(parent, dtype=np.int32), np.asarray(children), dist, None, np.zeros(n, dtype=np.int32), lists4)


def check_sample_against_oracle(orc, otree, nodes, gpu, sel, kw, max_ties=3):
    import bench
    o = orc.spr_worker(otree, nodes[sel], threads=bench.usable_host_threads(), **kw)
    tie = (np.abs(o["bestScore"] - gpu["bestScore"][sel]) <= 1e-11 * np.maximum(1.0, np.abs(o["bestScore"]))) \
        & (o["bestNode"] != gpu["bestNode"][sel]) & (o["bestNode"] >= 0) & (gpu["bestNode"][sel] >= 0)
    assert tie.sum() <= max_ties, int(tie.sum())
    for k in KEYS_EXACT:
        diff = o[k] != gpu[k][sel]
        if k in ("bestNode", "placement"):
            diff &= ~tie
        assert not diff.any(), (k, nodes[sel][diff][:5], o[k][diff][:5], gpu[k][sel][diff][:5])
    ok = (o["status"] == 0) & ~tie
    for k in ("bestScore", "currentLK", "improvement"):
        a, b = o[k][ok], gpu[k][sel][ok]
        both_inf = np.isinf(a) & np.isinf(b) & (a == b)
        assert (both_inf | (np.abs(a - b) <= 1e-9 * np.maximum(1.0, np.abs(a)))).all(), k
    assert np.allclose(o["blen"][ok], gpu["blen"][sel][ok], rtol=1e-8, atol=1e-15)
    return int(ok.sum()), int(o["nAppend"].sum())


def same_results(a, b, sel_a=None, sel_b=None):
    for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
        x = a[k] if sel_a is None else a[k][sel_a]
        y = b[k] if sel_b is None else b[k][sel_b]
        assert np.array_equal(x, y), k


def test_config3_100k_ratevar_deep_round_against_the_oracle():
    import bench
    t0 = time.time()
    data, dev, m, ref_idx, root_freqs, mkw, _ = build_bench_tree(100000, "ratevar", 16)
    kw = bench.search_kwargs(dev.lRef)
    nodes = bench.preorder_nodes(m)
    assert len(nodes) == 199999
    g = dev.spr_search_batch(nodes, **kw)
    assert not (g["status"] < -1).any()
    # the shape of a deep round on this tree: most nodes searched, ~1 in 8 a whole-tree search from a zero-length branch
    searched = g["status"] == 0
    assert searched.sum() > 150000 and g["nAppend"][searched].sum() > 1e9
    assert ((g["placement"] >= 0) <= searched).all() and (g["improvement"][g["placement"] >= 0] > kw["thresholdTopologyPlacement"]).all()
    # a second call gives the same answers (pools sized by the first), and so does the other implementation of the search
    same_results(dev.spr_search_batch(nodes, **kw), g)
    sub = np.arange(0, len(nodes), len(nodes) // 4096)[:4096]
    same_results(dev.spr_search_batch(nodes[sub], search_tier=1, **kw), g, None, sub)
    orc, otree = oracle_tree(dev, ref_idx, root_freqs, mkw, m.root, m.parent, m.children, m.dist, m.lower, m.up_right, m.up_left, m.tot_up)
    sel = np.arange(7, len(nodes), 400)
    n_ok, n_pl = check_sample_against_oracle(orc, otree, nodes, g, sel, kw)
    assert n_ok > 350 and n_pl > 1e7
    dev.close()
    print(f"config 3: {len(sel)} searches ({n_pl} candidate placements) equal the oracle's; {time.time() - t0:.0f} s")


def test_config3_on_the_headline_tree_form_local_references_and_optimised_lengths():
    """The tree `python bench.py` times (bench.build_bench_tree: synth v2, branch lengths optimised as MAPLE does before its SPR
    rounds, MAT local references -- the form MAPLE's own trees have): one whole deep round; a sample of the searches against the
    C oracle on the downloaded tree WITH its mutation lists (bench.spr_cpu_baseline: node ids, moves, candidate counts exact);
    the lane tier identical on 2 048 nodes; and the metric's second half -- the tree log-likelihood by the library against the
    oracle's value from the tips' lists alone (bench.tree_log_lk_check), 1e-6 relative by the north star, 1e-9 here."""
    import bench
    t0 = time.time()
    bt = bench.build_bench_tree(100000, "ratevar")
    dev, m = bt.dev, bt.mirror
    assert bt.ht is not None and bt.n_ref > 1000 and bt.blen_opt["passes"] >= 1
    kw = bench.search_kwargs(dev.lRef)
    nodes = bench.preorder_nodes(m)
    g = dev.spr_search_batch(nodes, **kw)
    assert not (g["status"] < -1).any()
    searched = g["status"] == 0
    assert searched.sum() > 150000 and g["nAppend"][searched].sum() > 1e9
    same_results(dev.spr_search_batch(nodes, **kw), g)
    # the other implementation of the search (one lane per search, which does what the reference does step by step -- the in-place
    # shorten() of M:7087 included, which the frontier tier emulates) on EVERY node of the round
    same_results(dev.spr_search_batch(nodes, search_tier=1, **kw), g)
    cb = bench.spr_cpu_baseline(dev, m, bt.ht, bt.ref_idx, bt.root_freqs, lambda i: nodes, [g], kw, 6.0, bt.mkw, 1)   # (SystemExit on a mismatch)
    assert cb["value"] > 0 and "identical" in cb["sample"]
    lk = bench.tree_log_lk_check(dev, m, bt.ht, bt.tip_ids, bt.mkw, bt.ref_idx, bt.root_freqs)
    assert lk["rel_delta"] <= 1e-9 and lk["gpu"] < 0, lk
    dev.close()
    print(f"config 3, headline form: {bt.n_ref} reference nodes; {cb['sample']}; tree log-LK {lk['gpu']:.6f} vs oracle {lk['oracle']:.6f} "
          f"(rel {lk['rel_delta']:.2e}); {time.time() - t0:.0f} s")


@pytest.fixture(scope="module")
def million():
    t0 = time.time()
    data, dev, m, ref_idx, root_freqs, mkw, tip_kw = build_bench_tree(1000000, "siteerr", 64)
    print(f"1 000 000-tip tree with the full model built and uploaded in {time.time() - t0:.0f} s")
    yield data, dev, m, ref_idx, root_freqs, mkw, tip_kw
    dev.close()


def test_config4_1M_full_model_searches_against_the_oracle(million):
    import bench
    data, dev, m, ref_idx, root_freqs, mkw, _ = million
    t0 = time.time()
    kw = bench.search_kwargs(dev.lRef)
    order = bench.preorder_nodes(m)
    assert len(order) == 1999999
    nodes = order[np.arange(3, len(order), len(order) // 16384)[:16384]]
    g = dev.spr_search_batch(nodes, **kw)
    t1 = time.time()
    assert not (g["status"] < -1).any()
    searched = g["status"] == 0
    assert searched.sum() > 12000 and g["nAppend"][searched].sum() > 1e8
    same_results(dev.spr_search_batch(nodes, **kw), g)
    t2 = time.time()
    sub = np.arange(0, len(nodes), 16)
    same_results(dev.spr_search_batch(nodes[sub], search_tier=1, **kw), g, None, sub)
    t3 = time.time()
    orc, otree = oracle_tree(dev, ref_idx, root_freqs, mkw, m.root, m.parent, m.children, m.dist, m.lower, m.up_right, m.up_left, m.tot_up)
    t4 = time.time()
    sel = np.arange(5, len(nodes), 48)
    n_ok, n_pl = check_sample_against_oracle(orc, otree, nodes, g, sel, kw)
    assert n_ok > 250 and n_pl > 1e6
    print(f"config 4: {len(sel)} of {len(nodes)} searches ({n_pl} candidate placements) equal the oracle's; {time.time() - t0:.0f} s "
          f"(first call {t1 - t0:.1f}, second {t2 - t1:.1f}, lane tier on {len(sub)} nodes {t3 - t2:.1f}, tree to the oracle {t4 - t3:.1f}, "
          f"oracle searches {time.time() - t4:.1f})")


def test_config5_online_update_of_the_1M_tree(million):
    """BASELINE configs[4]: new samples added to the 1 000 000-tip tree one after the other (M:11692-11752: placement search,
    tree edit, updatePartials), then an SPR round (--largeUpdate: the rounds run with every node dirty, M:12143-12159, 12274).
    Here all 50 000 samples through maple_placement_search_batch + maple_update_partials + maple_tree_patch (the tree edit is the
    bench's stand-in for placeSampleOnTree, host code of the reference that is out of scope), checked as follows
    (MAPLE_TEST_CONFIG5_ADD=<n> runs a shorter loop while working on the library):

    * every 500th sample: the search's score and the three branch lengths against the oracle's evaluation of the SAME placement
      (the reference's refinement, M:8109-8147: three estimateBranchLengthWithDerivative around three mergeVectors, one
      appendProbNode, the branch-length compensation), and the lists updatePartials wrote around the new nodes against the
      oracle's mergeVectors + shorten of their current inputs, entry for entry;
    * every list of the final tree that was never touched still IS the list of the tree before (same ids), and every touched node
      has all the lists it should have;
    * a deep SPR round over every node the additions touched plus 8 192 evenly spread others on the final tree: a sample of the
      searches against the oracle on the whole downloaded tree, the second call identical."""
    import bench
    from maple_amd.host import tip_genome_list
    from maple_amd.synth import perturb_diffs
    data, dev, m, ref_idx, root_freqs, mkw, tip_kw = million
    t0 = time.time()
    l_ref = dev.lRef
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=1.0 * ll, thresholdLogLKconsecutivePlacement=1.0)
    n_add = int(os.environ.get("MAPLE_TEST_CONFIG5_ADD", "50000"))
    every = 500 if n_add >= 20000 else 50
    prng = np.random.default_rng(21)
    src = prng.choice(len(data.diffs), size=n_add, replace=False)
    new_lists = [tip_genome_list(perturb_diffs(data.diffs[int(i)], data.ref, prng), ref_idx, **tip_kw) for i in src]
    n0, cap = m.n_nodes, m.n_nodes + 2 * n_add

    def grown(a, fill, dtype):
        out = np.full(cap, fill, dtype=dtype)
        out[:n0] = a
        return out
    up = grown(m.parent, -1, np.int32)
    c0, c1 = grown(m.children[:, 0], -1, np.int32), grown(m.children[:, 1], -1, np.int32)
    tip = grown(m.is_tip, 0, np.uint8)
    dist = grown(m.dist, 0.0, np.float64)
    mut = np.full(cap, -1, dtype=np.int32)
    lower, up_right = grown(m.lower, -1, np.int32), grown(m.up_right, -1, np.int32)
    up_left, tot_up = grown(m.up_left, -1, np.int32), grown(m.tot_up, -1, np.int32)
    before = dict(lower=lower.copy(), up_right=up_right.copy(), up_left=up_left.copy(), tot_up=tot_up.copy())
    depth, dstep = bench.tree_depths(m.root, c0, c1, n0, cap)
    n = n0
    dev.upload_tree(m.root, up[:n], c0[:n], c1[:n], dist[:n], tip[:n], lower[:n], up_right[:n], up_left[:n], tot_up[:n], mut[:n])
    from oracle.oracle_py import Oracle
    orc = Oracle(ref_idx, root_freqs)
    orc.set_model(**mkw)
    touched_all = []
    placed = checked = 0
    t_loop = time.time()
    waiting, rest = [], []
    for k, lst in enumerate(new_lists):
        dev.placement_prepare(**pkw)
        # (the loop as a maintainer would write it since round 6: the next 512 samples announced, maple_placement_ahead -- their
        # rows made by expansion, the traversals made ahead; every search below is still ONE sample on the tree as it is then, and
        # the checks against the oracle are the same)
        if not waiting:
            if not rest:
                rest = [int(x) for x in dev.upload(new_lists[k: k + 512])]
            got = dev.placement_ahead(np.asarray(rest, dtype=np.int32), **pkw)
            got = got if got > 0 else len(rest)
            waiting, rest = rest[:got], rest[got:]
        qid = waiting.pop(0)
        mark = dev.mark()
        out = dev.placement_search_batch(np.asarray([qid], dtype=np.int32), **pkw)
        dev.release(mark)
        b = int(out["bestNode"][0])
        if out["status"][0] != 0 or up[b] < 0:
            continue
        top, bottom, app = (float(x) for x in out["blen"][0])
        g, p, s = int(up[b]), n, n + 1
        check = (k % every == 0)
        if check:
            # the oracle's evaluation of this very placement on the lists of the tree as it is now (M:8109-8147)
            vu = int(up_right[g] if c0[g] == b else up_left[g])
            l_tot, l_low, l_up = dev.download([tot_up[b], lower[b], vu])
            is_tip_b = bool(tip[b])
            cost, o_bottom, o_top, o_app = orc.evaluatePlacement(l_tot, l_low, l_up, float(dist[b]), lst, True, is_tip_b)
            initial = orc.appendProbNode(l_up, l_low, is_tip_b, float(dist[b]))
            newpart = orc.appendProbNode(l_up, l_low, is_tip_b, o_bottom + o_top)
            want = cost + newpart - initial
            first = orc.appendProbNode(l_tot, lst, True, 1.0 / l_ref)       # the unrefined score the search arrived with (M:8049)
            got = float(out["bestScore"][0])
            if want >= first:                                               # M:8179: the refined placement replaces it
                assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (k, got, want, first)
                assert np.allclose([top, bottom, app], [o_top, o_bottom, o_app], rtol=1e-8, atol=1e-15), (k, out["blen"][0], (o_top, o_bottom, o_app))
            else:                                                           # ... or it stays, with the lengths of M:8072
                assert abs(got - first) <= 1e-9 * max(1.0, abs(first)), (k, got, want, first)
        # ---- the stand-in tree edit (bench.serial_phase): p on the branch above b, the sample s as p's other child
        depth, dstep = bench.place_depths(depth, dstep, g, p, b, s, m.root, c0, c1, n)
        if c0[g] == b:
            c0[g] = p
        else:
            c1[g] = p
        up[p], c0[p], c1[p], dist[p], tip[p] = g, b, s, top, 0
        up[b], dist[b] = p, bottom
        up[s], dist[s], tip[s], lower[s] = p, app, 1, qid
        n += 2
        dev.update_partials(m.root, up[:n], c0[:n], c1[:n], tip[:n], mut[:n], depth[:n], dist[:n], lower[:n], up_right[:n],
                            up_left[:n], tot_up[:n], [b, s, p])
        touched = np.unique(np.concatenate([dev.update_partials_touched(), [g, b, p, s]])).astype(np.int32)
        dev.tree_patch(n, touched, up[touched], c0[touched], c1[touched], dist[touched], tip[touched], lower[touched],
                       up_right[touched], up_left[touched], tot_up[touched])
        touched_all.append(touched)
        placed += 1
        if check:
            # what updatePartials wrote around the new nodes, from the lists that are there now (M:5479-5815)
            lb, ls, lp, lur, lul = dev.download([lower[b], lower[s], lower[p], up_right[p], up_left[p]])
            want_p = orc.mergeVectors(lb, float(dist[b]), bool(tip[b]), ls, float(dist[s]), True)
            assert want_p is not None
            want_p = orc.shorten(want_p)
            assert len(lp) == len(want_p) and all(x[0] == y[0] and x[1] == y[1] for x, y in zip(lp, want_p)), k
            # the new tip's mid-branch list: p's upper list for its child 1 (= s) merged with the sample (M:5700-5720)
            if dist[s] > 0 and tot_up[s] >= 0:
                (lt,) = dev.download([tot_up[s]])
                want_t = orc.mergeVectors(lul, float(dist[s]) / 2, False, ls, float(dist[s]) / 2, True, isUpDown=True)
                want_t = orc.shorten(want_t)
                assert len(lt) == len(want_t) and all(x[0] == y[0] and x[1] == y[1] for x, y in zip(lt, want_t)), k
            checked += 1
    loop_s = time.time() - t_loop
    ahead_stats = dev.placement_ahead_stats()
    assert placed > 0.95 * n_add and checked >= 0.85 * (n_add // every)
    assert ahead_stats["searches"] == n_add and ahead_stats["fallbacks"] <= 0.05 * n_add, ahead_stats
    assert ahead_stats["traversals_ahead_used"] > 0.5 * n_add, ahead_stats
    touched_all = np.unique(np.concatenate(touched_all))
    # lists that were never touched are the lists of the tree before; every node of the final tree has what it needs
    untouched = np.setdiff1d(np.arange(n0), touched_all)
    for name, col in (("lower", lower), ("up_right", up_right), ("up_left", up_left), ("tot_up", tot_up)):
        assert np.array_equal(col[untouched], before[name][untouched]), name
    inner = np.nonzero(c0[:n] >= 0)[0]
    assert (lower[:n] >= 0).all() and (up_right[inner] >= 0).all() and (up_left[inner] >= 0).all()
    assert (tot_up[:n][(dist[:n] > 0) & (up[:n] >= 0)] >= 0).all()
    # ---- the round that follows the update, on the final tree (the library's copy is current through the patches)
    kw = bench.search_kwargs(l_ref)
    rest = np.setdiff1d(np.arange(n), touched_all)
    # (at 50 000 additions ~10 nodes each are touched: the round takes an even sample of at most 122 880 of them -- with the 8 192
    # others, the 131 072 searches of a bench step on this tree)
    tsel = touched_all[:: max(1, -(-len(touched_all) // 122880))][:122880]
    nodes = np.concatenate([tsel, rest[:: max(1, len(rest) // 8192)][:8192]]).astype(np.int64)
    gres = dev.spr_search_batch(nodes, **kw)
    assert not (gres["status"] < -1).any()
    same_results(dev.spr_search_batch(nodes, **kw), gres)
    children = np.stack([c0[:n], c1[:n]], axis=1)
    orc2, otree = oracle_tree(dev, ref_idx, root_freqs, mkw, m.root, up[:n], children, dist[:n], lower[:n], up_right[:n], up_left[:n], tot_up[:n])
    sel = np.concatenate([np.arange(0, len(tsel), max(1, len(tsel) // 160)), len(tsel) + np.arange(0, 8192, 64)])
    sel = sel[sel < len(nodes)]
    n_ok, n_pl = check_sample_against_oracle(orc2, otree, nodes, gres, sel, kw)
    assert n_ok > 200
    print(f"config 5: {placed} samples added in {loop_s:.1f} s ({1e3 * loop_s / max(1, placed):.2f} ms per sample, {checked} checked against the "
          f"oracle; rows made ahead: {ahead_stats}), {len(touched_all)} nodes touched; the round after it: {len(sel)} of {len(nodes)} searches equal the oracle's; "
          f"{time.time() - t0:.0f} s")
