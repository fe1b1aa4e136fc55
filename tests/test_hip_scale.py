"""GPU checks at BASELINE-like sizes through properties that do not depend on the size (the reference cannot be run
at these sizes in minutes; the C oracle, itself pinned to the reference's golden calls, is the checker on samples):

* model modes of configs[2] and configs[3]: UNREST + per-site rates, + per-site error rates;
* the query-major kernel against the oracle on a seeded sample of (query, candidate) pairs of a 10 000-tip tree;
* tree log-likelihood of the whole GPU-built mirror against the oracle's post-order sum;
* passGenomeListThroughBranch up then down restores the list (after shorten);
* shorten is idempotent; a list is never "different" from itself; merging is symmetric in the likelihood it returns.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q = [[-0.5524, 0.0602, 0.3655, 0.1267], [0.1666, -2.6077, 0.0405, 2.4006],
     [0.8421, 0.1305, -2.4012, 1.4286], [0.0688, 0.4849, 0.0502, -0.6039]]


def world_model_kwargs(mode, l_ref, seed=2):
    rng = np.random.default_rng(seed + 100)
    kw = dict(Q=Q)
    if mode != "unrest":
        kw["siteRates"] = np.clip(rng.gamma(0.5, 2.0, size=l_ref), 0.001, 0.005 * l_ref)
    if mode == "siteerr":
        er = np.exp(rng.uniform(math.log(1e-10), math.log(1e-3), size=l_ref))
        kw.update(usingErrorRate=True, errorRates=er, errorRateGlobal=float(er.mean()))
    return kw


def build(n_tips, mode, seed=2):
    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset
    from maple_amd.tree_mirror import TreeMirror
    from oracle.oracle_py import Oracle
    data = make_dataset(n_samples=n_tips, l_ref=29903, seed=seed, mean_diffs=30.0, rate_variation=(mode != "unrest"),
                        frac_with_n=0.05, frac_ambig=0.05)
    ref_idx, rf = reference_tables(data.ref)
    kw = world_model_kwargs(mode, len(ref_idx), seed)
    dev = Device(ref_idx, rf, arena_bytes=3 << 30)
    dev.set_model(**kw)
    orc = Oracle(ref_idx, rf)
    orc.set_model(**kw)
    tip_kw = dict(error_rates=kw["errorRates"]) if mode == "siteerr" else {}
    tips = {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(data.tip_node, data.diffs)}
    mirror = TreeMirror(dev, data.parent, data.blen, tips).build()
    return data, dev, orc, mirror


@pytest.fixture(scope="module", params=["unrest", "ratevar", "siteerr"])
def world(request):
    n = 10000 if request.param == "unrest" else 3000
    data, dev, orc, mirror = build(n, request.param)
    yield request.param, data, dev, orc, mirror
    dev.close()


def rel(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / max(1.0, abs(a), abs(b))


def test_query_major_kernel_vs_oracle_sample(world):
    import torch
    mode, data, dev, orc, mirror = world
    l_ref = dev.lRef
    cand = mirror.candidate_nodes(1.0 / (10 * l_ref))
    q_nodes = np.asarray(data.tip_node[:64])
    cu = torch.device("cuda", 0)
    t_q = torch.from_numpy(mirror.lower[q_nodes].astype(np.int32)).to(cu)
    t_c = torch.from_numpy(mirror.tot_up[cand].astype(np.int32)).to(cu)
    out = torch.empty(len(q_nodes) * len(cand), dtype=torch.float64, device=cu)
    torch.cuda.synchronize()
    dev.append_queries_dev(len(q_nodes), t_q.data_ptr(), len(cand), t_c.data_ptr(), True, 1.0 / l_ref, out.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(len(q_nodes), len(cand))
    assert np.isfinite(got).mean() > 0.99
    rng = np.random.default_rng(5)
    qi = rng.integers(len(q_nodes), size=400)
    ci = rng.integers(len(cand), size=400)
    q_lists = dev.download(mirror.lower[q_nodes])
    keys = sorted(set(ci.tolist()))
    c_lists = dict(zip(keys, dev.download(mirror.tot_up[cand[keys]])))
    worst = 0.0
    for a, b in zip(qi, ci):
        want = orc.appendProbNode(c_lists[int(b)], q_lists[int(a)], True, 1.0 / l_ref)
        g = float(got[a, b])
        if math.isinf(want):
            assert math.isinf(g)
        else:
            worst = max(worst, rel(g, want))
    assert worst < 1e-12, worst
    # each query scored against its own branch must be (one of) its best placements: the tree was built from the truth
    own = np.asarray([np.nonzero(cand == v)[0][0] if (cand == v).any() else -1 for v in q_nodes])
    ok = own >= 0
    best = got.max(axis=1)
    assert (got[np.arange(len(q_nodes))[ok], own[ok]] >= best[ok] - 25.0).mean() > 0.9


def test_tree_log_likelihood_vs_oracle(world):
    mode, data, dev, orc, mirror = world
    n = mirror.n_nodes
    ch = mirror.children
    internal = [v for v in range(n) if ch[v, 0] >= 0]
    c0, c1 = ch[internal, 0], ch[internal, 1]
    out, lk = dev.merge_batch(mirror.lower[c0], mirror.dist[c0], mirror.is_tip[c0], mirror.lower[c1], mirror.dist[c1],
                              mirror.is_tip[c1], False, returnLK=True)
    assert (out >= 0).all()
    root_lk = float(dev.root_prob_batch([mirror.lower[mirror.root]])[0])
    total = float(lk.sum()) + root_lk
    # oracle on a seeded sample of internal nodes + the root
    rng = np.random.default_rng(9)
    pick = rng.choice(len(internal), size=150, replace=False)
    lists0 = dev.download(mirror.lower[c0[pick]])
    lists1 = dev.download(mirror.lower[c1[pick]])
    merged = dev.download(out[pick])
    for k, i in enumerate(pick):
        res, want = orc.mergeVectors(lists0[k], mirror.dist[c0[i]], bool(mirror.is_tip[c0[i]]), lists1[k],
                                     mirror.dist[c1[i]], bool(mirror.is_tip[c1[i]]), returnLK=True)
        assert rel(float(lk[i]), want) < 1e-12
        assert len(res) == len(merged[k]) and all(x[0] == y[0] and x[1] == y[1] for x, y in zip(res, merged[k]))
    want_root = orc.findProbRoot(dev.download([mirror.lower[mirror.root]])[0], [[]])
    assert rel(root_lk, want_root) < 1e-12
    assert math.isfinite(total) and total < 0
    # symmetry: swapping the two children gives the same likelihood contribution
    out2, lk2 = dev.merge_batch(mirror.lower[c1[pick]], mirror.dist[c1[pick]], mirror.is_tip[c1[pick]],
                                mirror.lower[c0[pick]], mirror.dist[c0[pick]], mirror.is_tip[c0[pick]], False, returnLK=True)
    assert np.allclose(lk2, lk[pick], rtol=1e-12, atol=0)


def test_structural_properties(world):
    mode, data, dev, orc, mirror = world
    rng = np.random.default_rng(3)
    ids = mirror.tot_up[mirror.candidate_nodes(0.0)][:2000]
    mark = dev.mark()
    # shorten is idempotent and a list never differs from itself or from its shortened form
    s1 = dev.shorten_batch(ids)
    s2 = dev.shorten_batch(s1)
    n1, _ = dev.sizes(s1)
    n2, _ = dev.sizes(s2)
    assert np.array_equal(n1, n2)
    assert not dev.differ_batch(ids, ids).any()
    assert not dev.differ_batch(ids, s1).any()
    # a change of reference frame and back restores the list
    muts = []
    nuc = {"a": 0, "c": 1, "g": 2, "t": 3}
    for _ in range(len(ids)):
        pos = sorted(rng.choice(dev.lRef, size=5, replace=False) + 1)
        ml = []
        for p in pos:
            r = nuc[data.ref[p - 1]]
            ml.append((int(p), r, int((r + 1 + rng.integers(3)) % 4)))
        muts.append(ml)
    mids = dev.upload_mutations(muts)
    down = dev.pass_branch_batch(ids, mids, False)
    back = dev.shorten_batch(dev.pass_branch_batch(down, mids, True))
    a = dev.download(s1[:300])
    b = dev.download(back[:300])
    assert a == b
    dev.release(mark)


def test_batched_placement_equals_single_query_search(world):
    """maple_placement_search_batch (device-side traversal, one lane per query) against the one-query-at-a-time
    PlacementSearcher (host replay) on new samples: same node, score, branch lengths, bestDiffs and the same number of
    reference-equivalent appendProbNode evaluations, in every model mode."""
    from maple_amd.host import tip_genome_list
    from maple_amd.search import PlacementParams, PlacementSearcher
    from maple_amd.synth import perturb_diffs
    from maple_amd.tree_host import HostTree
    mode, data, dev, orc, mirror = world
    l_ref = dev.lRef
    ll = math.log(l_ref)
    tree = HostTree.from_mirror(mirror, dev)
    ps = PlacementSearcher(dev, tree, PlacementParams(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref),
                                                      thresholdLogLK=18.0 * ll, thresholdLogLKoptimization=ll,
                                                      thresholdLogLKconsecutivePlacement=1.0,
                                                      onlyFindIdentical=(mode == "siteerr")))
    rng = np.random.default_rng(21)
    from maple_amd.host import reference_tables
    ref_idx, _ = reference_tables(data.ref)
    queries = [tip_genome_list(perturb_diffs(dl, data.ref, rng, n_extra=k % 3), ref_idx)
               for k, dl in enumerate(data.diffs[:48])]                  # every third query is a copy of a tip
    batch = ps.find_best_parent_batch(queries)
    n_minor = 0
    for q, got in zip(queries, batch):
        want = ps.find_best_parent_host_replay(q)
        assert got[0] == want[0] and got[1] == want[1], (got[:2], want[:2])
        assert got[4]["minor"] == want[4]["minor"] and got[4]["n_append"] == want[4]["n_append"]
        if want[2] is None:
            assert got[2] is None
            n_minor += 1
        else:
            assert tuple(0.0 if b is False else b for b in want[2]) == got[2]
        assert got[3] == want[3]
    assert 0 < n_minor < len(queries)


def test_update_partials_on_a_tree_with_local_references(world):
    """maple_update_partials at size, in every model mode, on the tree with MAT local references (lists cross reference
    frames on their way up and down): 150 simultaneous branch-length changes repaired incrementally, then every list of
    the tree against a full rebuild -- "not different" by the reference's own thresholds (areVectorsDifferent, both ways:
    the repair stops where that function says so) -- and the tree log-likelihood of both."""
    from maple_amd.mat import add_local_references
    from maple_amd.tree_host import HostTree, rebuild_genome_lists, tree_log_likelihood, update_genome_lists
    _, data, dev, orc, mirror = world
    mark = dev.mark()
    tree = HostTree.from_mirror(mirror)
    assert add_local_references(dev, tree, 30) > 10
    rng = np.random.default_rng(17)
    cand = np.nonzero((mirror.parent >= 0) & (mirror.dist > 1e-5))[0]
    pick = rng.choice(cand, size=150, replace=False)
    for v in pick:
        tree.dist[v] = tree.dist[v] * 2.5
    replaced = update_genome_lists(dev, tree, pick.tolist())
    assert replaced > 300
    lk_upd, _ = tree_log_likelihood(dev, tree)
    lo, ur, ul, tu = rebuild_genome_lists(dev, tree)
    for a, b in ((tree.id_lower, lo), (tree.id_upRight, ur), (tree.id_upLeft, ul), (tree.id_totUp, tu)):
        assert np.array_equal(a >= 0, b >= 0)
        ok = a >= 0
        assert not (dev.differ_batch(a[ok], b[ok]) | dev.differ_batch(b[ok], a[ok])).any()
    tree.id_lower, tree.id_upRight, tree.id_upLeft, tree.id_totUp = lo, ur, ul, tu
    lk_full, _ = tree_log_likelihood(dev, tree)
    assert abs(lk_upd - lk_full) <= 1e-9 * abs(lk_full)
    dev.release(mark)


def test_rebuild_inside_the_library_equals_the_python_level_loop(world):
    """maple_tree_rebuild_lists (reCalculateAllGenomeLists, M:6013-6347: the level loop inside the library, one fused
    mergeVectors -> shorten launch per level) against the Python level loop of separate merge / shorten launches: every list of
    the tree entry for entry -- on the tree with MAT local references (lists passed through reference branches) and as
    TreeMirror.build()."""
    from maple_amd.mat import add_local_references
    from maple_amd.tree_host import HostTree, rebuild_genome_lists
    from maple_amd.tree_mirror import TreeMirror
    _, data, dev, orc, mirror = world
    mark = dev.mark()
    tree = HostTree.from_mirror(mirror)
    assert add_local_references(dev, tree, 30) > 10
    a = rebuild_genome_lists(dev, tree, native=True)
    b = rebuild_genome_lists(dev, tree, native=False)
    for x, y in zip(a, b):
        assert np.array_equal(x >= 0, y >= 0)
        ok = np.nonzero(x >= 0)[0][::3]
        for lx, ly in zip(dev.download(x[ok]), dev.download(y[ok])):
            assert lx == ly
    dev.release(mark)
    mark = dev.mark()
    tips = np.nonzero(mirror.is_tip)[0]
    m1 = TreeMirror(dev, mirror.parent, mirror.dist.copy(), {int(v): l for v, l in zip(tips, dev.download(mirror.lower[tips]))})
    m2 = TreeMirror(dev, mirror.parent, mirror.dist.copy(), {int(v): l for v, l in zip(tips, dev.download(mirror.lower[tips]))})
    m1.build(native=True)
    m2.build(native=False)
    assert np.array_equal(m1.dist, m2.dist)
    for x, y in ((m1.lower, m2.lower), (m1.up_right, m2.up_right), (m1.up_left, m2.up_left), (m1.tot_up, m2.tot_up)):
        assert np.array_equal(x >= 0, y >= 0)
        ok = np.nonzero(x >= 0)[0][::3]
        for lx, ly in zip(dev.download(x[ok]), dev.download(y[ok])):
            assert lx == ly
    dev.release(mark)


def test_wavefront_wide_evaluate_placement_is_the_one_lane_chain_bit_for_bit(world, monkeypatch):
    """k_evalplace_wave (one wavefront per item: the three branch-length solves, three merges and the append of
    evaluatePlacement, M:6790-6806, each cut along its merge path) against k_evalplace (one lane per item), in every model
    mode: 600 (branch, sample) pairs -- the sample's own neighbourhood and random branches -- must give the same four
    numbers, bit for bit; and single-query placement searches (which also fold the refinement's two comparison scores into
    that launch) the same results either way."""
    mode, data, dev, orc, mirror = world
    rng = np.random.default_rng(31)
    l_ref = dev.lRef
    nodes = rng.choice(np.nonzero(mirror.parent >= 0)[0], size=600, replace=False)
    par = mirror.parent[nodes]
    up_ids = np.where(mirror.children[par, 0] == nodes, mirror.up_right[par], mirror.up_left[par]).astype(np.int32)
    tips = np.asarray(data.tip_node)
    # half of the samples next to the branch (its own subtree's first tip), half anywhere
    q_nodes = rng.choice(tips, size=600)
    q_ids = mirror.lower[q_nodes].astype(np.int32)
    args = (mirror.tot_up[nodes], mirror.lower[nodes], up_ids, mirror.dist[nodes], q_ids, True, mirror.is_tip[nodes])
    ok = (mirror.tot_up[nodes] >= 0) & (up_ids >= 0)
    args = tuple(a[ok] if isinstance(a, np.ndarray) else a for a in args)
    wave = dev.evaluate_placement_batch(*args)
    dev.set_tuning(wave_per_item_max=-1)
    lane = dev.evaluate_placement_batch(*args)
    dev.set_tuning()
    assert wave.shape == lane.shape and len(wave) > 300
    assert np.array_equal(wave.view(np.uint64), lane.view(np.uint64)), np.nonzero((wave != lane).any(axis=1))[0][:5]
    assert np.isfinite(wave[:, 1:]).all() and (wave[:, 1:] >= 0).all()
    # the search around it
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=ll, thresholdLogLKconsecutivePlacement=1.0)
    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.synth import perturb_diffs
    ref_idx, _ = reference_tables(data.ref)
    mark = dev.mark()
    dev.upload_tree(mirror.root, mirror.parent, mirror.children[:, 0], mirror.children[:, 1], mirror.dist, mirror.is_tip,
                    mirror.lower, mirror.up_right, mirror.up_left, mirror.tot_up, -np.ones(mirror.n_nodes, dtype=np.int32))
    dev.placement_prepare(**pkw)
    qs = dev.upload([tip_genome_list(perturb_diffs(data.diffs[i], data.ref, rng), ref_idx) for i in range(24)])
    res = {}
    for name in ("wave", "lane"):
        if name == "lane":
            dev.set_tuning(wave_per_item_max=-1)
        res[name] = [dev.placement_search_batch(np.asarray([q], dtype=np.int32), **pkw) for q in qs]
    dev.set_tuning()
    for a, b in zip(res["wave"], res["lane"]):
        for k in ("bestNode", "nAppend", "status"):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a["bestScore"].view(np.uint64), b["bestScore"].view(np.uint64))
        assert np.array_equal(a["blen"].view(np.uint64), b["blen"].view(np.uint64))
    dev.release(mark)


def test_explicit_pair_operators_one_wavefront_per_pair_are_the_one_lane_kernels(world, monkeypatch):
    """maple_append_batch, maple_merge_batch (without the likelihood), maple_blen_batch and maple_differ_batch send batches of up to 1 024 pairs --
    a single reference call, a handful of calls -- through one wavefront per pair (k_wave_append, k_merge_wave, k_blen_wave, k_differ_wave);
    in every model mode the scores, lists and lengths must be those of the one-lane kernels, bit for bit."""
    mode, data, dev, orc, mirror = world
    rng = np.random.default_rng(41)
    nodes = rng.choice(np.nonzero(mirror.parent >= 0)[0], size=900, replace=False)
    par = mirror.parent[nodes]
    up_ids = np.where(mirror.children[par, 0] == nodes, mirror.up_right[par], mirror.up_left[par]).astype(np.int32)
    ok = (up_ids >= 0) & (mirror.tot_up[nodes] >= 0)
    nodes, up_ids = nodes[ok], up_ids[ok]
    low, tip, dist = mirror.lower[nodes], mirror.is_tip[nodes], mirror.dist[nodes]
    q = mirror.lower[rng.choice(np.asarray(data.tip_node), size=len(nodes))]
    mark = dev.mark()

    def run():
        return dict(app=dev.append_batch(mirror.tot_up[nodes], q, True, 1.0 / dev.lRef),
                    app2=dev.append_batch(up_ids, low, tip, dist),
                    low=dev.merge_batch(low, dist, tip, q, 1.0 / dev.lRef, True, False),
                    up=dev.merge_batch(up_ids, dist / 2, False, low, dist / 2, tip, True),
                    blen=dev.blen_batch(mirror.tot_up[nodes], q, True), blen2=dev.blen_batch(up_ids, low, tip),
                    dif=dev.differ_batch(low, np.roll(low, 1)), dif2=dev.differ_batch(mirror.tot_up[nodes], mirror.tot_up[nodes]))

    wave = run()
    dev.set_tuning(wave_per_item_max=-1)
    lane = run()
    dev.set_tuning()
    assert len(nodes) > 500
    for k in ("app", "app2"):
        assert np.array_equal(wave[k].view(np.uint64), lane[k].view(np.uint64)), k
    for k in ("blen", "blen2"):
        assert np.array_equal(wave[k][0].view(np.uint64), lane[k][0].view(np.uint64)) and np.array_equal(wave[k][1], lane[k][1]), k
    assert np.array_equal(wave["dif"], lane["dif"]) and wave["dif"].sum() > 400 and not wave["dif2"].any() and not lane["dif2"].any()
    for k in ("low", "up"):
        assert np.array_equal(wave[k] >= 0, lane[k] >= 0), k
        good = wave[k] >= 0
        assert good.sum() > 400 and dev.download(wave[k][good]) == dev.download(lane[k][good]), k
    dev.release(mark)


def test_wavefront_wide_update_items_leave_the_one_lane_lists(world, monkeypatch):
    """k_update_items_wave (one wavefront per item: mergeVectors, shorten and areVectorsDifferent cut along the merge path,
    wave_update.h) against k_update_items (one lane per item), in every model mode: the same 60 changes -- one at a time,
    the single-change chains the wavefront-wide kernel is for, then 40 at once -- repaired by both; every list either
    leaves must be the other's entry for entry, bit for bit."""
    from maple_amd.tree_host import HostTree, update_genome_lists
    _, data, dev, orc, mirror = world
    rng = np.random.default_rng(29)
    cand = np.nonzero((mirror.parent >= 0) & (mirror.dist > 1e-5))[0]
    pick = rng.choice(cand, size=60, replace=False)
    scale = rng.choice([0.3, 2.5, 40.0], size=60)

    def run():
        tree = HostTree.from_mirror(mirror)
        n = 0
        for v, f in zip(pick[:20], scale[:20]):
            tree.dist[v] = tree.dist[v] * f
            n += update_genome_lists(dev, tree, [int(v)])
        for v, f in zip(pick[20:], scale[20:]):
            tree.dist[v] = tree.dist[v] * f
        n += update_genome_lists(dev, tree, pick[20:].tolist())
        return tree, n

    mark = dev.mark()
    wave, n_wave = run()
    dev.set_tuning(wave_per_item_max=-1)
    lane, n_lane = run()
    dev.set_tuning()
    assert n_wave == n_lane and n_wave > 150
    assert np.array_equal(wave.dist, lane.dist)
    n_cmp = 0
    for attr in ("id_lower", "id_upRight", "id_upLeft", "id_totUp"):
        a, b = getattr(wave, attr), getattr(lane, attr)
        assert np.array_equal(a >= 0, b >= 0), attr
        moved = np.nonzero((a >= 0) & (a != getattr(HostTree.from_mirror(mirror), attr)))[0]
        if len(moved):
            assert dev.download(a[moved]) == dev.download(b[moved]), attr
            n_cmp += len(moved)
    assert n_cmp > 150
    dev.release(mark)


def test_wavefront_wide_kernels_around_their_list_length_limit(monkeypatch):
    """The wavefront-wide walks stage lists of up to 256 entries; longer ones go through the one-lane code on lane 0 of the
    same launch.  A 400-tip tree of divergent samples (~80 differences each: lists on both sides of the limit) takes both routes in the same launches: updatePartials and evaluatePlacement must still be the one-lane
    kernels', bit for bit."""
    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset
    from maple_amd.tree_host import HostTree, update_genome_lists
    from maple_amd.tree_mirror import TreeMirror
    data = make_dataset(n_samples=400, l_ref=29903, seed=9, mean_diffs=80.0, frac_with_n=0.1, frac_ambig=0.05)
    ref_idx, rf = reference_tables(data.ref)
    dev = Device(ref_idx, rf, arena_bytes=1 << 30)
    dev.set_model(Q=Q)
    tips = {int(v): tip_genome_list(dl, ref_idx) for v, dl in zip(data.tip_node, data.diffs)}
    mirror = TreeMirror(dev, data.parent, data.blen, tips).build()
    n_ent = dev.sizes(mirror.tot_up[mirror.tot_up >= 0])[0]
    assert (n_ent > 256).sum() > 20 and (n_ent <= 200).sum() > 20, (n_ent.min(), int(np.median(n_ent)), n_ent.max())
    rng = np.random.default_rng(37)
    cand = np.nonzero((mirror.parent >= 0) & (mirror.dist > 1e-5))[0]
    pick = rng.choice(cand, size=40, replace=False)

    def run():
        tree = HostTree.from_mirror(mirror)
        n = 0
        for v in pick:
            tree.dist[v] = tree.dist[v] * 2.5
            n += update_genome_lists(dev, tree, [int(v)])
        return tree, n

    wave, n_wave = run()
    dev.set_tuning(wave_per_item_max=-1)
    lane, n_lane = run()
    dev.set_tuning()
    assert n_wave == n_lane and n_wave > 100
    base = HostTree.from_mirror(mirror)
    for attr in ("id_lower", "id_upRight", "id_upLeft", "id_totUp"):
        a, b = getattr(wave, attr), getattr(lane, attr)
        assert np.array_equal(a >= 0, b >= 0), attr
        moved = np.nonzero((a >= 0) & (a != getattr(base, attr)))[0]
        if len(moved):
            assert dev.download(a[moved]) == dev.download(b[moved]), attr
    nodes = np.nonzero(mirror.parent >= 0)[0]
    par = mirror.parent[nodes]
    up_ids = np.where(mirror.children[par, 0] == nodes, mirror.up_right[par], mirror.up_left[par]).astype(np.int32)
    ok = (mirror.tot_up[nodes] >= 0) & (up_ids >= 0)
    nodes, up_ids = nodes[ok], up_ids[ok]
    q_ids = mirror.lower[rng.choice(np.asarray(data.tip_node), size=len(nodes))].astype(np.int32)
    args = (mirror.tot_up[nodes], mirror.lower[nodes], up_ids, mirror.dist[nodes], q_ids, True, mirror.is_tip[nodes])
    wave4 = dev.evaluate_placement_batch(*args)
    dev.set_tuning(wave_per_item_max=-1)
    lane4 = dev.evaluate_placement_batch(*args)
    dev.set_tuning()
    assert len(wave4) > 500 and np.array_equal(wave4.view(np.uint64), lane4.view(np.uint64))
    # shorten() of single lists (k_shorten_wave for small batches): merged-but-unshortened lists are what it is for
    half = mirror.dist[nodes][:500] / 2
    merged = dev.merge_batch(up_ids[:500], half, False, mirror.lower[nodes][:500], half, mirror.is_tip[nodes][:500], True)
    merged = merged[merged >= 0]
    assert len(merged) > 300
    sw = dev.shorten_batch(merged)
    dev.set_tuning(wave_per_item_max=-1)
    sl = dev.shorten_batch(merged)
    dev.set_tuning()
    assert dev.download(sw) == dev.download(sl)
    assert (dev.sizes(sw)[0] < dev.sizes(merged)[0]).sum() > 50           # (it did shorten something)
    dev.close()


def test_serial_placement_through_tree_patch_on_a_tree_with_local_references(world):
    """The serial placement phase on a tree with MAT local references, in every model mode: 20 samples one after the other
    (single-query search, a new internal node + tip at the best branch, maple_update_partials, HostTree.sync =
    maple_tree_patch) must give, sample for sample, the same search results as the same sequence with a full
    maple_tree_upload after every sample -- the patched copy of the tree, its reference frames and its candidate / leaf
    columns are the rebuilt ones.  (The tree edit is a stand-in for placeSampleOnTree; the reference's own edits are
    replayed by test_online_sample_additions_through_tree_patch on trees without local references.)"""
    from maple_amd.host import tip_genome_list
    from maple_amd.mat import add_local_references
    from maple_amd.synth import perturb_diffs
    from maple_amd.tree_host import HostTree, update_genome_lists
    mode, data, dev, orc, mirror = world
    l_ref = dev.lRef
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=ll, thresholdLogLKconsecutivePlacement=1.0)
    prng = np.random.default_rng(23)
    from maple_amd.host import reference_tables
    ref_idx, _ = reference_tables(data.ref)
    new = [tip_genome_list(perturb_diffs(data.diffs[i], data.ref, prng), ref_idx) for i in range(20)]

    def run(patching):
        mark = dev.mark()
        tree = HostTree.from_mirror(mirror)
        assert add_local_references(dev, tree, 30) > 10
        tree.upload_topology(dev)
        results, patched = [], []
        for lst in new:
            patched.append(tree.sync(dev) if patching else (tree.upload_topology(dev), -1)[1])
            dev.placement_prepare(**pkw)
            qid = dev.upload([lst])
            out = dev.placement_search_batch(qid, **pkw)
            assert out["status"][0] >= 0
            results.append((int(out["status"][0]), int(out["bestNode"][0]), float(out["bestScore"][0]), tuple(out["blen"][0].tolist())))
            b = int(out["bestNode"][0])
            if out["status"][0] != 0 or tree.up[b] is None:
                continue
            top, bottom, app = (float(x) for x in out["blen"][0])
            n = tree.n
            g, p, s = tree.up[b], n, n + 1
            up = list(tree.up) + [g, p]
            children = [list(c) for c in tree.children] + [[b, s], []]
            children[g] = [p if c == b else c for c in children[g]]
            up[b] = p
            dist = list(tree.dist) + [top, app]
            dist[b] = bottom
            changed = tree.apply_topology(tree.root, up, children, dist, list(tree.n_minor) + [0, 0])
            tree.id_lower[s] = int(out["bestDiffs"][0])                  # the sample in the frame of its new place
            update_genome_lists(dev, tree, changed)
        dev.release(mark)
        return results, patched
    a, patched = run(True)
    b, _ = run(False)
    assert patched[0] == 0 and all(0 <= p < 64 for p in patched), patched
    assert len(a) == len(b) == 20
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[:2] == y[:2] and rel(x[2], y[2]) < 1e-12 and all(rel(u, v) < 1e-12 for u, v in zip(x[3], y[3])), (k, x, y)
    assert sum(r[0] == 0 for r in a) >= 10


def test_wavefront_wide_append_on_tree_lists(world):
    """wave_append on the lists of a real tree (probVectTotUp of 4 000 branches against tip lists, every model mode; lists
    near the root are longer than what the cooperative walk stages and take its one-lane fall-back): identical to
    maple_append_batch in every bit."""
    mode, data, dev, orc, mirror = world
    rng = np.random.default_rng(41)
    cand = np.nonzero(mirror.tot_up >= 0)[0]
    tips = np.nonzero(mirror.is_tip.astype(bool))[0]
    pl = mirror.tot_up[rng.choice(cand, size=4000)]
    cl = mirror.lower[rng.choice(tips, size=4000)]
    bl = np.where(rng.random(4000) < 0.2, 0.0, 1.0 / dev.lRef)          # some zero-length attachments: -inf results
    one = dev.append_batch(pl, cl, True, bl)
    # (the wavefront-wide kernel's entry point is a hook of libmaple_hip_debug.so: the same lists on a device of that library)
    from maple_amd.host import reference_tables
    from maple_amd.runtime import Device
    ref_idx, rf = reference_tables(data.ref)
    dbg = Device(ref_idx, rf, arena_bytes=1 << 30, debug=True)
    dbg.set_model(**world_model_kwargs(mode, len(ref_idx)))
    pl_d, cl_d = dbg.upload_packed(dev.download_packed(pl)), dbg.upload_packed(dev.download_packed(cl))
    assert np.array_equal(dbg.append_batch(pl_d, cl_d, True, bl), one)
    wave, ms = dbg.debug_wave_append_batch(pl_d, cl_d, True, bl)
    dbg.close()
    assert np.array_equal(one, wave), int((one != wave).sum())
    assert np.isfinite(one).any() and (mode == "siteerr" or np.isinf(one).any())   # (with an error model a zero-length mismatch is finite)
    print(f"{mode}: 4000 wavefront-wide appendProbNode in {ms:.3f} ms")


def test_batch_kernel_with_queries_longer_than_the_lds_stage():
    """k_append_queries keeps the tile's query words in LDS up to 192 entries and reads longer lists from memory: both
    paths against the oracle (samples with ~150 and ~400 differences give lists on either side of the limit)."""
    import torch
    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset
    from oracle.oracle_py import Oracle
    data = make_dataset(n_samples=96, l_ref=29903, seed=12, mean_diffs=30.0, frac_with_n=0.2, frac_ambig=0.2)
    long_a = make_dataset(n_samples=4, l_ref=29903, seed=13, mean_diffs=150.0)
    long_b = make_dataset(n_samples=4, l_ref=29903, seed=14, mean_diffs=400.0)
    ref_idx, rf = reference_tables(data.ref)
    dev = Device(ref_idx, rf, arena_bytes=256 << 20)
    dev.set_model(Q)
    orc = Oracle(ref_idx, rf)
    orc.set_model(Q)
    cands = [tip_genome_list(dl, ref_idx) for dl in data.diffs]
    # the long samples were drawn against their own random references: re-use only their positions / bases on this one
    queries = []
    for ds in (long_a, long_b):
        for dl in ds.diffs:
            fixed = [(("acgt".replace(data.ref[e[1] - 1], ""))[e[1] % 3], e[1]) for e in dl if e[0] in "acgt"]
            queries.append(tip_genome_list(sorted(set(fixed), key=lambda e: e[1]), ref_idx))
    n_ent = [len(q) for q in queries]
    assert min(n_ent) < 192 < max(n_ent), n_ent
    c_ids = dev.upload(cands)
    q_ids = dev.upload(queries)
    cu = torch.device("cuda", 0)
    t_q = torch.from_numpy(q_ids.astype(np.int32)).to(cu)
    t_c = torch.from_numpy(c_ids.astype(np.int32)).to(cu)
    out = torch.empty(len(queries) * len(cands), dtype=torch.float64, device=cu)
    torch.cuda.synchronize()
    dev.append_queries_dev(len(queries), t_q.data_ptr(), len(cands), t_c.data_ptr(), True, 1.0 / dev.lRef, out.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(len(queries), len(cands))
    for qi, q in enumerate(queries):
        for ci in range(0, len(cands), 7):
            want = orc.appendProbNode(cands[ci], q, True, 1.0 / dev.lRef)
            g = float(got[qi, ci])
            assert (math.isinf(want) and math.isinf(g)) or rel(g, want) < 1e-12, (qi, ci, g, want)
    dev.close()


def test_root_frame_candidate_copies_do_not_pile_up_in_the_arena():
    """A tree with local references: the whole-tree searches' rows are made in the ROOT's frame, from copies of every candidate list
    re-expressed there once per tree (ensure_cand_root, spr_batch.hip).  maple_tree_patch / maple_tree_upload drop the copies'
    validity; the next search makes new ones.  The stale ones must go when they are still the arena's last lists -- a loop of
    rounds and patches used to gain n_scored lists per turn -- and the rounds' results must not change."""
    import bench
    from maple_amd.mat import add_local_references
    from maple_amd.tree_host import HostTree
    data, dev, orc, m = build(3000, "unrest", seed=6)
    ht = HostTree.from_mirror(m)
    assert add_local_references(dev, ht, 50) > 20
    dist = np.asarray([float(x or 0.0) for x in ht.dist])
    cols = (m.parent, m.children[:, 0], m.children[:, 1], dist, m.is_tip, ht.id_lower, ht.id_upRight, ht.id_upLeft, ht.id_totUp)
    dev.upload_tree(ht.root, *cols, ht.id_mut)
    kw = bench.search_kwargs(dev.lRef)
    nodes = bench.preorder_nodes(m)
    first, counts, mut_ids = None, [], []
    before = dev.stats()["n_lists"]
    touched = np.asarray(nodes[:8], dtype=np.int32)
    for turn in range(4):
        g = dev.spr_search_batch(nodes, **kw)
        if first is None:
            first = g
        for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
            assert np.array_equal(g[k], first[k]), (turn, k)
        counts.append(dev.stats()["n_lists"])
        # a MAT mutation list uploaded between the rounds (as a caller does for a new reference branch) must survive the release of
        # the stale copies: an arena mark also carries the count of mutation lists, ensure_cand_root must not release through it
        mut_ids.append(int(dev.upload_mutations([[(10 + turn, 0, 1), (500 + turn, 2, 3)]])[0]))
        # the same records again: nothing changes but the library's tables are stale
        dev.tree_patch(m.n_nodes, touched, *[np.asarray(c)[touched] for c in cols])
    # every mutation list uploaded on the way is still there and still its own (pass a list through each: position 10+turn A->C)
    lst = dev.upload([[(4, dev.lRef)]])
    for turn, mid in enumerate(mut_ids):
        assert mid == mut_ids[0] + turn, mut_ids
        out = dev.download(dev.pass_branch_batch(lst, [mid], False))[0]
        # R up to the first mutated site, the site as an explicit nucleotide, R up to the second, the site, R to the end
        assert [e[0] for e in out] == [4, 0, 4, 2, 4] and [e[1] for e in out[::2]] == [9 + turn, 499 + turn, dev.lRef], (turn, out)
    assert (first["status"] == 0).sum() > 1000
    assert counts[0] - before > 1000, (before, counts)               # (the copies were made: one per scored branch)
    assert counts[0] == counts[1] == counts[2] == counts[3], counts  # (... and replaced, not added to, after every patch)
    dev.close()


def test_bench_two_ranks_plumbing():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank) on a box with ONE GPU:
    MAPLE_BENCH_BACKEND=gloo lets both ranks share it and sends the collectives through host memory.  Checks the N > 1
    path end to end: pre-order sharding of each step's batch, the all-gather of proposed moves, the max-over-ranks
    timing and the sum of the ranks' candidate placements."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--samples", "1500", "--model", "unrest", "--batch", "600", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--no-extras"]
    env = dict(os.environ, MAPLE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    line2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                         timeout=900, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line1["n_gpus"] == 1 and line2["scaling"] == "strong"
    # the same searches, whoever ran them: the same total of candidate placements
    assert line2["config"]["searches_timed"] == line1["config"]["searches_timed"] == 2 * 600
    assert line2["config"]["candidate_placements_timed"] == line1["config"]["candidate_placements_timed"]
    assert line2["value"] > 0 and line2["roofline"]["frac"] >= 0


def test_bench_two_ranks_rccl():
    """The same two-rank run over RCCL (backend "nccl"), one GPU per rank -- only where two GPUs are visible (the boxes of this
    project's rounds have one: skipped there; the gloo form above is what runs).  Proposal records then stay in device memory
    through the all-gather (maple_amd.parallel.gather_proposals with a cuda device)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the RCCL form of the two-rank run needs two")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--samples", "1500", "--model", "unrest", "--batch", "600", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MAPLE_BENCH_BACKEND", None)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    line2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                         timeout=900, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["config"]["searches_timed"] == line1["config"]["searches_timed"]
    assert line2["config"]["candidate_placements_timed"] == line1["config"]["candidate_placements_timed"]


def test_apply_phase_batched_equals_sequential(world):
    """applySPRMovesParallel (M:9470-9484) in its two drivers (maple_amd/spr_apply.py): the proposed moves of a deep round on a
    tree with misplaced tips, re-searched one at a time on the current tree and applied -- against the same moves re-searched
    32 at a time, a speculative result kept only while nothing its search may have read was touched.  Same applied sequence,
    same topology, same branch lengths, and every genome list the same entry for entry."""
    import bench
    from maple_amd.spr_apply import SprApplier
    mode, data, dev, orc, mirror = world
    kw = bench.search_kwargs(dev.lRef)
    m = mirror
    # misplace 60 tips: swap the lower lists of pairs of tips far apart in the tree, then rebuild the lists
    from maple_amd.tree_mirror import TreeMirror
    rng = np.random.default_rng(17)
    tips = np.asarray(data.tip_node)
    pick = rng.choice(tips, size=120, replace=False)
    mark = dev.mark()
    tip_lists = {int(v): dev.download([m.lower[v]])[0] for v in tips}
    for a, b in zip(pick[:60], pick[60:]):
        tip_lists[int(a)], tip_lists[int(b)] = tip_lists[int(b)], tip_lists[int(a)]
    m2 = TreeMirror(dev, m.parent, m.dist.copy(), tip_lists).build()
    no_mut = -np.ones(m2.n_nodes, dtype=np.int32)
    dev.upload_tree(m2.root, m2.parent, m2.children[:, 0], m2.children[:, 1], m2.dist, m2.is_tip, m2.lower, m2.up_right, m2.up_left,
                    m2.tot_up, no_mut)
    order = bench.preorder_nodes(m2)
    r = dev.spr_search_batch(order, **kw)
    prop = np.nonzero(r["placement"] >= 0)[0]
    prop = prop[np.argsort(-r["improvement"][prop], kind="stable")]
    moves = order[prop][:80]
    assert len(moves) >= 40
    seq = SprApplier.from_mirror(dev, m2).apply_sequential(moves, kw)
    bat = SprApplier.from_mirror(dev, m2).apply_batched(moves, kw, batch=32)
    assert len(seq.applied) >= 20
    assert seq.applied == bat.applied
    # (searches from misplaced tips expand wide regions, so speculative results are often dropped here; on the bench tree the
    # searches are local and a batch of 32 is mostly kept -- bench.py's apply_phase block)
    assert len(bat.times["search"]) < len(seq.times["search"]) and max(k for _, k in bat.batches) >= 2
    for a in ("up", "c0", "c1", "dist"):
        assert np.array_equal(getattr(seq, a), getattr(bat, a)), a
    for a in ("lower", "up_right", "up_left", "tot_up"):
        ia, ib = getattr(seq, a), getattr(bat, a)
        assert np.array_equal(ia >= 0, ib >= 0), a
        have = np.nonzero(ia >= 0)[0]
        assert dev.download(ia[have]) == dev.download(ib[have]), a
    dev.release(mark)


def test_deep_round_with_error_model_at_20k_tips_frontier_tier_equals_lane_tier():
    """A deep SPR round with the error model on a 20 000-tip tree: a fifth of the searches is long, the frontier tier's pools run
    over on the first call (searches handed back, the level ranges end at the pools' capacity) and the long searches leave at
    the lane tiers' budget.  Status, node ids, candidate counts, scores and branch lengths equal those of the
    one-lane-per-search kernels, bit for bit -- on the first call and on the next one, whose pools are sized by what the
    first asked for."""
    import bench
    data, dev, orc, m = build(20000, "siteerr", seed=1)
    dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left,
                    m.tot_up, -np.ones(m.n_nodes, dtype=np.int32))
    kw = bench.search_kwargs(dev.lRef)
    nodes = bench.preorder_nodes(m)
    lane = dev.spr_search_batch(nodes, search_tier=1, **kw)
    # (wide_search_budget = 20 000 items per search: more than the pools of a first call hold -- they run over, then grow)
    for budget in (20000, 20000, 0, 0):
        g = dev.spr_search_batch(nodes, wide_search_budget=budget, **kw)
        for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
            assert np.array_equal(g[k], lane[k]), (budget, k)
    assert (g["status"] == 0).sum() > 30000 and g["nAppend"].sum() > 1e7
    # (from the second call on, the nodes whose search ran over the budget went to the dense tier at once -- the library remembers
    # them per tree: maple_tuning.noOverHint switches that off; a hint left by ANOTHER budget is only a hint)
    dev.set_tuning(no_over_hint=True)
    g = dev.spr_search_batch(nodes, **kw)
    dev.set_tuning()
    for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
        assert np.array_equal(g[k], lane[k]), ("no hints", k)
    g = dev.spr_search_batch(nodes[::3], wide_search_budget=300, **kw)
    for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
        assert np.array_equal(g[k], lane[k][::3]), ("hints of another budget", k)
    dev.close()


@pytest.mark.parametrize("mode", ["unrest", "ratevar", "siteerr"])
def test_frontier_tier_on_a_tree_with_local_references_equals_the_lane_tier(mode):
    """Trees as MAPLE makes them carry MAT local references (makeNodeReference, M:8296-8354; here setUpMAT's rule, 50 descendants per
    reference node): a search re-expresses the lists it carries at every reference branch it crosses (M:6844-6847, 7111-7118,
    7148-7155, 7359-7366, 7388-7395).  The frontier tier (items carry the removed list of their frame; a list that leaves a frame goes
    through passGenomeListThroughBranch) against the one-lane-per-search kernels on a 10 000-tip tree, every search of a deep
    round: status, node ids, moves, candidate counts, scores and branch lengths bit for bit -- with and without the hand-over of
    long searches to the dense tier."""
    import bench
    from maple_amd.mat import add_local_references
    from maple_amd.tree_host import HostTree
    data, dev, orc, m = build(10000, mode, seed=4)
    ht = HostTree.from_mirror(m)
    n_ref = add_local_references(dev, ht, 50)
    assert n_ref > 100
    dist = np.asarray([float(x or 0.0) for x in ht.dist])
    dev.upload_tree(ht.root, m.parent, m.children[:, 0], m.children[:, 1], dist, m.is_tip, ht.id_lower, ht.id_upRight, ht.id_upLeft,
                    ht.id_totUp, ht.id_mut)
    kw = bench.search_kwargs(dev.lRef)
    nodes = bench.preorder_nodes(m)
    for budget in (-1, 0):
        lane = dev.spr_search_batch(nodes, search_tier=1, wide_search_budget=budget, **kw)
        for _ in range(2):                                              # (the second call: pools sized by the first)
            g = dev.spr_search_batch(nodes, wide_search_budget=budget, **kw)
            for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
                assert np.array_equal(g[k], lane[k]), (budget, k, int((g[k] != lane[k]).sum()))
    assert (g["status"] == 0).sum() > 15000 and g["nAppend"].sum() > 1e7
    if mode == "unrest":
        # bestRemovedPartials (M:7720: the removed list in the best branch's frame, as the reference's in-place shorten() left it):
        # the two tiers hand back the same list, entry for entry -- also where a list was shortened on the way and re-expressed
        # from its shortened form afterwards (k_fr_replay hands such a search to the one-lane kernel: the event item's own
        # children included since round 6)
        mark = dev.mark()
        a = dev.spr_search_batch(nodes, want_removed_partials=True, **kw)
        la = dev.download(a["removedPartials"][a["removedPartials"] >= 0])
        have_a = a["removedPartials"] >= 0
        dev.release(mark)
        b = dev.spr_search_batch(nodes, search_tier=1, want_removed_partials=True, **kw)
        assert np.array_equal(have_a, b["removedPartials"] >= 0) and have_a.sum() > 1000
        lb = dev.download(b["removedPartials"][b["removedPartials"] >= 0])
        assert la == lb
        dev.release(mark)
    dev.close()


@pytest.mark.parametrize("mode", ["ratevar", "siteerr"])
def test_frontier_levels_by_wavefronts_or_by_lanes_give_the_same_searches(mode):
    """maple_tuning.waveAllBelow only chooses HOW a level of the frontier tier is walked -- a wavefront per item that still updates
    genome lists (wave_update.h: the walks cut along their merge paths) or a lane per item: never, for the small levels only (the
    default), for every level.  Same searches bit for bit, on a plain tree and on the same tree with MAT local references."""
    import bench
    from maple_amd.mat import add_local_references
    from maple_amd.tree_host import HostTree
    data, dev, orc, m = build(3000, mode, seed=8)
    kw = bench.search_kwargs(dev.lRef)
    nodes = bench.preorder_nodes(m)
    no_mut = -np.ones(m.n_nodes, dtype=np.int32)
    ht = HostTree.from_mirror(m)
    for form in ("plain", "local references"):
        if form == "plain":
            dev.upload_tree(m.root, m.parent, m.children[:, 0], m.children[:, 1], m.dist, m.is_tip, m.lower, m.up_right, m.up_left, m.tot_up, no_mut)
        else:
            assert add_local_references(dev, ht, 40) > 20
            dist = np.asarray([float(x or 0.0) for x in ht.dist])
            dev.upload_tree(ht.root, m.parent, m.children[:, 0], m.children[:, 1], dist, m.is_tip, ht.id_lower, ht.id_upRight, ht.id_upLeft,
                            ht.id_totUp, ht.id_mut)
        res = []
        for wab in (-1, 0, 1 << 30):
            dev.set_tuning(wave_all_below=wab)
            res.append(dev.spr_search_batch(nodes, wide_search_budget=-1, **kw))
        dev.set_tuning()
        for r in res[1:]:
            for k in ("status", "bestNode", "placement", "nAppend", "bestScore", "currentLK", "improvement", "blen"):
                assert np.array_equal(r[k], res[0][k]), (form, k)
        assert (res[0]["status"] == 0).sum() > 4000
    dev.close()


def test_local_references_batched_equals_one_by_one():
    """maple_amd.mat.add_local_references (every step batched: what the 1 000 000-tip bench tree needs) against its first
    form, one reference node at a time: the same reference nodes, the same mutation lists, the same four genome lists of
    every node, entry for entry."""
    from maple_amd.mat import add_local_references, add_local_references_one_by_one
    from maple_amd.tree_host import HostTree
    data, dev, orc, m = build(3000, "ratevar", seed=11)
    trees = []
    for fn in (add_local_references_one_by_one, add_local_references):
        ht = HostTree.from_mirror(m)
        n_ref = fn(dev, ht, 40)
        trees.append((n_ref, ht))
    (n1, a), (n2, b) = trees
    assert n1 == n2 and n1 > 20
    assert a.mutations == b.mutations
    assert np.array_equal(np.asarray(a.id_mut) >= 0, np.asarray(b.id_mut) >= 0)
    for col in ("id_lower", "id_upRight", "id_upLeft", "id_totUp"):
        ia, ib = np.asarray(getattr(a, col)), np.asarray(getattr(b, col))
        assert np.array_equal(ia >= 0, ib >= 0), col
        have = np.nonzero(ia >= 0)[0]
        assert dev.download(ia[have]) == dev.download(ib[have]), col
    dev.close()


def test_oracle_tree_log_lk_from_the_tips_equals_the_library_on_plain_and_local_reference_trees(world):
    """bench.tree_log_lk_check, the bench line's `tree_log_lk`: calculateTreeLikelihood (M:9721-9779) by the library over the
    stored lower lists -- of the plain tree and of the same tree with MAT local references -- against the C oracle's value from
    the tips' lists alone (no list of the GPU's enters it)."""
    import bench
    from maple_amd.mat import add_local_references
    from maple_amd.tree_host import HostTree
    mode, data, dev, orc, mirror = world
    tips = np.nonzero(mirror.is_tip)[0]
    tip_ids = mirror.lower.copy()
    from maple_amd.host import reference_tables
    ref_idx, rf = reference_tables(data.ref)
    kw = world_model_kwargs(mode, len(ref_idx))
    plain = bench.tree_log_lk_check(dev, mirror, None, tip_ids, kw, ref_idx, rf)
    assert plain["rel_delta"] <= 1e-12, plain
    mark = dev.mark()
    ht = HostTree.from_mirror(mirror)
    assert add_local_references(dev, ht, 50) > 5
    mat = bench.tree_log_lk_check(dev, mirror, ht, tip_ids, kw, ref_idx, rf)
    # (the reference's whole-genome term, M:4487, uses the root's reference in every frame: the two forms of the tree differ in
    # about the fifth digit, each equal to the oracle's value for its own form)
    assert mat["rel_delta"] <= 1e-11 and 0 < abs(mat["oracle"] - plain["oracle"]) < 1e-3 * abs(plain["oracle"]), (mat, plain)
    dev.release(mark)
    assert len(tips) * 2 - 1 == mirror.n_nodes


@pytest.mark.parametrize("mode", ["ratevar", "siteerr"])
def test_serial_loop_with_rows_made_ahead_equals_the_plain_loop(mode):
    """maple_placement_ahead (include/maple_hip.h): the score rows of the next samples of the serial placement loop (M:11692-11752)
    made in ONE launch and kept current under maple_tree_patch.  The loop with the samples announced 64 at a time against the
    plain loop (one scoring launch per sample) on a 6 000-tip tree, 400 new samples -- among them copies of tips (minor sequences:
    skipped by the stand-in edit) and samples placed next to samples added before them: every search's status, node, score,
    branch lengths and candidate count bit for bit, the final trees identical (relatives, lengths), the lists of a sample of the
    final tree's nodes entry for entry; and the rows were really used (the library counts its refreshes)."""
    import bench
    from maple_amd.host import tip_genome_list
    from maple_amd.synth import perturb_diffs
    data, dev, orc, m = build(6000, mode, seed=9)
    l_ref = dev.lRef
    ll = math.log(l_ref)
    pkw = dict(oneMutBLen=1.0 / l_ref, effectivelyNon0BLen=1.0 / (10 * l_ref), thresholdLogLK=18.0 * ll,
               thresholdLogLKoptimization=1.0 * ll, thresholdLogLKconsecutivePlacement=1.0)
    kw = world_model_kwargs(mode, l_ref, 9)
    tip_kw = dict(error_rates=kw["errorRates"]) if mode == "siteerr" else {}
    prng = np.random.default_rng(3)
    src = prng.choice(len(data.diffs), size=400, replace=True)            # (with replacement: several samples next to one tip)
    new_lists = []
    for k, i in enumerate(src):
        dl = data.diffs[int(i)] if k % 25 == 0 else perturb_diffs(data.diffs[int(i)], data.ref, prng)
        new_lists.append(tip_genome_list(dl, dev.ref_idx, **tip_kw))
    runs, stats = [], []
    # plain loop; rows by expansion; every branch; expansion cut short; rows by expansion without the traversals made ahead
    for ahead, expansion, spec in ((0, 0, True), (64, 0, True), (64, 1, True), (64, 2, True), (64, 0, False)):
        mark = dev.mark()
        dev.set_tuning(no_ahead_expansion=expansion, no_ahead_speculation=not spec)
        before = dev.placement_ahead_stats()
        sp = bench.serial_phase(dev, m, new_lists, pkw, ahead=ahead)
        after = dev.placement_ahead_stats()
        stats.append({k: after[k] - before[k] for k in after})
        c = sp["cols"]
        n = c["n"]
        probe = np.arange(0, n, 97)
        lists = {name: dev.download(c[name][probe][c[name][probe] >= 0]) for name in ("lower", "tot_up", "up_right")}
        runs.append((sp, lists))
        dev.release(mark)
    dev.set_tuning()
    (a, la) = runs[0]
    assert a["placed"] > 300 and len(a["times"]["ahead"]) == 0 and stats[0]["searches"] == 0
    for (b, lb), st in zip(runs[1:], stats[1:]):
        assert a["placed"] == b["placed"]
        assert a["results"] == b["results"]
        ca, cb = a["cols"], b["cols"]
        assert ca["n"] == cb["n"]
        for name in ("up", "c0", "c1", "dist", "tip"):
            assert np.array_equal(ca[name][: ca["n"]], cb[name][: cb["n"]]), name
        for name in la:
            assert la[name] == lb[name], name
        assert len(b["times"]["ahead"]) >= 6 and st["searches"] == len(new_lists), st
    # rows by expansion: items were expanded, far fewer than samples x branches, and few rows had to be scored in full after all
    assert stats[1]["expanded_items"] > 0 and stats[2]["expanded_items"] == 0
    assert stats[1]["expanded_items"] < 0.5 * len(new_lists) * m.n_nodes
    assert stats[1]["fallbacks"] <= 0.2 * len(new_lists), stats[1]
    assert stats[2]["fallbacks"] == 0
    # ... and with the expansion cut short after six levels nearly every search has its row scored in full after all: same results
    assert stats[3]["fallbacks"] > 0.5 * len(new_lists), stats[3]
    # the traversal of the next sample made by a host thread during the placement before it: used where the placement touched
    # nothing it had visited, dropped (and made again) where it did -- and never made when switched off
    assert stats[1]["traversals_ahead_used"] > 0.3 * len(new_lists), stats[1]
    assert stats[1]["traversals_ahead_used"] + stats[1]["traversals_ahead_dropped"] <= len(new_lists)
    assert stats[4]["traversals_ahead_used"] == 0 and stats[4]["traversals_ahead_dropped"] == 0
    print("rows made ahead:", stats)
    dev.close()
