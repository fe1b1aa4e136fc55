"""GPU parity: the HIP kernels (through the C ABI) against the reference's recorded answers
(tests/golden) and against the C oracle on the same inputs.

Bar: integer structure (entry types, positions, reference nucleotides, tuple lengths, flags,
None / False / -inf outcomes) bit-exact; log-likelihoods, branch lengths and partial vectors
within 1e-9 relative of the reference (north_star allows 1e-6).
"""
import math

import numpy as np
import pytest

from golden_util import close, fixture_names, lists_match, load, model_args, ref_indices, tup

pytestmark = pytest.mark.gpu
REL = 1e-9
FIXTURES = fixture_names()


def make_device(f, debug=False):
    """debug: libmaple_hip_debug.so (the product library plus the test hooks of include/maple_hip_debug.h)"""
    from maple_amd.runtime import Device
    ctx = f["context"]
    return Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"],
                  minBLenSensitivity=ctx["minBLenSensitivity"], thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"],
                  thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"], defaultBLen=ctx["defaultBLen"],
                  arena_bytes=256 << 20, debug=debug)


def make_oracle(f):
    from oracle.oracle_py import Oracle
    ctx = f["context"]
    return Oracle(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"],
                  minBLenSensitivity=ctx["minBLenSensitivity"], thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"],
                  thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"], defaultBLen=ctx["defaultBLen"])


def by_model(f, fn):
    groups = {}
    for rec in f["calls"][fn]:
        if rec.get("raised"):
            continue
        groups.setdefault(rec["model"], []).append(rec)
    return groups


@pytest.fixture(scope="module", params=FIXTURES)
def env(request):
    f = load(request.param)
    dev = make_device(f)
    yield f, dev, make_oracle(f)
    dev.close()


def test_model_tables(env):
    f, dev, o = env
    for mod in f["models"]:
        dev.set_model(**model_args(mod))
        o.set_model(**model_args(mod))
        cr, ce, te = dev.get_model()
        assert np.array_equal(cr, o.cr)
        lRef = dev.lRef
        probe = mod.get("cumulativeRate_probe")
        if probe:
            assert [cr[1], cr[lRef // 2], cr[lRef]] == probe
        if mod["usingErrorRate"]:
            assert te == mod["totError"]


def test_appendProbNode(env):
    f, dev, o = env
    total = 0
    for mid, recs in by_model(f, "appendProbNode").items():
        dev.set_model(**model_args(f["models"][mid]))
        o.set_model(**model_args(f["models"][mid]))
        mark = dev.mark()
        ids = dev.upload([tup(r["P"]) for r in recs] + [tup(r["C"]) for r in recs])
        n = len(recs)
        got = dev.append_batch(ids[:n], ids[n:], [r["isTipC"] for r in recs], [r["bLen"] for r in recs])
        dev.release(mark)
        for g, r in zip(got, recs):
            assert close(float(g), r["ret"], REL), (g, r["ret"])
            assert close(float(g), o.appendProbNode(tup(r["P"]), tup(r["C"]), r["isTipC"], r["bLen"]), REL)
        total += n
    assert total > 100


def test_wavefront_wide_appendProbNode_is_the_one_lane_walk_bit_for_bit(env):
    """wave_append (maple_amd/csrc/wave_dev.h: the walk cut along its merge path, all steps' factors at once, the running
    product in walk order) against the one-lane walk of maple_append_batch on every recorded appendProbNode call of the
    fixture, in every model mode: identical doubles (including -inf), not just close ones."""
    f, dev, o = env
    dbg = make_device(f, debug=True)              # (the hook lives in libmaple_hip_debug.so; the one-lane walk is the product library's)
    total = 0
    for mid, recs in by_model(f, "appendProbNode").items():
        dev.set_model(**model_args(f["models"][mid]))
        dbg.set_model(**model_args(f["models"][mid]))
        mark, mark_d = dev.mark(), dbg.mark()
        n = len(recs)
        lists = [tup(r["P"]) for r in recs] + [tup(r["C"]) for r in recs]
        ids, ids_d = dev.upload(lists), dbg.upload(lists)
        tips, bls = [r["isTipC"] for r in recs], [r["bLen"] for r in recs]
        one = dev.append_batch(ids[:n], ids[n:], tips, bls)
        wave, _ = dbg.debug_wave_append_batch(ids_d[:n], ids_d[n:], tips, bls)
        assert np.array_equal(one, wave), [(a, b) for a, b in zip(one, wave) if a != b][:3]
        # ... and with the roles of the two lists exchanged (other merge paths, other ties)
        one = dev.append_batch(ids[n:], ids[:n], tips, bls)
        wave, _ = dbg.debug_wave_append_batch(ids_d[n:], ids_d[:n], tips, bls)
        assert np.array_equal(one, wave)
        dev.release(mark)
        dbg.release(mark_d)
        total += n
    dbg.close()
    assert total > 100


def test_lists_update_keeps_ids_and_changes_contents(env):
    """maple_lists_update: existing ids get new contents -- shorter ones in place, longer ones in fresh room --
    and every operator sees the new words under the old id."""
    f, dev, o = env
    mid, recs = next(iter(by_model(f, "appendProbNode").items()))
    recs = recs[:40]
    dev.set_model(**model_args(f["models"][mid]))
    o.set_model(**model_args(f["models"][mid]))
    mark = dev.mark()
    n = len(recs)
    ids = dev.upload([tup(r["P"]) for r in recs] + [tup(r["C"]) for r in recs])
    before = dev.stats()["n_lists"]
    # the parent lists are replaced by the parent lists of the NEXT record (other lengths: some fit, some do not)
    rolled = recs[1:] + recs[:1]
    dev.update_lists(ids[:n], [tup(r["P"]) for r in rolled])
    assert dev.stats()["n_lists"] == before
    assert [len(x) for x in dev.download(ids[:n])] == [len(tup(r["P"])) for r in rolled]
    got = dev.append_batch(ids[:n], ids[n:], [r["isTipC"] for r in recs], [r["bLen"] for r in recs])
    for g, r, rp in zip(got, recs, rolled):
        want = o.appendProbNode(tup(rp["P"]), tup(r["C"]), r["isTipC"], r["bLen"])
        assert close(float(g), want, REL), (g, want)
    dev.release(mark)


def test_mergeVectors(env):
    f, dev, o = env
    total = 0
    for mid, recs in by_model(f, "mergeVectors").items():
        dev.set_model(**model_args(f["models"][mid]))
        for want_lk in (False, True):
            sub = [r for r in recs if r["returnLK"] == want_lk]
            if not sub:
                continue
            mark = dev.mark()
            ids = dev.upload([tup(r["pv1"]) for r in sub] + [tup(r["pv2"]) for r in sub])
            n = len(sub)
            res = dev.merge_batch(ids[:n], [r["b1"] or 0.0 for r in sub], [r["tip1"] for r in sub], ids[n:],
                                  [r["b2"] or 0.0 for r in sub], [r["tip2"] for r in sub],
                                  [r["isUpDown"] for r in sub], returnLK=want_lk,
                                  numMinor1=[r["numMinor1"] for r in sub], numMinor2=[r["numMinor2"] for r in sub])
            out, lk = res if want_lk else (res, None)
            lists = dev.download(out)
            dev.release(mark)
            for k, r in enumerate(sub):
                want = r["ret"]
                if want_lk:
                    assert lists_match(lists[k], tup(want[0]), REL)
                    assert close(float(lk[k]), want[1], REL), (lk[k], want[1])
                else:
                    assert lists_match(lists[k], tup(want), REL), (lists[k], want)
            total += n
    assert total > 100


def test_estimateBranchLength(env):
    f, dev, o = env
    for mid, recs in by_model(f, "estimateBranchLengthWithDerivative").items():
        dev.set_model(**model_args(f["models"][mid]))
        mark = dev.mark()
        ids = dev.upload([tup(r["P"]) for r in recs] + [tup(r["C"]) for r in recs])
        n = len(recs)
        t, isf = dev.blen_batch(ids[:n], ids[n:], [r["fromTipC"] for r in recs])
        dev.release(mark)
        for k, r in enumerate(recs):
            if r["ret"] is False:
                assert isf[k]
            else:
                assert not isf[k] and close(float(t[k]), r["ret"], REL), (t[k], r["ret"])


def test_evaluatePlacement(env):
    f, dev, o = env
    for mid, recs in by_model(f, "evaluatePlacement").items():
        dev.set_model(**model_args(f["models"][mid]))
        mark = dev.mark()
        n = len(recs)
        ids = dev.upload([tup(r[k]) for k in ("midTot", "downVect", "upVect", "removedPartials") for r in recs])
        out = dev.evaluate_placement_batch(ids[:n], ids[n:2 * n], ids[2 * n:3 * n], [r["distance"] for r in recs],
                                           ids[3 * n:], [r["isRemovedTip"] for r in recs],
                                           [r["fromTip1"] for r in recs])
        dev.release(mark)
        for k, r in enumerate(recs):
            want = [0.0 if w is False else w for w in r["ret"]]
            assert all(close(float(g), w, 1e-8) for g, w in zip(out[k], want)), (out[k], want)


def test_rootVector(env):
    f, dev, o = env
    for mid, recs in by_model(f, "rootVector").items():
        dev.set_model(**model_args(f["models"][mid]))
        mark = dev.mark()
        ids = dev.upload([tup(r["pv"]) for r in recs])
        paths = [dev.upload_mutations(r["pathMutations"]) for r in recs]
        out = dev.root_vector_batch(ids, [r["bLen"] or 0.0 for r in recs], [r["isFromTip"] for r in recs], paths)
        lists = dev.download(out)
        dev.release(mark)
        for k, r in enumerate(recs):
            assert lists_match(lists[k], tup(r["ret"]), REL), (lists[k], r["ret"])


def _u_groups(f, fn):
    g = {}
    for rec in f["calls"][fn]:
        g.setdefault(bool(rec["usingErrorRate"]), []).append(rec)
    return g


def test_structural_functions(env):
    f, dev, o = env
    Q = f["models"][0]["Q"]
    for u, recs in _u_groups(f, "passGenomeListThroughBranch").items():
        dev.set_model(Q, usingErrorRate=u, errorRateGlobal=1e-4)
        mark = dev.mark()
        ids = dev.upload([tup(r["pv"]) for r in recs])
        mids = dev.upload_mutations([r["mutations"] for r in recs])
        lists = dev.download(dev.pass_branch_batch(ids, mids, [r["dirIsUp"] for r in recs]))
        dev.release(mark)
        for k, r in enumerate(recs):
            assert lists_match(lists[k], tup(r["ret"]), 0.0), (lists[k], r["ret"])
    for u, recs in _u_groups(f, "shorten").items():
        dev.set_model(Q, usingErrorRate=u, errorRateGlobal=1e-4)
        mark = dev.mark()
        lists = dev.download(dev.shorten_batch(dev.upload([tup(r["vec"]) for r in recs])))
        dev.release(mark)
        for k, r in enumerate(recs):
            assert lists_match(lists[k], tup(r["ret"]), 0.0)
    for u, recs in _u_groups(f, "areVectorsDifferent").items():
        recs = [r for r in recs if r["pv2"] is not None]
        dev.set_model(Q, usingErrorRate=u, errorRateGlobal=1e-4)
        mark = dev.mark()
        n = len(recs)
        ids = dev.upload([tup(r["pv1"]) for r in recs] + [tup(r["pv2"]) for r in recs])
        got = dev.differ_batch(ids[:n], ids[n:])
        dev.release(mark)
        assert [bool(g) for g in got] == [r["ret"] for r in recs]


def test_getPartialVec_and_simplify_on_the_gpu(env):
    """a3 / a4 directly on the HIP path: every recorded getPartialVec (M:4073-4141) and simplify (M:3697-3717) call of
    the reference through the one-lane-per-call hooks, plus the negative-clamp exits (M:4096, 4106, 4122, 4139: any
    component that goes negative turns the whole vector into [0.25] * 4) on inputs built to reach each of them."""
    f, _, o = env
    dev = make_device(f, debug=True)              # (the hooks of include/maple_hip_debug.h: the same device functions the product library is built from)
    Q = f["models"][0]["Q"]
    n_calls = 0
    for u, recs in _u_groups(f, "getPartialVec").items():
        dev.set_model(Q, usingErrorRate=u, errorRateGlobal=1e-4)
        got = dev.debug_gpv_batch([r["i12"] for r in recs], [r["totLen"] or 0.0 for r in recs], [r["mutMatrix"] for r in recs],
                                  [r["errorRate"] or 0.0 for r in recs],
                                  [r["vect"] if r["vect"] is not None else [0.0] * 4 for r in recs],
                                  [bool(r["upNode"]) for r in recs], [bool(r["flag"]) for r in recs])
        for g, r in zip(got, recs):
            assert all(close(float(a), b, REL, 0.0) for a, b in zip(g, r["ret"])), (g, r["ret"])
        n_calls += len(recs)
    assert n_calls > 50
    dev.set_model(Q)
    recs = f["calls"]["simplify"]
    got = dev.debug_simplify_batch([r["vec"] for r in recs], [r["refA"] for r in recs])
    assert [int(g) for g in got] == [r["ret"] for r in recs]
    # the clamp exits: a matrix with a strongly negative diagonal and a long branch drives a component below zero
    M = [[-40.0, 10.0, 20.0, 10.0], [5.0, -30.0, 5.0, 20.0], [10.0, 10.0, -50.0, 30.0], [1.0, 2.0, 3.0, -6.0]]
    cases = [(6, 0.5, [0.7, 0.1, 0.1, 0.1], False, False, 0.0),      # O vector, downward (M:4106)
             (6, 0.5, [0.7, 0.1, 0.1, 0.1], True, False, 0.0),       # O vector, upward (M:4096)
             (0, 0.5, None, False, False, 0.0),                      # one-hot, the observed nucleotide's entry (M:4139)
             (2, 0.1, None, True, False, 0.0),                       # one-hot, upward
             (0, 0.5, None, False, True, 0.01),                      # error-smeared one-hot (M:4122), only with an error model
             (1, 1e-3, None, False, False, 0.0)]                     # and one that stays positive
    for u in (False, True):
        dev.set_model(Q, usingErrorRate=u, errorRateGlobal=1e-4)
        o.set_model(Q, usingErrorRate=u, errorRateGlobal=1e-4)
        got = dev.debug_gpv_batch([c[0] for c in cases], [c[1] for c in cases], [M] * len(cases), [c[5] for c in cases],
                                  [c[2] or [0.0] * 4 for c in cases], [c[3] for c in cases], [c[4] for c in cases])
        n_clamped = 0
        for g, c in zip(got, cases):
            want = o.getPartialVec(c[0], c[1], M, c[5], c[2], c[3], c[4])
            assert [float(x) for x in g] == want, (c, g, want)
            n_clamped += want == [0.25] * 4
        assert n_clamped >= 4
    dev.close()


def test_appendProbNode_log_of_zero_is_minus_infinity(env):
    """Documented deviation: where appendProbNode's running product reaches exactly 0 (an O vector with a zero component
    met by that nucleotide over a zero-length branch) the reference's final math.log(totalFactor) raises ValueError, which
    its callers do not catch; kernel and oracle return -inf, the value the reference itself uses for an impossible
    attachment (M:6663, 6742)."""
    f, dev, o = env
    dev.set_model(f["models"][0]["Q"])
    o.set_model(f["models"][0]["Q"])
    l_ref = dev.lRef
    parent = [(4, 10), (6, int(dev.ref_idx[10]), [0.5, 0.5, 0.0, 0.0] if dev.ref_idx[10] != 2 else [0.5, 0.0, 0.0, 0.5]), (4, l_ref)]
    zero_nuc = 2 if dev.ref_idx[10] != 2 else 1
    child = [(4, 10), (zero_nuc, int(dev.ref_idx[10])), (4, l_ref)]
    mark = dev.mark()
    ids = dev.upload([parent, child])
    got = dev.append_batch([ids[0]], [ids[1]], [False], [0.0])[0]
    dev.release(mark)
    assert got == float("-inf"), got
    assert o.appendProbNode(parent, child, False, 0.0) == float("-inf")


def test_ops_mirror_reads_like_the_reference(env):
    """The same-name host functions (maple_amd.ops) on a few records per function."""
    from maple_amd.ops import GenomeOps
    f, dev, o = env
    ops = GenomeOps(dev)
    for mid, recs in list(by_model(f, "appendProbNode").items())[:2]:
        dev.set_model(**model_args(f["models"][mid]))
        for r in recs[:5]:
            assert close(ops.appendProbNode(tup(r["P"]), tup(r["C"]), r["isTipC"], r["bLen"]), r["ret"], REL)
    for mid, recs in list(by_model(f, "mergeVectors").items())[:2]:
        dev.set_model(**model_args(f["models"][mid]))
        for r in [x for x in recs if not x["returnLK"]][:5]:
            got = ops.mergeVectors(tup(r["pv1"]), r["b1"], r["tip1"], tup(r["pv2"]), r["b2"], r["tip2"],
                                   isUpDown=r["isUpDown"])
            assert lists_match(got, tup(r["ret"]), REL)
    for mid, recs in list(by_model(f, "estimateBranchLengthWithDerivative").items())[:2]:
        dev.set_model(**model_args(f["models"][mid]))
        for r in recs[:5]:
            got = ops.estimateBranchLengthWithDerivative(tup(r["P"]), tup(r["C"]), r["fromTipC"])
            assert (got is False) if r["ret"] is False else close(got, r["ret"], REL)


def test_append_queries_dev_matches_pair_batches(env):
    """The query-major tiled kernel (the bench's headline kernel) against the reference's recorded values:
    every recorded (P, C) pair is scored as query C against the candidate set of ALL recorded P lists."""
    import torch
    f, dev, o = env
    for mid, recs in list(by_model(f, "appendProbNode").items())[:3]:
        mod = f["models"][mid]
        dev.set_model(**model_args(mod))
        recs = [r for r in recs if r["isTipC"]][:40]
        if len(recs) < 4:
            continue
        bl = recs[0]["bLen"]
        mark = dev.mark()
        n = len(recs)
        ids = dev.upload([tup(r["P"]) for r in recs] + [tup(r["C"]) for r in recs])
        cu = torch.device("cuda", 0)
        t_q = torch.from_numpy(ids[n:].astype(np.int32)).to(cu)
        t_c = torch.from_numpy(ids[:n].astype(np.int32)).to(cu)
        t_out = torch.empty(n * n, dtype=torch.float64, device=cu)
        torch.cuda.synchronize()
        dev.append_queries_dev(n, t_q.data_ptr(), n, t_c.data_ptr(), True, bl, t_out.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = t_out.cpu().numpy().reshape(n, n)
        want = dev.append_batch(np.tile(ids[:n], n), np.repeat(ids[n:], n), True, bl).reshape(n, n)
        dev.release(mark)
        for q in range(n):
            for k in range(n):
                assert close(float(got[q, k]), float(want[q, k]), 1e-12), (q, k, got[q, k], want[q, k])
            if recs[q]["bLen"] == bl:
                assert close(float(got[q, q]), recs[q]["ret"], REL)


def test_fused_argmax_equals_argmax_of_the_score_matrix(env):
    """The wavefront-reduction output mode of the batch kernel (maple_append_queries_argmax_dev): per query the best score
    and its candidate, bit-identical to the arg-max of the full score matrix with exact ties going to the smallest visit
    rank; and the RCCL arg-max all-reduce on a one-rank communicator (identity)."""
    import torch
    f, dev, o = env
    mid, recs = max(by_model(f, "appendProbNode").items(), key=lambda kv: len(kv[1]))
    dev.set_model(**model_args(f["models"][mid]))
    mark = dev.mark()
    parents = dev.upload([tup(r["P"]) for r in recs] * 3)                  # duplicates: exact ties to break
    kids = dev.upload([tup(r["C"]) for r in recs[:24]])
    cu = torch.device("cuda", 0)
    t_q = torch.from_numpy(kids.astype(np.int32)).to(cu)
    t_c = torch.from_numpy(parents.astype(np.int32)).to(cu)
    Q, Cn = len(kids), len(parents)
    rank = np.random.default_rng(5).permutation(Cn).astype(np.int32)
    t_rank = torch.from_numpy(rank).to(cu)
    full = torch.empty(Q * Cn, dtype=torch.float64, device=cu)
    best = torch.empty(Q, dtype=torch.float64, device=cu)
    idx = torch.empty(Q, dtype=torch.int32, device=cu)
    torch.cuda.synchronize()
    dev.append_queries_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), False, 1e-4, full.data_ptr(), 0)
    dev.append_queries_argmax_dev(Q, t_q.data_ptr(), Cn, t_c.data_ptr(), t_rank.data_ptr(), False, 1e-4, best.data_ptr(),
                                  idx.data_ptr(), 0)
    torch.cuda.synchronize()
    m = full.cpu().numpy().reshape(Q, Cn)
    for q in range(Q):
        top = m[q].max()
        ties = np.nonzero(m[q] == top)[0]
        want = ties[np.argmin(rank[ties])]
        assert best[q].item() == top and int(idx[q].item()) == int(want), (q, best[q].item(), top, idx[q].item(), want)
        assert len(ties) >= 3 or not np.isfinite(top)
    # RCCL entry point, one rank: the all-reduce must leave (score, idx) as they are
    dev.comm_init(1, 0, dev.comm_unique_id())
    before = (best.clone(), idx.clone())
    dev.argmax_allreduce_dev(Q, best.data_ptr(), idx.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(best, before[0]) and torch.equal(idx, before[1])
    dev.release(mark)


def test_edge_cases_through_the_c_abi():
    """Empty batches, whole-genome N / R lists, bad list ids and an exhausted arena: defined results or a clean error
    (MapleError with the library's message), never a crash."""
    from maple_amd.runtime import Device, MapleError
    from oracle.oracle_py import Oracle
    l_ref = 500
    ref_idx = np.arange(l_ref, dtype=np.uint8) % 4
    rf = [0.3, 0.2, 0.2, 0.3]
    Qm = [[-0.9, 0.2, 0.5, 0.2], [0.3, -1.4, 0.1, 1.0], [0.8, 0.1, -1.3, 0.4], [0.1, 0.6, 0.1, -0.8]]
    dev = Device(ref_idx, rf, arena_bytes=1 << 20)
    dev.set_model(Qm)
    orc = Oracle(ref_idx, rf)
    orc.set_model(Qm)
    # empty batches
    assert len(dev.append_batch([], [], [], [])) == 0
    assert len(dev.merge_batch([], [], [], [], [], [], False)) == 0
    assert len(dev.shorten_batch([])) == 0
    # degenerate lists: everything unknown / everything reference / one site each side
    all_n, all_r = [(5, l_ref)], [(4, l_ref)]
    one = [(4, 249), (2, int(ref_idx[249])), (4, l_ref)] if ref_idx[249] != 2 else [(4, 249), (1, int(ref_idx[249])), (4, l_ref)]
    ids = dev.upload([all_n, all_r, one])
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (2, 1), (2, 2), (0, 2)]
    lists = [all_n, all_r, one]
    got = dev.append_batch([ids[a] for a, b in pairs], [ids[b] for a, b in pairs], True, 1e-3)
    for (a, b), g in zip(pairs, got):
        want = orc.appendProbNode(lists[a], lists[b], True, 1e-3)
        assert (math.isinf(want) and math.isinf(g)) or close(float(g), want, 1e-12), (a, b, g, want)
    merged = dev.download(dev.merge_batch([ids[a] for a, b in pairs], 1e-3, True, [ids[b] for a, b in pairs], 2e-3, True, False))
    for (a, b), g in zip(pairs, merged):
        want = orc.mergeVectors(lists[a], 1e-3, True, lists[b], 2e-3, True)
        assert (want is None and g is None) or lists_match(g, want, 1e-12), (a, b, g, want)
    # a list id that does not exist
    with pytest.raises(MapleError):
        dev.append_batch([999999], [ids[0]], True, 1e-3)
    with pytest.raises(MapleError):
        dev.shorten_batch([-7])
    # arena exhaustion is reported, and the context stays usable after releasing
    mark = dev.mark()
    big = [(4, 1)] + [((k % 3 + 1 + int(ref_idx[k])) % 4, int(ref_idx[k])) for k in range(1, l_ref - 1)] + [(4, l_ref)]
    with pytest.raises(MapleError):
        for _ in range(4000):
            dev.upload([big] * 64)
    dev.release(mark)
    assert close(float(dev.append_batch([ids[1]], [ids[2]], True, 1e-3)[0]), orc.appendProbNode(all_r, one, True, 1e-3), 1e-12)
    dev.close()
