"""CPU-side checks: the C ABI library loads and exports every symbol include/maple_hip.h declares, fails
loudly without a GPU, and the host logic (packing, MAPLE-format reader, tip lists, sharding) is right."""
import ctypes as C
import gzip
import json
import os
import re

import numpy as np
import pytest

from golden_util import GOLDEN, fixture_names, load, tup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="maple_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(maple_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from maple_amd import runtime
    lib = runtime.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/maple_hip.h but not exported"
    assert set(runtime.EXPORTS) <= set(names)
    assert lib.maple_abi_version() == 4
    # the product library is the operator boundary and nothing else: the measurement aids and test hooks of
    # include/maple_hip_debug.h are exported by libmaple_hip_debug.so only, which also has everything above
    dbg_names = declared_symbols("maple_hip_debug.h")
    assert sorted(dbg_names) == sorted(runtime.DEBUG_EXPORTS) and all(n.startswith("maple_debug_") for n in dbg_names)
    assert not [n for n in names if n.startswith("maple_debug_")]
    dbg = runtime.load_library(debug=True)
    for n in dbg_names:
        assert not hasattr(lib, n), f"{n} is exported by the product library"
        assert hasattr(dbg, n), n
    for n in names:
        assert hasattr(dbg, n), n


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from maple_amd.runtime import Device, MapleError
    with pytest.raises(MapleError):
        Device(np.zeros(100, dtype=np.uint8), [0.25] * 4)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "maple_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("C oracle", ""), os.path.join(dirpath, f)


@pytest.mark.parametrize("name", fixture_names())
def test_pack_unpack_roundtrip(name):
    from maple_amd.genome_list import pack_lists, unpack_lists
    f = load(name)
    for u in (False, True):
        lists = []
        for rec in f["calls"]["appendProbNode"]:
            if f["models"][rec["model"]]["usingErrorRate"] == u:
                lists += [tup(rec["P"]), tup(rec["C"])]
        for rec in f["calls"]["mergeVectors"]:
            if f["models"][rec["model"]]["usingErrorRate"] == u and not rec.get("raised"):
                r = rec["ret"][0] if rec["returnLK"] else rec["ret"]
                if r is not None:
                    lists.append(tup(r))
        if not lists:
            continue
        pl = pack_lists(lists, u)
        back = unpack_lists(pl, u)
        assert len(back) == len(lists)
        for a, b in zip(back, lists):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert len(x) == len(y) and x[0] == y[0] and x[1] == y[1]
                assert list(x[2:-1]) == list(y[2:-1]) if x[0] == 6 else tuple(x[2:]) == tuple(y[2:])
                if x[0] == 6:
                    assert list(x[-1]) == list(y[-1])
        # positions are explicit and increasing inside every list
        for i in range(len(pl)):
            p = pl.pos[pl.ent_off[i]:pl.ent_off[i + 1]]
            assert (np.diff(p) > 0).all() and p[-1] == len(f["context"]["ref"])


def _search_fixtures():
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith("search_"))


@pytest.mark.parametrize("fname", _search_fixtures())
def test_tip_lists_and_reader_match_reference(fname):
    from maple_amd.host import read_maple_file, reference_tables, tip_genome_list
    with gzip.open(os.path.join(GOLDEN, fname), "rt") as fh:
        f = json.load(fh)
    if f.get("input", "synth_small.maple.txt") == "synth_small.maple.txt":
        ref, data = read_maple_file(os.path.join(GOLDEN, "synth_small.maple.txt"))
        assert ref == f["context"]["ref"] and len(data) == 160
    else:                              # the reference's own example files are not copied into the repository
        ref = f["context"]["ref"]
    ref_idx, root_freqs = reference_tables(ref, "JC" if "JC" in f["flags"] else "UNREST")
    assert root_freqs == f["context"]["rootFreqs"]
    m = f["model"]
    for rec in f["placements"]:
        diffs = [tuple(e) for e in rec["diffs"]]
        kw = {}
        if m["usingErrorRate"]:
            kw = dict(error_rates=m["errorRates"]) if m["errorRateSiteSpecific"] else dict(error_rate=m["errorRateGlobal"])
        got = tip_genome_list(diffs, ref_idx, **kw)
        want = tup(rec["query"])
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a[0] == b[0] and a[1] == b[1] and len(a) == len(b)
            if a[0] == 6 and not m["usingErrorRate"]:
                # With the error model on, the reference's ambiguity vectors depend on hidden mutable state:
                # updateProbVectTerminalNode (M:3966-4008) edits the shared `ambiguities` table in place, so what
                # probVectTerminalNode returns afterwards depends on the last site it was called for.
                assert list(a[-1]) == list(b[-1])


@pytest.mark.parametrize("fname", _search_fixtures())
def test_sharding_covers_every_node_once(fname):
    from maple_amd.parallel import shard_nodes
    from maple_amd.tree_host import HostTree
    with gzip.open(os.path.join(GOLDEN, fname), "rt") as fh:
        t = json.load(fh)["tree"]
    tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"],
                    t["probVectUpRight"], t["probVectUpLeft"], t["probVectTotUp"])
    reachable = sorted(tree.preorder())       # tree surgery leaves unused node slots behind
    with gzip.open(os.path.join(GOLDEN, fname), "rt") as fh:
        core_ref = json.load(fh).get("coreNum", {})
    for nc, ref_core in core_ref.items():       # the reference's own assignCoreNumbers(tree, t1, nc), M:12164-12195
        got = tree.assign_core_numbers(int(nc))
        assert [got[v] for v in reachable] == [ref_core[v] for v in reachable], nc
        for r in range(int(nc)):
            assert shard_nodes(tree, r, int(nc)) == [v for v in tree.preorder() if ref_core[v] == r]
    for world in (1, 2, 8):
        shards = [shard_nodes(tree, r, world) for r in range(world)]
        flat = sorted(v for s in shards for v in s)
        assert flat == reachable
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    # pre-order with child 0 first (assignCoreNumbers) vs the worker's pop order (child 1 first)
    core = tree.assign_core_numbers(tree.n + 5)
    assert core[tree.root] == 0 and sorted(c for c in core if c is not None) == list(range(len(reachable)))
    if tree.children[tree.root]:
        assert core[tree.children[tree.root][0]] == 1
        assert tree.preorder()[1] == tree.children[tree.root][1]


def test_perturb_diffs_makes_a_distinct_valid_sample():
    """synth.perturb_diffs: the new sample keeps every entry of the original, adds exactly n_extra substitutions at
    untouched positions, stays sorted and converts to a well-formed genome list."""
    import numpy as np
    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.synth import make_dataset, perturb_diffs
    d = make_dataset(n_samples=40, l_ref=3000, seed=9, mean_diffs=12.0, frac_with_n=0.3, frac_ambig=0.3, n_run_len=(5, 80))
    ref_idx, _ = reference_tables(d.ref)
    rng = np.random.default_rng(1)
    for dl in d.diffs:
        new = perturb_diffs(dl, d.ref, rng, n_extra=2)
        assert len(new) == len(dl) + 2 and all(e in new for e in dl)
        assert [e[1] for e in new] == sorted(e[1] for e in new)
        added = [e for e in new if e not in dl]
        assert all(e[0] != d.ref[e[1] - 1] for e in added)
        gl = tip_genome_list(new, ref_idx)
        pos = 0
        for e in gl:
            pos = e[1] if e[0] in (4, 5) else pos + 1
        assert pos == len(d.ref)


def test_usable_host_threads_is_positive_and_bounded():
    import os
    import bench
    n = bench.usable_host_threads()
    assert 1 <= n <= (os.cpu_count() or 1)


def _check_compact_line(line, text):
    import bench
    assert len(text) <= bench.LINE_LIMIT and "\n" not in text
    assert json.loads(text) == json.loads(json.dumps(line))
    for k in ("metric", "value", "value_walked", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["higher_is_better"] is True and line["scaling"] == "strong" and line["vs_baseline"] is None and line["dtype"] == "f64"
    cfg = line["config"]
    assert "workload" in cfg and cfg["samples"] == 100000 and cfg["model"] == "ratevar"
    assert "100000" in cfg["workload"] and "per-site rates" in cfg["workload"]
    # value = the search's own candidate placements over the wall time of the timed steps
    assert abs(line["value"] - cfg["candidate_placements_timed"] / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-4 * line["value"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["kernel"].startswith("k_")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-4 * max(1.0, r["achieved"])
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["sample"]) <= 120


def test_bench_line_is_small_and_follows_the_contract():
    """bench.compact_line -- the function that makes the ONE line `python bench.py` prints -- on the full record of an MI355X run
    (profiles/r04_bench_line.json: round 4's 25 KB line, which the driver could not parse; profiles/r05_bench_detail.json: this
    round's bench_detail.json): the line stays under bench.LINE_LIMIT bytes and carries the fields the driver and the judge read
    (metric / value / roofline / cpu_baseline / tree_log_lk ...), with consistent arithmetic, on BASELINE.json's configs[2]."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    line, text = bench.compact_line(full)
    _check_compact_line(line, text)
    assert line["config_1M"]["samples"] == 1000000 and line["config_1M"]["roofline"]["frac"] > 0
    # the worst case still fits: every prose field at its bound
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    assert len(bench.compact_line(full)[1]) <= bench.LINE_LIMIT
    cur = os.path.join(ROOT, "profiles", "r05_bench_detail.json")
    if os.path.exists(cur):
        full = json.load(open(cur))
        line, text = bench.compact_line(full)
        _check_compact_line(line, text)
        lk = line["tree_log_lk"]
        assert lk["rel_delta"] <= 1e-6 and abs(lk["gpu"] - lk["oracle"]) <= 1e-6 * abs(lk["oracle"])
        printed = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
        again = json.loads(text)                              # (what the run printed: every field of it as compact_line makes it today)
        for k, v in printed.items():
            if isinstance(v, dict):
                assert all(again[k][k2] == v2 for k2, v2 in v.items()), k
            else:
                assert again[k] == v, k


def test_oracle_tree_log_lk_from_the_tips_equals_calculateTreeLikelihood_over_stored_lists(monkeypatch):
    """bench.oracle_tree_log_lk (the oracle's side of the metric's "tree log-LK delta": calculateTreeLikelihood from the tips'
    lists alone, M:9721-9779 over M:6031-6200) against tree_host.tree_log_likelihood over the lists a full build stored -- both
    through oracle/libmaple_cpu.so here (no GPU); the GPU test of the same name compares the library's value."""
    import ctypes
    import subprocess
    import bench
    from maple_amd.host import reference_tables, tip_genome_list
    from maple_amd.runtime import Device
    from maple_amd.synth import make_dataset
    import maple_amd.mat as mat
    from maple_amd.tree_host import HostTree, rebuild_genome_lists, tree_log_likelihood
    from maple_amd.tree_mirror import TreeMirror
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmaple_cpu.so"))
    lib.maple_last_error.restype = ctypes.c_char_p
    d = make_dataset(n_samples=200, l_ref=3000, seed=5, mean_diffs=12.0, frac_with_n=0.3, frac_ambig=0.3, n_run_len=(5, 80),
                     rate_variation=True)
    ref_idx, rf = reference_tables(d.ref)
    for mode in ("unrest", "ratevar", "siteerr"):
        mkw = bench.model_kwargs(mode, len(ref_idx))
        tip_kw = dict(error_rates=mkw["errorRates"]) if mode == "siteerr" else {}
        dev = Device(ref_idx, rf, lib=lib)
        dev.set_model(**mkw)
        m = TreeMirror(dev, d.parent, d.blen, {int(v): tip_genome_list(dl, ref_idx, **tip_kw) for v, dl in zip(d.tip_node, d.diffs)})
        tip_ids = m.lower.copy()
        m.build(native=False)
        tree = HostTree.from_mirror(m)
        want = tree_log_likelihood(dev, tree)
        tips = np.nonzero(m.is_tip)[0]
        cpu = Device(ref_idx, rf, lib=lib)
        cpu.set_model(**mkw)
        got = bench.oracle_tree_log_lk(cpu, m.root, m.parent, m.children, m.dist, tips, dev.download_packed(tip_ids[tips]))
        assert got == want and want[0] < want[1] < 0
        # ... and in the form with MAT local references (the twin has no maple_tree_rebuild_lists: the Python level loop)
        monkeypatch.setattr(mat, "rebuild_genome_lists", lambda d_, t_: rebuild_genome_lists(d_, t_, native=False))
        assert mat.add_local_references(dev, tree, 20) > 3
        want_mat = tree_log_likelihood(dev, tree)
        got_mat = bench.oracle_tree_log_lk(cpu, m.root, m.parent, m.children, m.dist, tips, dev.download_packed(tip_ids[tips]), tree.mutations)
        assert got_mat == want_mat and 0 < abs(want_mat[0] - want[0]) < 1e-3 * abs(want[0])
        cpu.close()
        dev.close()


def test_native_generator_and_packed_tip_lists_match_the_python_forms():
    """The C generator of the big synthetic inputs (csrc/synth_gen.c, "synth v2") writes valid MAPLE entries (increasing,
    non-overlapping, never equal to the reference), a binary tree whose parents precede their children, and is seeded; and
    host.tip_lists_packed gives, for all samples at once, exactly pack_lists(tip_genome_list(...)) -- with and without the
    error model's smearing of ambiguity vectors (M:3921-3937)."""
    from maple_amd.genome_list import pack_lists
    from maple_amd.host import reference_tables, tip_genome_list, tip_lists_packed
    from maple_amd.synth import make_dataset_native
    d = make_dataset_native(n_samples=3000, l_ref=29903, seed=7, mean_diffs=30.0, rate_variation=True, frac_with_n=0.3,
                            frac_ambig=0.3)
    d2 = make_dataset_native(n_samples=3000, l_ref=29903, seed=7, mean_diffs=30.0, rate_variation=True, frac_with_n=0.3,
                             frac_ambig=0.3)
    c = d.diffs
    assert np.array_equal(c.pos, d2.diffs.pos) and np.array_equal(d.parent, d2.parent) and d.ref == d2.ref
    n = len(d.parent)
    assert n == 2 * 3000 - 1 and d.parent[0] == -1 and (d.parent[1:] < np.arange(1, n)).all()
    assert (np.bincount(d.parent[1:], minlength=n) == np.where(np.isin(np.arange(n), d.tip_node), 0, 2)).all()
    assert 20 < len(c.code) / 3000 < 45
    seen_n = seen_o = 0
    for i in range(len(c)):
        last = 0
        for m in c[i]:
            assert m[1] > last, (i, m)
            last = m[1] + (m[2] - 1 if len(m) > 2 else 0)
            if m[0] in "acgt":
                assert d.ref[m[1] - 1] != m[0]
            seen_n += m[0] == "n"
            seen_o += m[0] not in "acgtn"
        assert last <= 29903
    assert seen_n > 300 and seen_o > 300
    ref_idx, _ = reference_tables(d.ref)
    er = np.exp(np.random.default_rng(4).uniform(np.log(1e-10), np.log(1e-3), size=len(ref_idx)))
    for kw in ({}, {"error_rates": er}, {"error_rate": 1e-4}):
        got = tip_lists_packed(c.off, c.code, c.pos, c.length, ref_idx, **kw)
        want = pack_lists([tip_genome_list(c[i], ref_idx, **kw) for i in range(len(c))], bool(kw))
        for a in ("ent_off", "pos", "meta", "aux_off", "aux"):
            assert np.array_equal(getattr(got, a), getattr(want, a)), (kw.keys(), a)


def test_depth_numbering_of_the_serial_loop_keeps_room_and_renumbers():
    """bench.tree_depths / place_depths (the depths maple_update_partials compares): a node put on a branch gets a depth strictly
    between its neighbours', a new tip one below its parent with room for a later node above it, and the tree is numbered again
    when a branch has no depth left -- checked on a random tree with thousands of placements piled onto the same branches."""
    import bench
    rng = np.random.default_rng(1)
    n_tips = 150
    cap = 2 * n_tips - 1 + 2 * 2000
    up, c0, c1 = (np.full(cap, -1, np.int32) for _ in range(3))
    nodes, nxt = list(range(n_tips)), n_tips
    while len(nodes) > 1:
        i, j = (int(x) for x in rng.choice(len(nodes), 2, replace=False))
        a, b = nodes[i], nodes[j]
        up[a] = up[b] = nxt
        c0[nxt], c1[nxt] = a, b
        nodes = [x for k, x in enumerate(nodes) if k not in (i, j)] + [nxt]
        nxt += 1
    root, n = nodes[0], 2 * n_tips - 1
    depth, step = bench.tree_depths(root, c0, c1, n, cap)
    assert step >= bench.DEPTH_STEP and depth[root] == 0
    renumbered = 0
    for k in range(2000):
        b = 0 if k % 2 == 0 else n - 1                                 # the same tip again and again / the sample added last
        g, p, s = int(up[b]), n, n + 1
        before = depth
        depth, step = bench.place_depths(depth, step, g, p, b, s, root, c0, c1, n)
        renumbered += depth is not before
        if c0[g] == b:
            c0[g] = p
        else:
            c1[g] = p
        up[p], c0[p], c1[p], up[b], up[s] = g, b, s, p, p
        n += 2
        assert depth[g] < depth[p] < depth[b] and depth[p] < depth[s]
    ch = np.nonzero(up[:n] >= 0)[0]
    assert (depth[ch] > depth[up[ch]]).all() and renumbered > 0
