"""Synthetic MAPLE-format inputs (SURVEY.md §8d "Synthetic inputs").

A seeded generator of SARS-CoV-2-like data: a reference genome, a random
bifurcating tree, mutations dropped on its branches, and the resulting
per-sample difference lists written in MAPLE format (the format parsed by the
reference's ``readConciseAlignment``, MAPLEv0.7.5.4.py:3498-3553), so that the
reference and this package read the same file.

Nothing here is on the hot path; it only produces inputs for tests, golden
vector generation and ``bench.py``.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass, field

import numpy as np

NUC = "acgt"
# two-state IUPAC codes and the pair of states they stand for
AMBIG2 = {"r": (0, 2), "y": (1, 3), "s": (1, 2), "w": (0, 3), "k": (2, 3), "m": (0, 1)}
AMBIG2_BY_PAIR = {v: k for k, v in AMBIG2.items()}
AMBIG3 = {"b": (1, 2, 3), "d": (0, 2, 3), "h": (0, 1, 3), "v": (0, 1, 2)}

# UNREST matrix estimated by the reference on example_files/sameRef_B.1.429
# (rounded; only used as a plausible generator, never as a golden value).
SARS2_Q = np.array(
    [
        [-0.56, 0.06, 0.37, 0.13],
        [0.16, -2.60, 0.04, 2.40],
        [0.84, 0.13, -2.40, 1.43],
        [0.07, 0.48, 0.05, -0.60],
    ]
)
SARS2_FREQS = np.array([0.299, 0.183, 0.196, 0.322])


@dataclass
class SynthData:
    ref: str                      # lower-case reference, length lRef
    names: list                   # sample names
    diffs: list                   # per sample: list of (char, pos[, length]) MAPLE entries
    parent: np.ndarray            # true tree: parent index per node (root = -1)
    blen: np.ndarray              # true branch lengths (subs/site)
    tip_node: np.ndarray          # node index of each sample
    site_rates: np.ndarray | None = None
    meta: dict = field(default_factory=dict)


def random_reference(l_ref: int, rng: np.random.Generator) -> np.ndarray:
    return rng.choice(4, size=l_ref, p=SARS2_FREQS / SARS2_FREQS.sum()).astype(np.int8)


def random_tree(n_tips: int, rng: np.random.Generator):
    """Random bifurcating topology by sequential random attachment.

    Returns (parent[2n-1], is_tip[2n-1]); node 0 is the root.
    """
    n_nodes = 2 * n_tips - 1
    parent = np.full(n_nodes, -1, dtype=np.int64)
    is_tip = np.zeros(n_nodes, dtype=bool)
    if n_tips == 1:
        is_tip[0] = True
        return parent, is_tip
    # start: root 0 with two tip children 1, 2
    is_tip[1] = is_tip[2] = True
    parent[1] = parent[2] = 0
    nxt = 3
    tips = [1, 2]
    for _ in range(n_tips - 2):
        # split a random existing tip into an internal node with two tips
        k = int(rng.integers(len(tips)))
        t = tips[k]
        a, b = nxt, nxt + 1
        nxt += 2
        is_tip[t] = False
        is_tip[a] = is_tip[b] = True
        parent[a] = parent[b] = t
        tips[k] = a
        tips.append(b)
    return parent, is_tip


def make_dataset(
    n_samples: int,
    l_ref: int = 29903,
    seed: int = 1,
    mean_diffs: float = 30.0,
    rate_variation: bool = False,
    frac_with_n: float = 0.01,
    n_run_len=(50, 500),
    frac_ambig: float = 0.01,
    frac_ambig3: float = 0.0,
    zero_branch_frac: float = 0.15,
) -> SynthData:
    rng = np.random.default_rng(seed)
    ref = random_reference(l_ref, rng)
    parent, is_tip = random_tree(n_samples, rng)
    n_nodes = len(parent)
    # depth of each node (parents always have a smaller index than children)
    depth = np.zeros(n_nodes, dtype=np.int64)
    for v in range(1, n_nodes):
        depth[v] = depth[parent[v]] + 1
    mean_depth = max(1.0, float(depth[is_tip].mean()))
    per_branch = mean_diffs / mean_depth / max(1e-9, 1.0 - zero_branch_frac)
    site_rates = None
    if rate_variation:
        site_rates = np.clip(rng.gamma(0.5, 2.0, size=l_ref), 0.001, 0.005 * l_ref)
        site_p = site_rates / site_rates.sum()
    else:
        site_p = None
    # exit probabilities per from-state
    exitp = SARS2_Q.copy()
    np.fill_diagonal(exitp, 0.0)
    exitp = exitp / exitp.sum(axis=1, keepdims=True)

    nmut = rng.poisson(per_branch, size=n_nodes)
    nmut[rng.random(n_nodes) < zero_branch_frac] = 0
    nmut[0] = 0
    blen = nmut.astype(np.float64) / l_ref
    # state of every node as a sparse dict pos->nuc (relative to ref)
    states = [None] * n_nodes
    states[0] = {}
    for v in range(1, n_nodes):
        st = states[parent[v]]
        k = int(nmut[v])
        if k:
            st = dict(st)
            pos = rng.choice(l_ref, size=k, p=site_p) if site_p is not None else rng.integers(l_ref, size=k)
            for p in pos:
                p = int(p)
                cur = st.get(p, int(ref[p]))
                new = int(rng.choice(4, p=exitp[cur]))
                if new == int(ref[p]):
                    st.pop(p, None)
                else:
                    st[p] = new
        states[v] = st
        # free memory of fully processed internal nodes lazily: children always follow

    tip_nodes = np.nonzero(is_tip)[0]
    order = rng.permutation(len(tip_nodes))
    tip_nodes = tip_nodes[order]
    names, diffs = [], []
    for i, v in enumerate(tip_nodes):
        st = dict(states[v])
        entries = {}
        for p, c in st.items():
            entries[p + 1] = (NUC[c], p + 1)
        if rng.random() < frac_ambig:
            for _ in range(int(rng.integers(1, 3))):
                p = int(rng.integers(l_ref))
                r = int(ref[p])
                cur = st.get(p, r)
                other = int(rng.choice([x for x in range(4) if x != cur]))
                pair = tuple(sorted((cur, other)))
                entries[p + 1] = (AMBIG2_BY_PAIR[pair], p + 1)
        if rng.random() < frac_ambig3:
            p = int(rng.integers(l_ref))
            code = list(AMBIG3)[int(rng.integers(4))]
            entries[p + 1] = (code, p + 1)
        n_runs = []
        if rng.random() < frac_with_n:
            for _ in range(int(rng.integers(1, 4))):
                ln = int(rng.integers(n_run_len[0], n_run_len[1] + 1))
                s = int(rng.integers(1, max(2, l_ref - ln)))
                n_runs.append((s, min(ln, l_ref - s + 1)))
        # merge N runs, drop covered entries
        n_runs.sort()
        merged = []
        for s, ln in n_runs:
            if merged and s <= merged[-1][0] + merged[-1][1]:
                e = max(merged[-1][0] + merged[-1][1], s + ln)
                merged[-1] = (merged[-1][0], e - merged[-1][0])
            else:
                merged.append((s, ln))
        for s, ln in merged:
            for p in [q for q in entries if s <= q < s + ln]:
                del entries[p]
            entries[s] = ("n", s, ln)
        diffs.append([entries[k] for k in sorted(entries)])
        names.append(f"S{i:07d}")
    ref_s = "".join(NUC[c] for c in ref)
    return SynthData(
        ref=ref_s, names=names, diffs=diffs, parent=parent, blen=blen,
        tip_node=tip_nodes, site_rates=site_rates,
        meta=dict(n_samples=n_samples, l_ref=l_ref, seed=seed, mean_diffs=mean_diffs,
                  per_branch=per_branch, rate_variation=rate_variation),
    )


def perturb_diffs(diffs, ref: str, rng: np.random.Generator, n_extra: int = 2) -> list:
    """A new sample close to an existing one: the same MAPLE entries plus `n_extra` substitutions at positions the
    sample does not already touch (so it is neither identical to, nor a minor sequence of, the original)."""
    covered = set()
    for m in diffs:
        for p in range(m[1], m[1] + (m[2] if len(m) > 2 else 1)):
            covered.add(p)
    out = list(diffs)
    while n_extra > 0:
        p = int(rng.integers(1, len(ref) + 1))
        if p in covered:
            continue
        alt = [b for b in "acgt" if b != ref[p - 1]]
        out.append((alt[int(rng.integers(3))], p))
        covered.add(p)
        n_extra -= 1
    out.sort(key=lambda m: m[1])
    return out


def write_maple(data: SynthData, path: str) -> None:
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wt") as fh:
        fh.write(">reference\n")
        fh.write(data.ref + "\n")
        for name, dl in zip(data.names, data.diffs):
            fh.write(">" + name + "\n")
            for e in dl:
                if len(e) == 3:
                    fh.write(f"{e[0]}\t{e[1]}\t{e[2]}\n")
                else:
                    fh.write(f"{e[0]}\t{e[1]}\n")


# ---- "synth v2": the same model generated by csrc/synth_gen.c (seconds instead of minutes at 1 000 000 samples) --------------
class DiffCSR:
    """The samples' MAPLE entries as three flat arrays + offsets; indexing gives the tuple form make_dataset returns."""

    def __init__(self, off, code, pos, length):
        self.off, self.code, self.pos, self.length = off, code, pos, length

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        lo, hi = int(self.off[i]), int(self.off[i + 1])
        out = []
        for c, p, ln in zip(self.code[lo:hi].tolist(), self.pos[lo:hi].tolist(), self.length[lo:hi].tolist()):
            ch = chr(c)
            out.append((ch, p, ln) if ch == "n" else (ch, p))
        return out

    def __iter__(self):
        return (self[i] for i in range(len(self)))


_synth_lib = None


def _load_synth_lib():
    global _synth_lib
    if _synth_lib is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmaple_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path)

        class S(C.Structure):
            _fields_ = [("n_samples", C.c_int64), ("n_nodes", C.c_int64), ("l_ref", C.c_int64), ("n_diffs", C.c_int64),
                        ("ref", C.POINTER(C.c_int8)), ("parent", C.POINTER(C.c_int64)), ("blen", C.POINTER(C.c_double)),
                        ("tip_node", C.POINTER(C.c_int64)), ("diff_off", C.POINTER(C.c_int64)),
                        ("diff_code", C.POINTER(C.c_uint8)), ("diff_pos", C.POINTER(C.c_int32)),
                        ("diff_len", C.POINTER(C.c_int32)), ("mean_depth", C.c_double), ("per_branch", C.c_double)]
        lib.maple_synth_generate.restype = C.POINTER(S)
        lib.maple_synth_generate.argtypes = [C.c_int64, C.c_int64, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_double]
        lib.maple_synth_free.argtypes = [C.POINTER(S)]
        lib.maple_synth_free.restype = None
        _synth_lib = lib
    return _synth_lib


def make_dataset_native(n_samples: int, l_ref: int = 29903, seed: int = 1, mean_diffs: float = 30.0,
                        rate_variation: bool = False, frac_with_n: float = 0.01, n_run_len=(50, 500),
                        frac_ambig: float = 0.01, zero_branch_frac: float = 0.15) -> SynthData:
    """make_dataset's model (same tree process, substitution process, ambiguities and missing-data runs) from the C generator:
    ``diffs`` is a DiffCSR (``.off/.code/.pos/.length``), everything else as in make_dataset.  Its own seeded stream: the
    data differ from make_dataset's for the same seed."""
    lib = _load_synth_lib()
    site_rates = None
    site_cdf = None
    if rate_variation:
        site_rates = np.clip(np.random.default_rng(seed).gamma(0.5, 2.0, size=l_ref), 0.001, 0.005 * l_ref)
        site_cdf = np.ascontiguousarray(np.cumsum(site_rates / site_rates.sum()))
        site_cdf[-1] = 2.0
    freqs_cdf = np.ascontiguousarray(np.cumsum(SARS2_FREQS / SARS2_FREQS.sum()))
    exitp = SARS2_Q.copy()
    np.fill_diagonal(exitp, 0.0)
    exit_cdf = np.ascontiguousarray(np.cumsum(exitp / exitp.sum(axis=1, keepdims=True), axis=1))
    exit_cdf[:, 3] = 2.0
    ptr = lambda a: None if a is None else a.ctypes.data  # noqa: E731
    h = lib.maple_synth_generate(n_samples, l_ref, seed, mean_diffs, ptr(site_cdf), ptr(freqs_cdf), ptr(exit_cdf),
                                 frac_with_n, int(n_run_len[0]), int(n_run_len[1]), frac_ambig, zero_branch_frac)
    if not h:
        raise MemoryError("maple_synth_generate failed")
    try:
        s = h.contents
        n, nn, nd = int(s.n_samples), int(s.n_nodes), int(s.n_diffs)
        arr = lambda p, k: np.ctypeslib.as_array(p, shape=(max(1, k),))[:k].copy()  # noqa: E731
        ref = arr(s.ref, l_ref)
        parent, blen, tip_node = arr(s.parent, nn), arr(s.blen, nn), arr(s.tip_node, n)
        diffs = DiffCSR(arr(s.diff_off, n + 1), arr(s.diff_code, nd), arr(s.diff_pos, nd), arr(s.diff_len, nd))
        meta = dict(n_samples=n_samples, l_ref=l_ref, seed=seed, mean_diffs=mean_diffs, per_branch=float(s.per_branch),
                    rate_variation=rate_variation, generator="synth v2 (csrc/synth_gen.c)")
    finally:
        lib.maple_synth_free(h)
    ref_s = np.frombuffer(b"acgt", dtype=np.uint8)[ref].tobytes().decode()
    names = [f"S{i:07d}" for i in range(n)]
    return SynthData(ref=ref_s, names=names, diffs=diffs, parent=parent, blen=blen, tip_node=tip_node,
                     site_rates=site_rates, meta=meta)
