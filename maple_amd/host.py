"""Host-side pieces of MAPLE that sit either side of the accelerated path: the MAPLE-format
reader, the reference-derived tables and the tip genome lists (the *query* lists of a placement).

Behaviour follows MAPLEv0.7.5.4.py (cited as M:<line>); the code is written from scratch.
"""
from __future__ import annotations

import gzip

import numpy as np

NUC_INDEX = {"a": 0, "c": 1, "g": 2, "t": 3}
# IUPAC ambiguity codes -> unnormalised 0/1 state vectors (as the reference stores them, M:3665)
AMBIGUITY = {
    "y": (0.0, 1.0, 0.0, 1.0), "r": (1.0, 0.0, 1.0, 0.0), "w": (1.0, 0.0, 0.0, 1.0), "s": (0.0, 1.0, 1.0, 0.0),
    "k": (0.0, 0.0, 1.0, 1.0), "m": (1.0, 1.0, 0.0, 0.0), "d": (1.0, 0.0, 1.0, 1.0), "v": (1.0, 1.0, 1.0, 0.0),
    "h": (1.0, 1.0, 0.0, 1.0), "b": (0.0, 1.0, 1.0, 1.0),
}


def read_maple_file(path):
    """MAPLE-format reader (same format as readConciseAlignment, M:3498-3553).

    Returns (reference string in lower case, {sample name: [(char, pos[, length]), ...]}).
    """
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        header = fh.readline()
        if not header.startswith(">"):
            raise ValueError("MAPLE file must start with the reference record")
        ref_parts = []
        line = fh.readline()
        while line and not line.startswith(">"):
            ref_parts.append(line.strip())
            line = fh.readline()
        ref = "".join(ref_parts).lower()
        data = {}
        while line and line != "\n":
            name = line[1:].rstrip("\n")
            entries = []
            last = 0
            line = fh.readline()
            while line and line != "\n" and not line.startswith(">"):
                cols = line.split()
                if len(cols) < 2:
                    raise ValueError(f"line with a single column in {path}: {line!r}")
                ch, pos = cols[0].lower(), int(cols[1])
                e = (ch, pos, int(cols[2])) if len(cols) > 2 else (ch, pos)
                if ch not in ("n", "-") and ref[pos - 1] == ch:
                    raise ValueError(f"entry {e} equals the reference nucleotide")
                if pos <= last:
                    raise ValueError(f"entry {e} of sample {name} overlaps the previous one")
                entries.append(e)
                last = pos if len(e) == 2 else pos + e[2] - 1
                line = fh.readline()
            data[name] = entries
    return ref, data


def reference_tables(ref: str, model: str = "UNREST"):
    """refIndeces and rootFreqs (M:3669-3689)."""
    ref_idx = np.fromiter((NUC_INDEX.get(ch, 0) for ch in ref.lower()), dtype=np.uint8, count=len(ref))
    counts = [0, 0, 0, 0]
    for ch in ref.lower():
        k = NUC_INDEX.get(ch)
        if k is not None:
            counts[k] += 1
    l_ref = float(len(ref))
    root_freqs = [c / l_ref for c in counts]
    if model == "JC":
        root_freqs = [0.25, 0.25, 0.25, 0.25]
    return ref_idx, root_freqs


def tip_genome_list(diffs, ref_idx, *, only_n_ambiguities=False, error_rate=None, error_rates=None,
                    num_minor=0):
    """Genome list of a sample in the root frame (probVectTerminalNode with node=None, M:3882-3944).

    ``error_rate``/``error_rates`` switch on the error-model smearing of ambiguity vectors
    (M:3921-3937); leave both None when the error model is off.
    """
    l_ref = len(ref_idx)
    if diffs is None:
        return [(5, l_ref)]
    out = []
    pos = 1
    use_err = (error_rate is not None or error_rates is not None) and num_minor == 0
    for m in diffs:
        cur = m[1]
        if cur > pos:
            out.append((4, cur - 1))
            pos = cur
        ch = m[0]
        if ch == "n" or ch == "-":
            length = m[2] if len(m) > 2 else 1
            out.append((5, cur + length - 1))
            pos = cur + length
            continue
        r = int(ref_idx[cur - 1])
        if ch in NUC_INDEX:
            k = NUC_INDEX[ch]
            out.append((4, cur) if k == r else (k, r))
        elif only_n_ambiguities:
            out.append((5, cur))
        else:
            vec = list(AMBIGUITY[ch])
            if use_err:
                e = error_rates[cur - 1] if error_rates is not None else error_rate
                nstates = sum(1 for v in vec if v)
                if nstates == 2:
                    vec = [e * 0.33333 if v == 0 else v - e * 0.33333 for v in vec]
                elif nstates == 3:
                    vec = [e * 0.33333 if v == 0 else v - e / 9 for v in vec]
            out.append((6, r, vec))
        pos = cur + 1
    if pos <= l_ref:
        out.append((4, l_ref))
    return out


_CODE_STATES = np.zeros((256, 4), dtype=np.float64)
for _ch, _v in AMBIGUITY.items():
    _CODE_STATES[ord(_ch)] = _v


def tip_lists_packed(off, code, pos, length, ref_idx, *, error_rates=None, error_rate=None):
    """tip_genome_list for MANY samples at once, straight into the packed form (genome_list.PackedLists): the samples'
    MAPLE entries as flat arrays (``off[n + 1]``, ``code`` = the entry's character, ``pos`` 1-based, ``length`` = 1 or the
    length of an n / - run) -- what synth.DiffCSR holds.  The same lists as ``pack_lists([tip_genome_list(d, ref_idx, ...)])``
    (tests/test_abi_and_host.py), without a Python object per entry."""
    from .genome_list import PackedLists
    off = np.asarray(off, dtype=np.int64)
    code = np.asarray(code, dtype=np.uint8)
    pos = np.asarray(pos, dtype=np.int64)
    length = np.asarray(length, dtype=np.int64)
    l_ref = len(ref_idx)
    n, nd = len(off) - 1, len(code)
    cnt = np.diff(off)
    has = cnt > 0
    first = off[:-1][has]                                            # the first entry of every sample that has one
    sample = np.repeat(np.arange(n, dtype=np.int64), cnt)
    is_n = (code == ord("n")) | (code == ord("-"))
    nuc = np.full(nd, 255, dtype=np.uint32)
    for ch, k in NUC_INDEX.items():
        nuc[code == ord(ch)] = k
    is_nuc = nuc < 4
    is_o = ~is_n & ~is_nuc
    if is_o.any() and (_CODE_STATES[code[is_o]].sum(axis=1) == 0).any():
        raise ValueError("unknown character in a MAPLE entry")
    end = np.where(is_n, pos + length - 1, pos)                     # last position the entry covers
    prev_end = np.zeros(nd, dtype=np.int64)
    prev_end[1:] = end[:-1]
    prev_end[first] = 0
    pre_r = pos > prev_end + 1                                       # a reference run before the entry (M:3894-3896)
    last_end = np.zeros(n, dtype=np.int64)
    last_end[has] = end[off[1:][has] - 1]
    tail_r = last_end < l_ref                                        # ... and after the sample's last entry (M:3940-3941)
    per_diff = pre_r.astype(np.int64) + 1
    ent_cnt = np.bincount(sample, weights=per_diff, minlength=n).astype(np.int64) + tail_r
    ent_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ent_cnt, out=ent_off[1:])

    def within(x):
        """exclusive running sum of x inside every sample"""
        c = np.cumsum(x) - x
        return c - np.repeat(c[first], cnt[has])
    at = ent_off[sample] + within(per_diff) + pre_r                  # where the entry itself goes
    # aux stream: four doubles per O entry; every entry's word carries the running offset (as pack_lists stores it)
    o_idx = np.nonzero(is_o)[0]
    n_o = np.bincount(sample[o_idx], minlength=n).astype(np.int64)
    aux_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(4 * n_o, out=aux_off[1:])
    auxoff = (4 * within(is_o.astype(np.int64))).astype(np.uint32) << 8
    r = ref_idx[np.clip(pos - 1, 0, l_ref - 1)].astype(np.uint32)
    same = is_nuc & (nuc == r)                                       # an explicit nucleotide equal to the reference: an R entry
    meta = np.where(is_n, 5, np.where(is_o, 6 | (r << 3), np.where(same, 4, (nuc & 3) | (r << 3)))).astype(np.uint32)
    out_pos = np.zeros(int(ent_off[-1]), dtype=np.int32)
    out_meta = np.zeros(int(ent_off[-1]), dtype=np.uint32)
    out_pos[at] = end
    out_meta[at] = meta | auxoff
    pr = np.nonzero(pre_r)[0]
    out_pos[at[pr] - 1] = pos[pr] - 1
    out_meta[at[pr] - 1] = 4 | auxoff[pr]
    tl = np.nonzero(tail_r)[0]
    out_pos[ent_off[1:][tl] - 1] = l_ref
    out_meta[ent_off[1:][tl] - 1] = 4 | ((4 * n_o[tl]).astype(np.uint32) << 8)
    vec = _CODE_STATES[code[o_idx]].copy()
    if error_rates is not None or error_rate is not None:            # the error model smears the ambiguity vectors, M:3921-3937
        e = (np.asarray(error_rates, dtype=np.float64)[pos[o_idx] - 1] if error_rates is not None
             else np.full(len(o_idx), float(error_rate)))
        ns = (vec != 0).sum(axis=1)[:, None]
        e3 = (e * 0.33333)[:, None]
        vec = np.where(ns == 2, np.where(vec == 0, e3, vec - e3),
                       np.where(ns == 3, np.where(vec == 0, e3, vec - (e / 9)[:, None]), vec))
    return PackedLists(ent_off, out_pos, out_meta, aux_off, vec.reshape(-1))
