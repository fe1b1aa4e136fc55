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
