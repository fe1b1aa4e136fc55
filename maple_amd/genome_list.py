"""Genome lists: the reference's tuple form <-> the packed CSR form kept in HBM.

The host keeps MAPLE's tree with genome lists as Python lists of tuples, exactly
the grammar documented at MAPLEv0.7.5.4.py:378-390 (entry = ``(type, x[, d0[, d1]][, flag])``
or ``(6, ref[, d0], vec)``).  The device keeps them packed (``include/maple_hip.h``):

    word = {int32 pos, uint32 meta},  meta = type | ref<<3 | hasD0<<5 | hasD1<<6 | flag<<7 | auxoff<<8
    aux  = f64 stream per list: [d0][d1][vec0..3] per entry, in entry order

``pos`` is the last genome position covered by the entry (explicit also for the
single-site types, whose tuple stores the local reference nucleotide instead).
"""
from __future__ import annotations

import numpy as np

T_R, T_N, T_O = 4, 5, 6
HASD0, HASD1, FLAG = 1 << 5, 1 << 6, 1 << 7


class PackedLists:
    """CSR bundle of packed genome lists (host side, numpy)."""

    __slots__ = ("ent_off", "pos", "meta", "aux_off", "aux")

    def __init__(self, ent_off, pos, meta, aux_off, aux):
        self.ent_off = np.ascontiguousarray(ent_off, dtype=np.int64)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.meta = np.ascontiguousarray(meta, dtype=np.uint32)
        self.aux_off = np.ascontiguousarray(aux_off, dtype=np.int64)
        self.aux = np.ascontiguousarray(aux, dtype=np.float64)

    def __len__(self):
        return len(self.ent_off) - 1


def pack_lists(lists, using_error_rate: bool) -> PackedLists:
    """Pack genome lists given in the reference's tuple form."""
    u = 1 if using_error_rate else 0
    ent_off = [0]
    aux_off = [0]
    pos_a, meta_a, aux_a = [], [], []
    for gl in lists:
        p = 0
        base = len(aux_a)
        for e in gl:
            t = e[0]
            n = len(e)
            auxoff = len(aux_a) - base
            if auxoff >= (1 << 24):
                raise ValueError("genome list too long for the 24-bit aux offset")
            if t == T_N:
                p = e[1]
                meta = T_N
            elif t == T_O:
                p += 1
                meta = T_O | (e[1] << 3)
                if n == 4:
                    meta |= HASD0
                    aux_a.append(float(e[2]))
                elif n != 3:
                    raise ValueError(f"bad O entry {e!r}")
                aux_a.extend(float(x) for x in e[-1])
            else:
                if t == T_R:
                    p = e[1]
                    meta = T_R
                else:
                    p += 1
                    meta = t | (e[1] << 3)
                if n == 2:
                    pass
                elif n == 3 + u:
                    meta |= HASD0
                    aux_a.append(float(e[2]))
                    if u and e[3]:
                        meta |= FLAG
                elif n == 4 + u:
                    meta |= HASD0 | HASD1
                    aux_a.append(float(e[2]))
                    aux_a.append(float(e[3]))
                    if u and e[4]:
                        meta |= FLAG
                else:
                    raise ValueError(f"entry {e!r} has a tuple length the reference never builds (usingErrorRate={u})")
            pos_a.append(p)
            meta_a.append(meta | (auxoff << 8))
        ent_off.append(len(pos_a))
        aux_off.append(len(aux_a))
    return PackedLists(ent_off, pos_a, meta_a, aux_off, aux_a)


def unpack_list(pos, meta, aux, using_error_rate: bool):
    """One packed list -> the reference's tuple form (a list of tuples)."""
    u = bool(using_error_rate)
    out = []
    for p, m in zip(pos.tolist(), meta.tolist()):
        t = m & 7
        ref = (m >> 3) & 3
        a = m >> 8
        d0 = d1 = None
        if m & HASD0:
            d0 = float(aux[a]); a += 1
        if m & HASD1:
            d1 = float(aux[a]); a += 1
        fl = bool(m & FLAG)
        if t == T_N:
            out.append((5, p))
        elif t == T_O:
            vec = [float(x) for x in aux[a:a + 4]]
            out.append((6, ref, vec) if d0 is None else (6, ref, d0, vec))
        else:
            x = p if t == T_R else ref
            if d0 is None:
                out.append((t, x))
            elif d1 is None:
                out.append((t, x, d0, fl) if u else (t, x, d0))
            else:
                out.append((t, x, d0, d1, fl) if u else (t, x, d0, d1))
    return out


def unpack_lists(pl: PackedLists, using_error_rate: bool):
    res = []
    for i in range(len(pl)):
        e0, e1 = pl.ent_off[i], pl.ent_off[i + 1]
        a0, a1 = pl.aux_off[i], pl.aux_off[i + 1]
        res.append(unpack_list(pl.pos[e0:e1], pl.meta[e0:e1], pl.aux[a0:a1], using_error_rate))
    return res


def pack_mutations(mut_lists):
    """tree.mutations[node] lists of (pos, from, to) -> CSR (off int64[n+1], mut3 int32[m,3])."""
    off = [0]
    flat = []
    for ml in mut_lists:
        for m in ml:
            flat.append((int(m[0]), int(m[1]), int(m[2])))
        off.append(len(flat))
    arr = np.asarray(flat, dtype=np.int32).reshape(-1, 3)
    return np.asarray(off, dtype=np.int64), np.ascontiguousarray(arr)
