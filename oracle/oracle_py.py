"""ctypes binding of oracle/libmaple_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product (``maple_amd``) never does.
Lists are exchanged in the reference's tuple form (M:378-390).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmaple_oracle.so")

OENTRY = np.dtype(
    [("type", np.int32), ("x", np.int32), ("len", np.int32), ("flag", np.int32),
     ("d0", np.float64), ("d1", np.float64), ("vec", np.float64, (4,))], align=True)
assert OENTRY.itemsize == 64


class OModel(C.Structure):
    _fields_ = [
        ("lRef", C.c_int), ("useRateVariation", C.c_int), ("usingErrorRate", C.c_int),
        ("errorRateSiteSpecific", C.c_int),
        ("refIdx", C.c_void_p), ("siteRates", C.c_void_p), ("errorRates", C.c_void_p),
        ("cumulativeRate", C.c_void_p), ("cumulativeErrorRate", C.c_void_p),
        ("Q", C.c_double * 16), ("rootFreqs", C.c_double * 4),
        ("errorRate", C.c_double), ("totError", C.c_double), ("globalTotRate", C.c_double),
        ("minimumCarryOver", C.c_double), ("thresholdProb", C.c_double), ("minBLenSensitivity", C.c_double),
        ("thresholdDiffForUpdate", C.c_double), ("thresholdFoldChangeUpdate", C.c_double),
    ]


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("maple_oracle.c", "maple_oracle_search.c", "maple_oracle.h")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


def to_entries(gl, u):
    """Reference tuple list -> OENTRY array."""
    arr = np.zeros(len(gl), dtype=OENTRY)
    for k, e in enumerate(gl):
        r = arr[k]
        t = e[0]
        r["type"] = t
        r["x"] = e[1]
        r["len"] = len(e)
        if t == 6:
            if len(e) == 4:
                r["d0"] = e[2]
            r["vec"] = e[-1]
        elif t < 5 and len(e) > 2:
            if u:
                if len(e) == 3:
                    raise ValueError("length-3 tuple with the error model on")
                r["d0"] = e[2]
                if len(e) == 5:
                    r["d1"] = e[3]
                r["flag"] = 1 if e[-1] else 0
            else:
                r["d0"] = e[2]
                if len(e) == 4:
                    r["d1"] = e[3]
    return arr


def from_entries(arr, n, u):
    out = []
    for k in range(n):
        r = arr[k]
        t, x, ln = int(r["type"]), int(r["x"]), int(r["len"])
        if t == 5:
            out.append((5, x))
        elif t == 6:
            vec = [float(v) for v in r["vec"]]
            out.append((6, x, vec) if ln == 3 else (6, x, float(r["d0"]), vec))
        elif ln == 2:
            out.append((t, x))
        elif u:
            fl = bool(r["flag"])
            out.append((t, x, float(r["d0"]), fl) if ln == 4 else (t, x, float(r["d0"]), float(r["d1"]), fl))
        else:
            out.append((t, x, float(r["d0"])) if ln == 3 else (t, x, float(r["d0"]), float(r["d1"])))
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OTreeC(C.Structure):
    _fields_ = [("n", C.c_int), ("root", C.c_int), ("up", C.c_void_p), ("c0", C.c_void_p), ("c1", C.c_void_p),
                ("dist", C.c_void_p), ("nMinor", C.c_void_p), ("ent", C.c_void_p), ("start", C.c_void_p * 4),
                ("len", C.c_void_p * 4), ("mut3", C.c_void_p), ("mutOff", C.c_void_p)]


class OSearchParamsC(C.Structure):
    _fields_ = [("strict", C.c_int), ("allowedFails", C.c_int), ("thrLKtopology", C.c_double),
                ("thrPlacement", C.c_double), ("thrOptTopo", C.c_double), ("thrConsec", C.c_double),
                ("effNon0", C.c_double), ("defaultBLen", C.c_double)]


class OSearchResultC(C.Structure):
    _fields_ = [("bestNode", C.c_int), ("placement", C.c_int), ("status", C.c_int), ("nAppend", C.c_int),
                ("bestScore", C.c_double), ("improvement", C.c_double), ("currentLK", C.c_double),
                ("blen", C.c_double * 3), ("rpr", C.c_void_p), ("rprCap", C.c_int), ("rprN", C.c_int)]


def packed_to_entries_numpy(pk, u):
    """Packed CSR lists (maple_amd.genome_list.PackedLists, e.g. Device.download_packed) -> one OENTRY array + CSR
    offsets, vectorised with numpy (kept as the check of the C conversion below)."""
    n = int(pk.ent_off[-1])
    pos = pk.pos[:n].astype(np.int64)
    meta = pk.meta[:n].astype(np.int64)
    typ = meta & 7
    ref = (meta >> 3) & 3
    has0 = (meta >> 5) & 1
    has1 = (meta >> 6) & 1
    flag = (meta >> 7) & 1
    auxoff = meta >> 8
    # aux base of the list each entry belongs to
    counts = np.diff(pk.ent_off)
    base = np.repeat(pk.aux_off[:-1], counts)
    a0 = base + auxoff
    arr = np.zeros(n, dtype=OENTRY)
    arr["type"] = typ
    arr["x"] = np.where((typ == 4) | (typ == 5), pos, ref)
    arr["flag"] = flag if u else 0
    aux = np.concatenate([pk.aux, np.zeros(8)])
    d0 = np.where(has0 == 1, aux[np.minimum(a0, len(aux) - 1)], 0.0)
    d1 = np.where(has1 == 1, aux[np.minimum(a0 + has0, len(aux) - 1)], 0.0)
    arr["d0"], arr["d1"] = d0, d1
    vbase = a0 + has0 + has1
    isO = typ == 6
    for k in range(4):
        arr["vec"][:, k] = np.where(isO, aux[np.minimum(vbase + k, len(aux) - 1)], 0.0)
    tail = has0 + has1
    ln = np.where(isO, 3 + has0, np.where(typ == 5, 2, 2 + tail + np.where((tail > 0) & bool(u), 1, 0)))
    arr["len"] = ln
    return arr, pk.ent_off.copy()


def packed_to_entries(pk, u, threads=None):
    """Packed CSR lists -> one OENTRY array + CSR offsets, by the oracle library's own converter (omo_entries_from_packed:
    a whole 1 000 000-tip tree is 4 x 10^8 entries)."""
    lib = C.CDLL(build())
    n = int(pk.ent_off[-1])
    arr = np.empty(max(1, n), dtype=OENTRY)[:n]
    if threads is None:
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1))
    lib.omo_entries_from_packed(C.c_longlong(len(pk.ent_off) - 1), _p(pk.ent_off), _p(pk.pos), _p(pk.meta), _p(pk.aux_off),
                                _p(np.concatenate([pk.aux, np.zeros(8)]) if len(pk.aux) < 8 else pk.aux), int(bool(u)), _p(arr),
                                int(threads))
    return arr, pk.ent_off.copy()


class OracleTree:
    """A frozen tree in the oracle's layout: topology + the four genome lists of every node (tuple form or None)."""

    def __init__(self, oracle, root, up, children, dist, mutations, n_minor, lists4):
        u = oracle.u
        n = len(up)
        self.n = n
        if isinstance(up, np.ndarray):                            # (numpy columns: up[n] with -1 at the root, children[n, 2] with -1)
            self.up = np.ascontiguousarray(up, dtype=np.int32)
            self.c0 = np.ascontiguousarray(children[:, 0], dtype=np.int32)
            self.c1 = np.ascontiguousarray(children[:, 1], dtype=np.int32)
        else:
            self.up = np.asarray([-1 if x is None else x for x in up], dtype=np.int32)
            self.c0 = np.asarray([c[0] if c else -1 for c in children], dtype=np.int32)
            self.c1 = np.asarray([c[1] if c else -1 for c in children], dtype=np.int32)
        self.dist = np.asarray(dist, dtype=np.float64)
        self.nMinor = np.asarray(n_minor, dtype=np.int32)
        chunks, self.start, self.len = [], [], []
        tot = 0
        for kind in range(4):
            st = np.zeros(n, dtype=np.int64)
            ln = np.zeros(n, dtype=np.int32)
            if isinstance(lists4[kind], tuple):                   # (node ids, packed lists) straight from the device
                nodes_k, pk = lists4[kind]
                arr, off = packed_to_entries(pk, u)
                st[nodes_k] = tot + off[:-1]
                ln[nodes_k] = np.diff(off)
                tot += len(arr)
                chunks.append(arr)
            else:
                for v in range(n):
                    gl = lists4[kind][v]
                    if gl:
                        arr = to_entries([tuple(e) for e in gl], u)
                        st[v], ln[v] = tot, len(arr)
                        tot += len(arr)
                        chunks.append(arr)
            self.start.append(st)
            self.len.append(ln)
        self.ent = np.concatenate(chunks) if chunks else np.zeros(1, dtype=OENTRY)
        off = np.zeros(n + 1, dtype=np.int64)
        flat = []
        if mutations is not None:                                 # (None: a tree without MAT local references)
            for v in range(n):
                flat.extend(mutations[v])
                off[v + 1] = len(flat)
        self.mut3 = np.ascontiguousarray(np.asarray(flat if flat else [[0, 0, 0]], dtype=np.int32).reshape(-1, 3))
        self.mutOff = off
        t = OTreeC()
        t.n, t.root = n, root
        t.up, t.c0, t.c1, t.dist, t.nMinor = _p(self.up), _p(self.c0), _p(self.c1), _p(self.dist), _p(self.nMinor)
        t.ent = _p(self.ent)
        for k in range(4):
            t.start[k] = self.start[k].ctypes.data
            t.len[k] = self.len[k].ctypes.data
        t.mut3, t.mutOff = _p(self.mut3), _p(self.mutOff)
        self.c = t


class Oracle:
    """The reference's functions, by name, evaluated by the C restatement."""

    def __init__(self, ref_idx, root_freqs, thresholdProb=1e-8, minBLenSensitivity=None,
                 thresholdDiffForUpdate=1e-5, thresholdFoldChangeUpdate=1.01, defaultBLen=0.000033):
        build()
        self.lib = C.CDLL(LIB)
        self.lib.omo_simplify.restype = C.c_int
        self.ref_idx = np.ascontiguousarray(ref_idx, dtype=np.uint8)
        self.lRef = len(self.ref_idx)
        self.m = OModel()
        m = self.m
        m.lRef = self.lRef
        m.refIdx = _p(self.ref_idx)
        for i in range(4):
            m.rootFreqs[i] = root_freqs[i]
        m.globalTotRate = -float(self.lRef)
        import sys
        m.minimumCarryOver = sys.float_info.min * 1e50
        m.thresholdProb = thresholdProb
        m.minBLenSensitivity = minBLenSensitivity if minBLenSensitivity is not None else 0.001 / self.lRef
        m.thresholdDiffForUpdate = thresholdDiffForUpdate
        m.thresholdFoldChangeUpdate = thresholdFoldChangeUpdate
        self.defaultBLen = defaultBLen
        self._keep = []
        self.u = False

    def set_model(self, Q, siteRates=None, usingErrorRate=False, errorRateGlobal=0.0, errorRates=None):
        m = self.m
        Qf = np.asarray(Q, dtype=np.float64).reshape(16)
        for i in range(16):
            m.Q[i] = Qf[i]
        self.sr = None if siteRates is None else np.ascontiguousarray(siteRates, dtype=np.float64)
        self.er = None if errorRates is None else np.ascontiguousarray(errorRates, dtype=np.float64)
        m.useRateVariation = 1 if self.sr is not None else 0
        m.usingErrorRate = 1 if usingErrorRate else 0
        m.errorRateSiteSpecific = 1 if (usingErrorRate and self.er is not None) else 0
        m.errorRate = errorRateGlobal if errorRateGlobal is not None else 0.0
        self.u = bool(usingErrorRate)
        # cumulativeRate (M:6350-6370) / cumulativeErrorRate, totError (M:6373-6390), plain Python floats
        lRef = self.lRef
        cr = [0.0] * (lRef + 1)
        nm = [Qf[0], Qf[5], Qf[10], Qf[15]]
        ri = self.ref_idx.tolist()
        if self.sr is not None:
            srl = self.sr.tolist()
            for i in range(lRef):
                cr[i + 1] = cr[i] + nm[ri[i]] * srl[i]
        else:
            for i in range(lRef):
                cr[i + 1] = cr[i] + nm[ri[i]]
        self.cr = np.asarray(cr, dtype=np.float64)
        self.cer = None
        m.totError = 0.0
        if usingErrorRate:
            if self.er is not None:
                ce = [0.0] * (lRef + 1)
                erl = self.er.tolist()
                for i in range(lRef):
                    ce[i + 1] = ce[i] + erl[i]
                self.cer = np.asarray(ce, dtype=np.float64)
                m.totError = -ce[-1]
            else:
                m.totError = -m.errorRate * lRef
        m.siteRates = _p(self.sr)
        m.errorRates = _p(self.er)
        m.cumulativeRate = _p(self.cr)
        m.cumulativeErrorRate = _p(self.cer)

    # ---- the reference's functions -------------------------------------------------------
    def appendProbNode(self, P, Cl, isTipC, bLen):
        a, b = to_entries(P, self.u), to_entries(Cl, self.u)
        out = C.c_double()
        rc = self.lib.omo_appendProbNode(C.byref(self.m), _p(a), len(a), _p(b), len(b), int(bool(isTipC)),
                                         C.c_double(bLen), C.byref(out))
        assert rc == 0
        return out.value

    def mergeVectors(self, pv1, b1, tip1, pv2, b2, tip2, returnLK=False, isUpDown=False, numMinor1=0, numMinor2=0):
        a, b = to_entries(pv1, self.u), to_entries(pv2, self.u)
        out = np.zeros(len(a) + len(b) + 2, dtype=OENTRY)
        lk = C.c_double()
        n = self.lib.omo_mergeVectors(C.byref(self.m), _p(a), len(a), C.c_double(b1), int(bool(tip1)), _p(b), len(b),
                                      C.c_double(b2), int(bool(tip2)), int(bool(returnLK)), int(bool(isUpDown)),
                                      int(numMinor1), int(numMinor2), _p(out), C.byref(lk))
        if n == -1:
            return None
        if n < 0:
            raise RuntimeError(f"mergeVectors fatal {n}")
        res = from_entries(out, n, self.u)
        return (res, lk.value) if returnLK else res

    def estimateBranchLengthWithDerivative(self, P, Cl, fromTipC=False):
        a, b = to_entries(P, self.u), to_entries(Cl, self.u)
        t = C.c_double()
        f = C.c_int()
        scratch = np.zeros(len(a) + len(b) + 2)
        self.lib.omo_estimateBranchLength(C.byref(self.m), _p(a), len(a), _p(b), len(b), int(bool(fromTipC)),
                                          C.byref(t), C.byref(f), _p(scratch))
        return False if f.value else t.value

    def areVectorsDifferent(self, pv1, pv2):
        if pv2 is None:
            return True
        a, b = to_entries(pv1, self.u), to_entries(pv2, self.u)
        return bool(self.lib.omo_areVectorsDifferent(C.byref(self.m), _p(a), len(a), _p(b), len(b)))

    def passGenomeListThroughBranch(self, pv, mutations, dirIsUp=False):
        a = to_entries(pv, self.u)
        mut = np.ascontiguousarray(np.asarray(mutations, dtype=np.int32).reshape(-1, 3))
        out = np.zeros(len(a) + 2 * len(mut) + 2, dtype=OENTRY)
        n = self.lib.omo_passGenomeListThroughBranch(C.byref(self.m), _p(a), len(a), _p(mut), len(mut),
                                                     int(bool(dirIsUp)), _p(out))
        return from_entries(out, n, self.u)

    def shorten(self, vec):
        a = to_entries(vec, self.u)
        n = self.lib.omo_shorten(C.byref(self.m), _p(a), len(a))
        return from_entries(a, n, self.u)

    @staticmethod
    def _path(pathMutations):
        off = [0]
        flat = []
        for ml in pathMutations:
            flat.extend(ml)
            off.append(len(flat))
        mut = np.ascontiguousarray(np.asarray(flat, dtype=np.int32).reshape(-1, 3))
        return mut, np.asarray(off, dtype=np.int32)

    def rootVector(self, pv, bLen, isFromTip, pathMutations):
        a = to_entries(pv, self.u)
        mut, off = self._path(pathMutations)
        cap = len(a) + 4 * len(mut) + 4
        out = np.zeros(cap, dtype=OENTRY)
        tmp = np.zeros(cap, dtype=OENTRY)
        n = self.lib.omo_rootVector(C.byref(self.m), _p(a), len(a), C.c_double(bLen), int(bool(isFromTip)), _p(mut),
                                    _p(off), len(off) - 1, _p(out), _p(tmp), cap)
        return from_entries(out, n, self.u)

    def getPartialVec(self, i12, totLen, mutMatrix, errorRate, vect=None, upNode=False, flag=False):
        M = np.ascontiguousarray(np.asarray(mutMatrix, dtype=np.float64).reshape(16))
        v = None if vect is None else np.ascontiguousarray(vect, dtype=np.float64)
        out = np.zeros(4)
        self.lib.omo_getPartialVec(C.byref(self.m), int(i12), C.c_double(totLen if totLen else 0.0), _p(M),
                                   C.c_double(errorRate if errorRate else 0.0), _p(v), int(bool(upNode)),
                                   int(bool(flag)), _p(out))
        return out.tolist()

    def simplify(self, vec, refA):
        v = np.ascontiguousarray(vec, dtype=np.float64)
        st = C.c_int()
        rc = self.lib.omo_simplify(C.byref(self.m), _p(v), int(refA), C.byref(st))
        if rc < 0:
            raise RuntimeError("simplify fatal")
        return st.value

    def evaluatePlacement(self, midTot, downVect, upVect, distance, removedPartials, isRemovedTip, fromTip1):
        a, d, u_, r = (to_entries(x, self.u) for x in (midTot, downVect, upVect, removedPartials))
        cap = len(d) + len(u_) + len(r) + 4
        tmp = np.zeros(3 * cap, dtype=OENTRY)
        scratch = np.zeros(len(a) + cap + 4)
        out = np.zeros(4)
        rc = self.lib.omo_evaluatePlacement(C.byref(self.m), _p(a), len(a), _p(d), len(d), _p(u_), len(u_),
                                            C.c_double(distance), _p(r), len(r), int(bool(isRemovedTip)),
                                            int(bool(fromTip1)), C.c_double(self.defaultBLen), _p(out), _p(tmp), cap,
                                            _p(scratch))
        if rc < 0:
            raise RuntimeError("evaluatePlacement fatal")
        return out.tolist()

    def findProbRoot(self, pv, pathMutations, rootFreqsLogErrorCumulative=None):
        a = to_entries(pv, self.u)
        mut, off = self._path(pathMutations)
        cap = len(a) + 4 * len(mut) + 4
        t1 = np.zeros(cap, dtype=OENTRY)
        t2 = np.zeros(cap, dtype=OENTRY)
        if not hasattr(self, "_cb"):
            cb = np.zeros((self.lRef + 1, 4), dtype=np.int32)
            ri = self.ref_idx
            oh = np.zeros((self.lRef, 4), dtype=np.int32)
            oh[np.arange(self.lRef), ri] = 1
            cb[1:] = np.cumsum(oh, axis=0)
            self._cb = np.ascontiguousarray(cb)
        rl = None
        if self.u:
            rf = [self.m.rootFreqs[i] for i in range(4)]
            acc = [0.0] * (self.lRef + 1)
            ri = self.ref_idx.tolist()
            erl = self.er.tolist() if self.er is not None else [self.m.errorRate] * self.lRef
            for i in range(self.lRef):
                acc[i + 1] = acc[i] + math.log(rf[ri[i]] * (1.0 - 1.33333 * erl[i]) + 0.333333 * erl[i])
            rl = np.asarray(acc)
        out = C.c_double()
        self.lib.omo_findProbRoot(C.byref(self.m), _p(a), len(a), _p(mut), _p(off), len(off) - 1, _p(self._cb), _p(rl),
                                  _p(t1), _p(t2), cap, C.byref(out))
        return out.value

    # ---- batch driver for bench.py's cpu_baseline leg -------------------------------------------
    def pack_many(self, lists):
        arrs = [to_entries(gl, self.u) for gl in lists]
        off = np.zeros(len(arrs) + 1, dtype=np.int64)
        np.cumsum([len(a) for a in arrs], out=off[1:])
        return np.concatenate(arrs), off

    def appendProbNode_batch(self, packed, parent_idx, child_idx, isTipC, bLen, threads=1):
        allent, off = packed
        pl = np.ascontiguousarray(parent_idx, dtype=np.int32)
        cl = np.ascontiguousarray(child_idx, dtype=np.int32)
        n = len(pl)
        tip = np.ascontiguousarray(np.broadcast_to(isTipC, n), dtype=np.uint8)
        bl = np.ascontiguousarray(np.broadcast_to(bLen, n), dtype=np.float64)
        out = np.zeros(n)
        if threads > 1:
            self.lib.omo_appendProbNode_batch_mt(C.byref(self.m), _p(allent), _p(off), n, _p(pl), _p(cl), _p(tip), _p(bl),
                                                 _p(out), int(threads))
        else:
            self.lib.omo_appendProbNode_batch(C.byref(self.m), _p(allent), _p(off), n, _p(pl), _p(cl), _p(tip), _p(bl), _p(out))
        return out

    # ---- SPR search -----------------------------------------------------------------------------
    def spr_worker(self, tree, nodes, *, strict, allowedFails, thresholdLogLKtopology, thresholdTopologyPlacement,
                   thresholdLogLKoptimizationTopology, thresholdLogLKconsecutivePlacement, effectivelyNon0BLen,
                   arena_mb=256, want_removed_partials=False, threads=1):
        """startTopologyUpdatesParallel's worker body (M:9615-9711) for the pruned nodes `nodes`."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        n = len(nodes)
        sp = OSearchParamsC(int(bool(strict)), int(allowedFails), thresholdLogLKtopology, thresholdTopologyPlacement,
                            thresholdLogLKoptimizationTopology, thresholdLogLKconsecutivePlacement, effectivelyNon0BLen,
                            self.defaultBLen)
        res = (OSearchResultC * n)()
        bufs = []
        if want_removed_partials:
            for i in range(n):
                b = np.zeros(4096, dtype=OENTRY)
                bufs.append(b)
                res[i].rpr = b.ctypes.data
                res[i].rprCap = 4096
        threads = max(1, int(threads))
        arena = np.empty((arena_mb << 20) * threads, dtype=np.uint8)
        self.lib.omo_sprWorker_mt(C.byref(self.m), C.byref(tree.c), C.byref(sp), n, _p(nodes), res, _p(arena),
                                  C.c_size_t(arena_mb << 20), threads)
        out = dict(bestNode=np.array([r.bestNode for r in res]), placement=np.array([r.placement for r in res]),
                   status=np.array([r.status for r in res]), nAppend=np.array([r.nAppend for r in res]),
                   bestScore=np.array([r.bestScore for r in res]), improvement=np.array([r.improvement for r in res]),
                   currentLK=np.array([r.currentLK for r in res]), blen=np.array([[r.blen[0], r.blen[1], r.blen[2]] for r in res]))
        if want_removed_partials:
            out["removedPartials"] = [from_entries(bufs[i], res[i].rprN, self.u) if res[i].status == 0 else None
                                      for i in range(n)]
        return out
