"""ctypes binding of libmaple_hip.so (the C ABI in include/maple_hip.h).

This is the only way the host reaches the GPU kernels.  There is no CPU
fallback: if the shared library is missing or no MI355X is visible,
construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .genome_list import PackedLists, pack_lists, pack_mutations, unpack_list

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmaple_hip.so")

EXPORTS = [
    "maple_abi_version", "maple_create", "maple_destroy", "maple_set_tuning", "maple_last_error", "maple_set_model", "maple_get_model",
    "maple_lists_upload", "maple_lists_update", "maple_lists_sizes", "maple_lists_download", "maple_arena_mark", "maple_arena_release",
    "maple_arena_stats", "maple_mutations_upload", "maple_append_batch", "maple_merge_batch", "maple_blen_batch",
    "maple_differ_batch", "maple_pass_branch_batch", "maple_shorten_batch", "maple_root_vector_batch",
    "maple_evaluate_placement_batch", "maple_update_partials", "maple_update_partials_touched", "maple_tree_patch", "maple_append_batch_dev", "maple_append_queries_dev", "maple_timing_reset", "maple_timing_read", "maple_timing_read_each",
    "maple_append_algorithmic_bytes", "maple_tree_upload", "maple_spr_search_batch", "maple_minor_batch", "maple_root_prob_batch", "maple_candset_create", "maple_append_candset",
    "maple_minor_candset", "maple_placement_search_batch", "maple_placement_prepare", "maple_placement_ahead", "maple_placement_ahead_stats", "maple_set_fatal_policy",
    "maple_timing_read_kind", "maple_placement_supports_batch",
    "maple_candset_destroy", "maple_spr_search_visited", "maple_arena_compact", "maple_append_queries_argmax_dev", "maple_comm_unique_id", "maple_comm_init", "maple_argmax_allreduce_dev", "maple_tree_rebuild_lists",
]


# include/maple_hip_debug.h: measurement aids and test hooks, exported by libmaple_hip_debug.so only (Device(..., debug=True))
DEBUG_EXPORTS = ["maple_debug_wave_append_batch", "maple_debug_trace_query", "maple_debug_trace_read", "maple_debug_calib_walk",
                 "maple_debug_calib_write", "maple_debug_gpv_batch", "maple_debug_simplify_batch", "maple_debug_frontier_levels"]
LIB_PATH_DEBUG = os.path.join(HERE, "libmaple_hip_debug.so")


class MapleParams(C.Structure):
    _fields_ = [("thresholdProb", C.c_double), ("minBLenSensitivity", C.c_double),
                ("thresholdDiffForUpdate", C.c_double), ("thresholdFoldChangeUpdate", C.c_double),
                ("defaultBLen", C.c_double)]


class MapleSearchParams(C.Structure):
    _fields_ = [("strictTopologyStopRules", C.c_int32), ("allowedFailsTopology", C.c_int32),
                ("thresholdLogLKtopology", C.c_double), ("thresholdTopologyPlacement", C.c_double),
                ("thresholdLogLKoptimizationTopology", C.c_double), ("thresholdLogLKconsecutivePlacement", C.c_double),
                ("effectivelyNon0BLen", C.c_double), ("wideSearchBudget", C.c_int32), ("searchTier", C.c_int32)]


class MapleTuning(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("wavePerItemMax", C.c_int32), ("placementChunkMax", C.c_int32), ("noCladeScan", C.c_int32), ("verbose", C.c_int32),
                ("wideOutsideFrontier", C.c_int32), ("denseWideScoring", C.c_int32), ("waveAllBelow", C.c_int32), ("noOverHint", C.c_int32), ("noAheadExpansion", C.c_int32), ("noAheadSpeculation", C.c_int32)]


class MaplePlacementParams(C.Structure):
    _fields_ = [("oneMutBLen", C.c_double), ("effectivelyNon0BLen", C.c_double), ("thresholdLogLK", C.c_double),
                ("thresholdLogLKoptimization", C.c_double), ("thresholdLogLKconsecutivePlacement", C.c_double),
                ("allowedFails", C.c_int32), ("strictStopRules", C.c_int32), ("onlyFindIdentical", C.c_int32)]


class MapleError(RuntimeError):
    """An error status of libmaple_hip.so; ``code`` is the MAPLE_ERR_* value (include/maple_hip.h)."""
    ERR_ARG, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_FATAL = -1, -2, -3, -4, -5

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


_lib = None
_lib_debug = None


def load_library(debug=False):
    """Load libmaple_hip.so (``debug``: libmaple_hip_debug.so, the same library with the entry points of
    include/maple_hip_debug.h); raises if it has not been built (see __graft_entry__.build)."""
    global _lib, _lib_debug
    if (_lib_debug if debug else _lib) is None:
        default = LIB_PATH_DEBUG if debug else LIB_PATH
        # (kernel-variant experiments: another build of the library.  MAPLE_HIP_LIB names a product build and never the debug
        # one -- a product variant has no maple_debug_* symbols, a debug build behind the product handle would carry the debug ABI;
        # MAPLE_HIP_LIB_DEBUG names a debug variant)
        lib_path = os.environ.get("MAPLE_HIP_LIB_DEBUG" if debug else "MAPLE_HIP_LIB", default)
        if not os.path.exists(lib_path):
            raise MapleError(f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`; "
                             "there is no CPU fallback for the placement path")
        # PyTorch-ROCm ships its own HIP runtime; whichever copy of libamdhip64 is loaded first serves the whole
        # process, and torch cannot see the GPU if the system copy got there before it.  Load torch's first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(lib_path)
        lib.maple_last_error.restype = C.c_char_p
        for name in EXPORTS + (DEBUG_EXPORTS if debug else []):
            if not hasattr(lib, name):       # fail early if a declared symbol is not exported
                raise MapleError(f"{lib_path} does not export {name} (include/maple_hip{'_debug' if name in DEBUG_EXPORTS else ''}.h): "
                                 "rebuild it with __graft_entry__.build()")
        if debug:
            _lib_debug = lib
        else:
            _lib = lib
    return _lib_debug if debug else _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def _u8(x):
    return np.ascontiguousarray(x, dtype=np.uint8)


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class Device:
    """One GPU context: model tables, the genome-list arena and the batched operators."""

    def __init__(self, ref_idx, root_freqs, *, device=0, thresholdProb=1e-8, minBLenSensitivity=None,
                 thresholdDiffForUpdate=1e-5, thresholdFoldChangeUpdate=1.01, defaultBLen=0.000033,
                 arena_bytes=0, lib=None, debug=False):
        # (lib: another library with the same C ABI, handed in explicitly -- the tests diff libmaple_hip.so against its CPU twin
        # this way; nothing in the package ever passes one)
        # (debug: libmaple_hip_debug.so -- the same library plus the measurement aids and test hooks of include/maple_hip_debug.h)
        self.lib = lib if lib is not None else load_library(debug)
        self.ref_idx = _u8(ref_idx)
        self.lRef = int(len(self.ref_idx))
        if minBLenSensitivity is None:
            minBLenSensitivity = 0.001 / self.lRef
        p = MapleParams(thresholdProb, minBLenSensitivity, thresholdDiffForUpdate, thresholdFoldChangeUpdate, defaultBLen)
        rf = _f64(root_freqs)
        h = C.c_void_p()
        rc = self.lib.maple_create(C.byref(h), int(device), self.lRef, _ptr(self.ref_idx), _ptr(rf), C.byref(p),
                                   C.c_uint64(arena_bytes))
        if rc != 0:
            raise MapleError(f"maple_create failed ({rc}): no usable MI355X / HIP runtime; the placement path has no "
                             "CPU fallback")
        self.h = h
        self.device = int(device)
        self.u = False

    # -- plumbing --------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.maple_last_error(self.h)
            raise MapleError(f"libmaple_hip error {rc}: {msg.decode() if msg else ''}", rc)

    def close(self):
        if getattr(self, "h", None):
            self.lib.maple_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model -------------------------------------------------------------------------------
    def set_tuning(self, *, wave_per_item_max=0, placement_chunk_max=0, no_clade_scan=False, verbose=0, wide_outside_frontier=False,
                   dense_wide_scoring=False, wave_all_below=0, no_over_hint=False, no_ahead_expansion=False, no_ahead_speculation=False):
        """How the library schedules its work (never what it computes): see maple_tuning in include/maple_hip.h."""
        t = MapleTuning(C.sizeof(MapleTuning), int(wave_per_item_max), int(placement_chunk_max), int(bool(no_clade_scan)), int(verbose),
                        int(bool(wide_outside_frontier)), int(bool(dense_wide_scoring)), int(wave_all_below), int(bool(no_over_hint)),
                        int(no_ahead_expansion), int(bool(no_ahead_speculation)))
        self._ck(self.lib.maple_set_tuning(self.h, C.byref(t)))

    def set_model(self, Q, siteRates=None, usingErrorRate=False, errorRateGlobal=0.0, errorRates=None):
        Qf = _f64(np.asarray(Q, dtype=np.float64).reshape(16))
        sr = None if siteRates is None else _f64(siteRates)
        er = None if (errorRates is None or not usingErrorRate) else _f64(errorRates)
        self._ck(self.lib.maple_set_model(self.h, _ptr(Qf), _ptr(sr), int(bool(usingErrorRate)),
                                          C.c_double(errorRateGlobal or 0.0), _ptr(er)))
        self.u = bool(usingErrorRate)

    def get_model(self):
        cr = np.zeros(self.lRef + 1)
        ce = np.zeros(self.lRef + 1)
        te = C.c_double()
        self._ck(self.lib.maple_get_model(self.h, _ptr(cr), _ptr(ce), C.byref(te)))
        return cr, ce, te.value

    # -- lists -------------------------------------------------------------------------------
    def upload_packed(self, pl: PackedLists):
        first = C.c_int32()
        self._ck(self.lib.maple_lists_upload(self.h, len(pl), _ptr(pl.ent_off), _ptr(pl.pos), _ptr(pl.meta),
                                             _ptr(pl.aux_off), _ptr(pl.aux), C.byref(first)))
        return np.arange(first.value, first.value + len(pl), dtype=np.int32)

    def update_lists(self, ids, lists):
        """maple_lists_update: new contents (reference tuple form) for the existing list ids; the ids stay valid."""
        pl = pack_lists(lists, self.u)
        ids = _i32(ids)
        assert len(ids) == len(pl)
        self._ck(self.lib.maple_lists_update(self.h, len(pl), _ptr(ids), _ptr(pl.ent_off), _ptr(pl.pos), _ptr(pl.meta),
                                             _ptr(pl.aux_off), _ptr(pl.aux)))

    def upload(self, lists):
        """Upload genome lists given in the reference's tuple form; returns their ids."""
        return self.upload_packed(pack_lists(lists, self.u))

    def sizes(self, ids):
        ids = _i32(ids)
        ne = np.zeros(len(ids), dtype=np.int32)
        na = np.zeros(len(ids), dtype=np.int32)
        self._ck(self.lib.maple_lists_sizes(self.h, len(ids), _ptr(ids), _ptr(ne), _ptr(na)))
        return ne, na

    def download_packed(self, ids) -> PackedLists:
        """Download lists (all ids >= 0) in the packed CSR form."""
        good = _i32(ids)
        ne, na = self.sizes(good)
        eo = np.zeros(len(good) + 1, dtype=np.int64)
        ao = np.zeros(len(good) + 1, dtype=np.int64)
        np.cumsum(ne, out=eo[1:])
        np.cumsum(na, out=ao[1:])
        pos = np.zeros(max(1, eo[-1]), dtype=np.int32)
        meta = np.zeros(max(1, eo[-1]), dtype=np.uint32)
        aux = np.zeros(max(1, ao[-1]), dtype=np.float64)
        if len(good):
            self._ck(self.lib.maple_lists_download(self.h, len(good), _ptr(good), _ptr(eo), _ptr(pos), _ptr(meta),
                                                   _ptr(ao), _ptr(aux)))
        return PackedLists(eo, pos, meta, ao, aux)

    def download(self, ids):
        """Download lists by id into the reference's tuple form (None for id -1)."""
        ids = _i32(ids)
        good = ids[ids >= 0]
        pk = self.download_packed(good)
        eo, ao, pos, meta, aux = pk.ent_off, pk.aux_off, pk.pos, pk.meta, pk.aux
        out, k = [], 0
        for i in ids:
            if i < 0:
                out.append(None)
            else:
                out.append(unpack_list(pos[eo[k]:eo[k + 1]], meta[eo[k]:eo[k + 1]], aux[ao[k]:ao[k + 1]], self.u))
                k += 1
        return out

    def set_fatal_policy(self, tolerate: bool):
        """tolerate=True: an item that hits a state the reference raises on gets list id -2 instead of failing its batch."""
        self._ck(self.lib.maple_set_fatal_policy(self.h, int(bool(tolerate))))

    def mark(self):
        m = C.c_int64()
        self._ck(self.lib.maple_arena_mark(self.h, C.byref(m)))
        return m.value

    def release(self, mark):
        self._ck(self.lib.maple_arena_release(self.h, C.c_int64(mark)))

    def stats(self):
        a, b, c_, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.maple_arena_stats(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d)))
        return dict(n_lists=a.value, n_entries=b.value, n_aux=c_.value, cap_entries=d.value)

    def arena_compact(self, live):
        """Keep only the lists `live` (-1 entries stay -1): returns their new ids.  Every other id, mark, candidate set and the
        uploaded tree are gone afterwards (maple_arena_compact)."""
        live = _i32(live)
        new = np.zeros(len(live), np.int32)
        self._ck(self.lib.maple_arena_compact(self.h, C.c_int64(len(live)), _ptr(live), _ptr(new)))
        return new

    def upload_mutations(self, mut_lists):
        off, mut3 = pack_mutations(mut_lists)
        first = C.c_int32()
        self._ck(self.lib.maple_mutations_upload(self.h, len(off) - 1, _ptr(off), _ptr(mut3), C.byref(first)))
        return np.arange(first.value, first.value + len(off) - 1, dtype=np.int32)

    # -- batched operators ----------------------------------------------------------------------
    def append_batch(self, parent, child, isTipC, bLen):
        parent, child = _i32(parent), _i32(child)
        n = len(parent)
        tip = _u8(np.broadcast_to(isTipC, n))
        bl = _f64(np.broadcast_to(bLen, n))
        out = np.zeros(n)
        self._ck(self.lib.maple_append_batch(self.h, n, _ptr(parent), _ptr(child), _ptr(tip), _ptr(bl), _ptr(out)))
        return out

    def merge_batch(self, l1, b1, tip1, l2, b2, tip2, isUpDown, returnLK=False, numMinor1=None, numMinor2=None):
        l1, l2 = _i32(l1), _i32(l2)
        n = len(l1)
        b1, b2 = _f64(np.broadcast_to(b1, n)), _f64(np.broadcast_to(b2, n))
        t1, t2 = _u8(np.broadcast_to(tip1, n)), _u8(np.broadcast_to(tip2, n))
        ud = _u8(np.broadcast_to(isUpDown, n))
        nm1 = None if numMinor1 is None else _i32(np.broadcast_to(numMinor1, n))
        nm2 = None if numMinor2 is None else _i32(np.broadcast_to(numMinor2, n))
        out = np.zeros(n, dtype=np.int32)
        lk = np.zeros(n) if returnLK else None
        self._ck(self.lib.maple_merge_batch(self.h, n, _ptr(l1), _ptr(b1), _ptr(t1), _ptr(l2), _ptr(b2), _ptr(t2),
                                            _ptr(ud), _ptr(nm1), _ptr(nm2), _ptr(out), _ptr(lk)))
        return (out, lk) if returnLK else out

    def blen_batch(self, parent, child, fromTipC):
        parent, child = _i32(parent), _i32(child)
        n = len(parent)
        tip = _u8(np.broadcast_to(fromTipC, n))
        t = np.zeros(n)
        f = np.zeros(n, dtype=np.uint8)
        self._ck(self.lib.maple_blen_batch(self.h, n, _ptr(parent), _ptr(child), _ptr(tip), _ptr(t), _ptr(f)))
        return t, f.astype(bool)

    def differ_batch(self, l1, l2):
        l1, l2 = _i32(l1), _i32(l2)
        out = np.zeros(len(l1), dtype=np.uint8)
        self._ck(self.lib.maple_differ_batch(self.h, len(l1), _ptr(l1), _ptr(l2), _ptr(out)))
        return out.astype(bool)

    def candset_create(self, lists, frame_idx, n_frames):
        lists, frame_idx = _i32(lists), _i32(frame_idx)
        sid = C.c_int32()
        self._ck(self.lib.maple_candset_create(self.h, len(lists), _ptr(lists), _ptr(frame_idx), int(n_frames), C.byref(sid)))
        return sid.value, len(lists)

    def candset_destroy(self, cset):
        self._ck(self.lib.maple_candset_destroy(self.h, int(cset[0])))

    def append_candset(self, cset, frame_lists, isTipC, bLen):
        fl = _i32(frame_lists)
        out = np.zeros(cset[1])
        self._ck(self.lib.maple_append_candset(self.h, cset[0], _ptr(fl), int(bool(isTipC)), C.c_double(bLen), _ptr(out)))
        return out

    def minor_candset(self, cset, frame_lists, onlyFindIdentical=False):
        fl = _i32(frame_lists)
        out = np.zeros(cset[1], dtype=np.uint8)
        self._ck(self.lib.maple_minor_candset(self.h, cset[0], _ptr(fl), int(bool(onlyFindIdentical)), _ptr(out)))
        return out

    def root_prob_batch(self, lists):
        lists = _i32(lists)
        out = np.zeros(len(lists))
        self._ck(self.lib.maple_root_prob_batch(self.h, len(lists), _ptr(lists), _ptr(out)))
        return out

    def minor_batch(self, l1, l2, onlyFindIdentical=False):
        l1, l2 = _i32(l1), _i32(l2)
        out = np.zeros(len(l1), dtype=np.uint8)
        self._ck(self.lib.maple_minor_batch(self.h, len(l1), _ptr(l1), _ptr(l2), int(bool(onlyFindIdentical)), _ptr(out)))
        return out

    def pass_branch_batch(self, lists, mutLists, dirIsUp):
        lists, mutLists = _i32(lists), _i32(mutLists)
        n = len(lists)
        up = _u8(np.broadcast_to(dirIsUp, n))
        out = np.zeros(n, dtype=np.int32)
        self._ck(self.lib.maple_pass_branch_batch(self.h, n, _ptr(lists), _ptr(mutLists), _ptr(up), _ptr(out)))
        return out

    def shorten_batch(self, lists):
        lists = _i32(lists)
        out = np.zeros(len(lists), dtype=np.int32)
        self._ck(self.lib.maple_shorten_batch(self.h, len(lists), _ptr(lists), _ptr(out)))
        return out

    def root_vector_batch(self, lists, bLen, isFromTip, paths):
        """paths[i] = mutation-list ids on the walk node -> root (node first)."""
        lists = _i32(lists)
        n = len(lists)
        bl = _f64(np.broadcast_to(bLen, n))
        tip = _u8(np.broadcast_to(isFromTip, n))
        off = np.zeros(n + 1, dtype=np.int64)
        flat = []
        for i, p in enumerate(paths):
            flat.extend(int(x) for x in p)
            off[i + 1] = len(flat)
        pm = _i32(flat if flat else [0])
        out = np.zeros(n, dtype=np.int32)
        self._ck(self.lib.maple_root_vector_batch(self.h, n, _ptr(lists), _ptr(bl), _ptr(tip), _ptr(off), _ptr(pm),
                                                  _ptr(out)))
        return out

    def tree_patch(self, n_total, nodes, up, c0, c1, dist, tip, lower, up_right, up_left, tot_up):
        """maple_tree_patch: the records of the touched (and new) nodes; see include/maple_hip.h."""
        nodes = _i32(nodes)
        cols = [_i32(x) for x in (up, c0, c1)] + [_f64(dist), _u8(tip)] + [_i32(x) for x in (lower, up_right, up_left, tot_up)]
        assert all(len(x) == len(nodes) for x in cols)
        self._ck(self.lib.maple_tree_patch(self.h, int(n_total), len(nodes), _ptr(nodes), *[_ptr(x) for x in cols]))

    def update_partials(self, root, up, c0, c1, tip, mut, depth, dist, lower, up_right, up_left, tot_up, changed):
        """maple_update_partials: the four list-id columns and ``dist`` are updated IN PLACE (they must be C-contiguous
        int32 / float64 arrays -- nothing is copied); returns the number of lists replaced."""
        for name, arr, dt in (("up", up, np.int32), ("child0", c0, np.int32), ("child1", c1, np.int32), ("isTip", tip, np.uint8),
                              ("mutList", mut, np.int32), ("depth", depth, np.int32), ("dist", dist, np.float64),
                              ("lower", lower, np.int32), ("upRight", up_right, np.int32), ("upLeft", up_left, np.int32),
                              ("totUp", tot_up, np.int32)):
            if not (isinstance(arr, np.ndarray) and arr.dtype == dt and arr.flags.c_contiguous and len(arr) == len(up)):
                raise ValueError(f"update_partials: column {name} must be a C-contiguous {np.dtype(dt).name} array of {len(up)}")
        ch = _i32(changed)
        n_rep = C.c_int32(0)
        self._ck(self.lib.maple_update_partials(self.h, len(up), int(root), _ptr(up), _ptr(c0), _ptr(c1), _ptr(tip), _ptr(mut),
                                                _ptr(depth), _ptr(dist), _ptr(lower), _ptr(up_right), _ptr(up_left), _ptr(tot_up),
                                                len(ch), _ptr(ch), C.byref(n_rep)))
        return int(n_rep.value)

    def debug_wave_append_batch(self, pl, cl, isTipC, bLen):
        """appendProbNode with one wavefront per pair (maple_debug_wave_append_batch); returns (values, kernel ms)."""
        pl, cl = _i32(pl), _i32(cl)
        n = len(pl)
        tip = _u8(np.broadcast_to(isTipC, n))
        bl = _f64(np.broadcast_to(bLen, n))
        out = np.zeros(n)
        ms = C.c_float(0.0)
        self._ck(self.lib.maple_debug_wave_append_batch(self.h, n, _ptr(pl), _ptr(cl), _ptr(tip), _ptr(bl), _ptr(out), C.byref(ms)))
        return out, float(ms.value)

    def update_partials_touched(self, cap=65536):
        """Nodes whose lists / branch length the last update_partials replaced (ascending)."""
        out = np.zeros(cap, dtype=np.int32)
        n = C.c_int32(0)
        self._ck(self.lib.maple_update_partials_touched(self.h, cap, _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def evaluate_placement_batch(self, midTot, down, up, distance, removed, isRemovedTip, fromTip1):
        midTot, down, up, removed = _i32(midTot), _i32(down), _i32(up), _i32(removed)
        n = len(midTot)
        dist = _f64(np.broadcast_to(distance, n))
        rt, ft = _u8(np.broadcast_to(isRemovedTip, n)), _u8(np.broadcast_to(fromTip1, n))
        out = np.zeros((n, 4))
        self._ck(self.lib.maple_evaluate_placement_batch(self.h, n, _ptr(midTot), _ptr(down), _ptr(up), _ptr(dist),
                                                         _ptr(removed), _ptr(rt), _ptr(ft), _ptr(out)))
        return out

    # -- tree mirror + device-resident SPR search ------------------------------------------------------
    def upload_tree(self, root, up, child0, child1, dist, isTip, lower, upRight, upLeft, totUp, mutList):
        n = len(up)
        arrs = [_i32(x) for x in (up, child0, child1)]
        d, tip = _f64(dist), _u8(isTip)
        ls = [_i32(x) for x in (lower, upRight, upLeft, totUp, mutList)]
        self._ck(self.lib.maple_tree_upload(self.h, n, int(root), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(d),
                                            _ptr(tip), _ptr(ls[0]), _ptr(ls[1]), _ptr(ls[2]), _ptr(ls[3]), _ptr(ls[4])))
        self.n_nodes = n

    def spr_search_batch(self, nodes, *, strict, allowedFails, thresholdLogLKtopology, thresholdTopologyPlacement,
                         thresholdLogLKoptimizationTopology, thresholdLogLKconsecutivePlacement, effectivelyNon0BLen,
                         ws_entries_per_lane=0, want_removed_partials=False, wide_search_budget=0, search_tier=0):
        """startTopologyUpdatesParallel's worker body (M:9615-9711) for `nodes`, searches run on the GPU."""
        nodes = _i32(nodes)
        n = len(nodes)
        sp = MapleSearchParams(int(bool(strict)), int(allowedFails), thresholdLogLKtopology, thresholdTopologyPlacement,
                               thresholdLogLKoptimizationTopology, thresholdLogLKconsecutivePlacement,
                               effectivelyNon0BLen, int(wide_search_budget), int(search_tier))
        out = dict(bestNode=np.zeros(n, np.int32), bestScore=np.zeros(n), blen=np.zeros((n, 3)),
                   placement=np.zeros(n, np.int32), improvement=np.zeros(n), currentLK=np.zeros(n),
                   nAppend=np.zeros(n, np.int32), status=np.zeros(n, np.int32))
        rpr = np.zeros(n, np.int32) if want_removed_partials else None
        self._ck(self.lib.maple_spr_search_batch(self.h, n, _ptr(nodes), C.byref(sp), int(ws_entries_per_lane),
                                                 _ptr(out["bestNode"]), _ptr(out["bestScore"]), _ptr(out["blen"]),
                                                 _ptr(out["placement"]), _ptr(out["improvement"]),
                                                 _ptr(out["currentLK"]), _ptr(out["nAppend"]), _ptr(out["status"]),
                                                 _ptr(rpr)))
        if want_removed_partials:
            out["removedPartials"] = rpr
        return out

    def spr_search_visited(self, cap=1 << 22):
        """(query index, node) pairs of every branch the searches of the last spr_search_batch may have read (see the header).
        If they do not fit in ``cap`` the call is repeated with the room it asked for."""
        for _ in range(2):
            q = np.zeros(cap, np.int32)
            v = np.zeros(cap, np.int32)
            n = C.c_int64()
            rc = self.lib.maple_spr_search_visited(self.h, C.c_int64(cap), _ptr(q), _ptr(v), C.byref(n))
            if rc == MapleError.ERR_ARG and n.value > cap:
                cap = int(n.value)
                continue
            self._ck(rc)
            return q[: n.value], v[: n.value]
        self._ck(rc)

    def placement_prepare(self, *, oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                          thresholdLogLKconsecutivePlacement, allowedFails=5, strictStopRules=True, onlyFindIdentical=False):
        """Per-tree tables + root vector of the placement search, created outside any arena mark of the caller."""
        pp = MaplePlacementParams(oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                                  thresholdLogLKconsecutivePlacement, int(allowedFails), int(bool(strictStopRules)),
                                  int(bool(onlyFindIdentical)))
        self._ck(self.lib.maple_placement_prepare(self.h, C.byref(pp)))

    def placement_ahead(self, q_lists, *, oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                        thresholdLogLKconsecutivePlacement, allowedFails=5, strictStopRules=True, onlyFindIdentical=False):
        """Announce the next samples of a serial placement loop (maple_placement_ahead): their score rows are made in one launch
        and kept current under tree_patch; single-sample placement_search_batch calls for them, in this order and with these
        parameters, then skip the scoring.  Returns how many of the leading samples got rows (0: nothing changes)."""
        q = _i32(q_lists)
        pp = MaplePlacementParams(oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                                  thresholdLogLKconsecutivePlacement, int(allowedFails), int(bool(strictStopRules)),
                                  int(bool(onlyFindIdentical)))
        taken = C.c_int32(0)
        self._ck(self.lib.maple_placement_ahead(self.h, len(q), _ptr(q), C.byref(pp), C.byref(taken)))
        return taken.value

    def placement_ahead_stats(self):
        out = np.zeros(5, dtype=np.int64)
        self._ck(self.lib.maple_placement_ahead_stats(self.h, _ptr(out)))
        return dict(zip(("searches", "fallbacks", "expanded_items", "traversals_ahead_used", "traversals_ahead_dropped"), (int(x) for x in out)))

    def placement_search_batch(self, q_lists, *, oneMutBLen, effectivelyNon0BLen, thresholdLogLK,
                               thresholdLogLKoptimization, thresholdLogLKconsecutivePlacement, allowedFails=5,
                               strictStopRules=True, onlyFindIdentical=False):
        """findBestParentForNewSample (M:7912-8292) for many query lists against the uploaded (frozen) tree."""
        q = _i32(q_lists)
        n = len(q)
        pp = MaplePlacementParams(oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                                  thresholdLogLKconsecutivePlacement, int(allowedFails), int(bool(strictStopRules)),
                                  int(bool(onlyFindIdentical)))
        out = dict(bestNode=np.zeros(n, np.int32), bestScore=np.zeros(n), blen=np.zeros((n, 3)),
                   bestDiffs=np.zeros(n, np.int32), nAppend=np.zeros(n, np.int32), status=np.zeros(n, np.int32))
        self._ck(self.lib.maple_placement_search_batch(self.h, n, _ptr(q), C.byref(pp), _ptr(out["bestNode"]),
                                                       _ptr(out["bestScore"]), _ptr(out["blen"]), _ptr(out["bestDiffs"]),
                                                       _ptr(out["nAppend"]), _ptr(out["status"])))
        return out

    def placement_supports_batch(self, q_lists, *, oneMutBLen, effectivelyNon0BLen, thresholdLogLK,
                                 thresholdLogLKoptimization, thresholdLogLKconsecutivePlacement,
                                 thresholdLogLKoptimizationTopology, minBranchSupport=0.01, allowedFails=5,
                                 strictStopRules=True, onlyFindIdentical=False):
        """findBestParentForNewSample(..., computePlacementSupportOnly=True) (M:8101-8290) for many query lists: per query
        (possiblePlacements = [(node, support, (top, bottom, appending)), ...], list id of bestPlacementTotalLh or -1)."""
        q = _i32(q_lists)
        n = len(q)
        pp = MaplePlacementParams(oneMutBLen, effectivelyNon0BLen, thresholdLogLK, thresholdLogLKoptimization,
                                  thresholdLogLKconsecutivePlacement, int(allowedFails), int(bool(strictStopRules)),
                                  int(bool(onlyFindIdentical)))
        cap = 128 * max(1, n)
        off = np.zeros(n + 1, np.int64)
        node, supp, bl = np.zeros(cap, np.int32), np.zeros(cap), np.zeros((cap, 3))
        best, status = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._ck(self.lib.maple_placement_supports_batch(self.h, n, _ptr(q), C.byref(pp),
                                                         C.c_double(thresholdLogLKoptimizationTopology),
                                                         C.c_double(minBranchSupport), C.c_int64(cap), _ptr(off), _ptr(node),
                                                         _ptr(supp), _ptr(bl), _ptr(best), _ptr(status)))
        out = []
        for g in range(n):
            a, b = off[g], off[g + 1]
            out.append(([(int(node[i]), float(supp[i]), tuple(float(x) for x in bl[i])) for i in range(a, b)], int(best[g])))
        return out, status

    def debug_gpv_batch(self, i12, totLen, mutMatrix, errorRate, vect, upNode, flag):
        """getPartialVec (M:4073-4141) for n calls, each with its own 4x4 matrix; vect rows are read where i12 == 6."""
        i12 = _i32(i12)
        n = len(i12)
        M = _f64(np.asarray(mutMatrix, dtype=np.float64).reshape(n, 16))
        v = _f64(np.asarray(vect, dtype=np.float64).reshape(n, 4))
        out = np.zeros((n, 4))
        self._ck(self.lib.maple_debug_gpv_batch(self.h, n, _ptr(i12), _ptr(_f64(totLen)), _ptr(M), _ptr(_f64(errorRate)), _ptr(v),
                                                _ptr(_u8(upNode)), _ptr(_u8(flag)), _ptr(out)))
        return out

    def debug_simplify_batch(self, vec, refA):
        v = _f64(np.asarray(vec, dtype=np.float64).reshape(-1, 4))
        out = np.zeros(len(v), dtype=np.int32)
        self._ck(self.lib.maple_debug_simplify_batch(self.h, len(v), _ptr(v), _ptr(_i32(refA)), _ptr(out)))
        return out

    def debug_calib_walk(self, nbytes, repeats=1):
        ms = C.c_float()
        self._ck(self.lib.maple_debug_calib_walk(self.h, C.c_uint64(nbytes), int(repeats), C.byref(ms)))
        return ms.value

    def debug_calib_write(self, nbytes, mode, repeats=1):
        ms = C.c_float()
        self._ck(self.lib.maple_debug_calib_write(self.h, C.c_uint64(nbytes), int(mode), int(repeats), C.byref(ms)))
        return ms.value

    def debug_trace_query(self, q):
        self._ck(self.lib.maple_debug_trace_query(self.h, int(q)))

    def debug_trace_read(self):
        n = C.c_int32()
        it = np.zeros(4 * 4096, dtype=np.int32)
        va = np.zeros(2 * 4096, dtype=np.float64)
        self._ck(self.lib.maple_debug_trace_read(self.h, C.byref(n), _ptr(it), _ptr(va)))
        k = min(n.value, 4096)
        return it[: 4 * k].reshape(k, 4), va[: 2 * k].reshape(k, 2)

    # -- device-resident forms (pointers into HBM, e.g. torch tensors' data_ptr()) ------------------
    def append_batch_dev(self, n, parent_ptr, child_ptr, tip_ptr, blen_ptr, out_ptr, stream=0):
        self._ck(self.lib.maple_append_batch_dev(self.h, int(n), C.c_void_p(parent_ptr), C.c_void_p(child_ptr),
                                                 C.c_void_p(tip_ptr), C.c_void_p(blen_ptr), C.c_void_p(out_ptr),
                                                 C.c_void_p(stream)))

    def append_queries_dev(self, nQ, qlist_ptr, nC, cand_ptr, isTipC, bLen, out_ptr, stream=0):
        self._ck(self.lib.maple_append_queries_dev(self.h, int(nQ), C.c_void_p(qlist_ptr), int(nC), C.c_void_p(cand_ptr),
                                                   int(bool(isTipC)), C.c_double(bLen), C.c_void_p(out_ptr),
                                                   C.c_void_p(stream)))

    def append_queries_argmax_dev(self, nQ, qlist_ptr, nC, cand_ptr, rank_ptr, isTipC, bLen, best_score_ptr, best_idx_ptr, stream=0):
        self._ck(self.lib.maple_append_queries_argmax_dev(self.h, int(nQ), C.c_void_p(qlist_ptr), int(nC), C.c_void_p(cand_ptr),
                                                          C.c_void_p(rank_ptr), int(bool(isTipC)), C.c_double(bLen),
                                                          C.c_void_p(best_score_ptr), C.c_void_p(best_idx_ptr), C.c_void_p(stream)))

    def comm_unique_id(self):
        buf = np.zeros(128, dtype=np.uint8)
        self._ck(self.lib.maple_comm_unique_id(self.h, _ptr(buf)))
        return buf

    def comm_init(self, world, rank, unique_id):
        self._ck(self.lib.maple_comm_init(self.h, int(world), int(rank), _ptr(_u8(unique_id))))

    def argmax_allreduce_dev(self, n, score_ptr, idx_ptr, stream=0):
        self._ck(self.lib.maple_argmax_allreduce_dev(self.h, int(n), C.c_void_p(score_ptr), C.c_void_p(idx_ptr), C.c_void_p(stream)))

    def timing_reset(self):
        self._ck(self.lib.maple_timing_reset(self.h))

    def timing_read_each(self, cap=256):
        ms = np.zeros(cap, dtype=np.float32)
        n = C.c_int32()
        self._ck(self.lib.maple_timing_read_each(self.h, int(cap), _ptr(ms), C.byref(n)))
        return ms[: n.value].tolist()

    def frontier_levels(self, cap=4096):
        """Per level of the last frontier-tier pass: (updating items, cached items, ms of k_fr_updating, ms of k_fr_cached)."""
        iu, ic = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
        ws, wb = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
        mu, mc = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        n = C.c_int32()
        self._ck(self.lib.maple_debug_frontier_levels(self.h, int(cap), _ptr(iu), _ptr(ic), _ptr(mu), _ptr(mc), C.byref(n), _ptr(ws), _ptr(wb)))
        k = min(cap, n.value)
        self.last_wave_items = (ws[:k], wb[:k])      # (of the updating items: walked a wavefront each, small / 512-entry class)
        return iu[:k], ic[:k], mu[:k], mc[:k]

    def tree_rebuild_lists(self, root, up, c0, c1, tip, mut, dist, lower, bump_len=0.0):
        """reCalculateAllGenomeLists (M:6013-6347) with the level loop inside the library (maple_tree_rebuild_lists): returns
        (lower, upRight, upLeft, totUp, bad) -- new id columns, ``dist`` (float64, contiguous) updated in place when bump_len > 0,
        ``bad`` = nodes whose branches are still inconsistent (empty when the build went through)."""
        n = len(up)
        up, c0, c1 = _i32(up), _i32(c0), _i32(c1)
        tip = np.ascontiguousarray(np.asarray(tip, np.uint8))
        mutp = None if mut is None else _i32(mut)
        assert dist.dtype == np.float64 and dist.flags.c_contiguous
        lower = _i32(lower).copy()
        ur, ul, tu = (np.full(n, -1, np.int32) for _ in range(3))
        bad = np.zeros(max(64, n), np.int32)                  # (every node can be named: nothing is dropped)
        nbad = C.c_int32()
        self._ck(self.lib.maple_tree_rebuild_lists(self.h, n, int(root), _ptr(up), _ptr(c0), _ptr(c1), _ptr(tip), _ptr(mutp), _ptr(dist),
                                                   _ptr(lower), _ptr(ur), _ptr(ul), _ptr(tu), C.c_double(bump_len), len(bad), _ptr(bad),
                                                   C.byref(nbad)))
        return lower, ur, ul, tu, np.unique(bad[: min(nbad.value, len(bad))])

    def timing_read(self):
        """(number of timed *_dev launches since the last reset, their summed HIP-event time in ms)."""
        n, ms = C.c_int32(), C.c_double()
        self._ck(self.lib.maple_timing_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    KIND_SPR_SCORE, KIND_SPR_SEARCH, KIND_SPR_REPLAY, KIND_APPEND_QUERIES, KIND_APPEND_PAIRS, KIND_PLACE_SCORE = 1, 2, 3, 4, 5, 6
    KIND_FR_UPDATING, KIND_FR_CACHED, KIND_FR_REPLAY, KIND_FR_WIDE = 7, 8, 9, 10

    def timing_read_kind(self, kind):
        """(launches, summed HIP-event ms, units of work, algorithmic bytes) of one kind of timed launch since the reset."""
        n, ms, u, b = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
        self._ck(self.lib.maple_timing_read_kind(self.h, int(kind), C.byref(n), C.byref(ms), C.byref(u), C.byref(b)))
        return n.value, ms.value, u.value, b.value

    def append_algorithmic_bytes(self, parent, child=None, child_once=False):
        parent = _i32(parent)
        ch = None if child is None else _i32(child)
        b = C.c_uint64()
        self._ck(self.lib.maple_append_algorithmic_bytes(self.h, len(parent), _ptr(parent), _ptr(ch),
                                                         int(bool(child_once)), C.byref(b)))
        return b.value
