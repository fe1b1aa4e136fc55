// maple_amd/csrc/ctx_host.h -- the library's context (host side) and the few helpers every translation unit of
// libmaple_hip.so shares.  Included by maple_hip.hip (C ABI, batch operators, dense / replay tiers of the SPR search) and
// frontier.hip (the frontier tier of the SPR search).
#pragma once
#include "../../include/maple_hip.h"
#ifdef MAPLE_DEBUG_ABI
#include "../../include/maple_hip_debug.h"
#endif
#include "genome_dev.h"
#include "search_dev.h"
#include "placement_dev.h"

#include <cstdarg>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

using namespace maple;

// =================================================================================================
// context
// =================================================================================================
typedef maple::ArenaViewS ArenaView;   // {words, aux, ent_off[], aux_off[], n_ent[], n_aux[]} per list
typedef maple::MutViewS MutView;       // {mut3, off[], cnt[]} per mutation list

struct DevBufStats { size_t allocs = 0, bytes = 0; };   // (for the verbose account of a call: what it had to allocate)
inline DevBufStats &devbuf_stats() { static DevBufStats s; return s; }
template <class T> struct DevBuf {     // grow-only device scratch
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        size_t want = n + n / 2 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        cap = (e == hipSuccess) ? want : 0;
        devbuf_stats().allocs++; devbuf_stats().bytes += want * sizeof(T);
        return e;
    }
    hipError_t reserve_exact(size_t n)     // for the very large buffers: no growth margin
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        cap = (e == hipSuccess) ? n : 0;
        devbuf_stats().allocs++; devbuf_stats().bytes += n * sizeof(T);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf {                          // grow-only page-locked host scratch (a D2H copy into pageable memory is staged
    void *p = nullptr;                  // by the runtime at a few GB/s: 0.3 ms for the 1.6 MB of one query's scores)
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        const size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        cap = (e == hipSuccess) ? want : 0;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct PlaceMeta {                     // derived from the uploaded tree, rebuilt when it or effectivelyNon0BLen changes
    bool valid = false;
    double effNon0 = -1.0;
    int32_t nF = 0, maxDepth = 0;
    std::vector<int32_t> frameOf;                  // per node
    std::vector<int32_t> frameNode, frameParent;   // per frame (frame 0 = the root's reference, node -1)
    std::vector<int32_t> levelStart;               // frames 1.. sorted by nesting depth; level l = [levelStart[l], levelStart[l+1])
    std::vector<int32_t> cand, leaves;             // node ids
    std::vector<int32_t> order;                    // nodes reachable from the root, depth-first
    int32_t rootVect = -1;                         // rootVector(probVect[root]) of the uploaded tree (list id), kept while it lives
    std::vector<int32_t> h_candIdx, h_leafIdx;     // per node: column in the score / minor matrix or -1
    struct PNode { int32_t candCol, leafCol, c0, c1; };
    std::vector<PNode> h_pn;                       // per node, for the host traversal of a single query: what a visit reads, in ONE 16-byte
                                                   // record (four arrays of 4-8 MB each were four cache misses per visit at 1 000 000 tips)
    std::vector<int32_t> h_candList, h_candFrame, h_leafList, h_leafFrame;   // host copies of the column arrays (maple_tree_patch)
    bool scanStale = false;                        // the tree changed through maple_tree_patch: h_scan / d_scan / order are old
    std::vector<ScanRec> h_scan;                   // the tree in traversal order (placement_dev.h)
    DevBuf<ScanRec> d_scan;
    DevBuf<int32_t> d_frameOf, d_candIdx, d_leafIdx, d_candList, d_candFrame, d_leafList, d_leafFrame;
    DevBuf<int32_t> d_pn;                          // h_pn on the device (4 words per node; kept current by maple_tree_patch): the expansion of
                                                   // maple_placement_ahead walks the tree there
};

// maple_placement_ahead: the score rows of the NEXT samples of a serial placement loop (M:11692-11752), made in one launch of
// the batch kernel and kept current under maple_tree_patch -- see placement_host.h
struct PlaceAhead {
    bool active = false;
    int32_t K = 0, next = 0;           // rows; the row the next single-query search takes
    std::vector<int32_t> q;            // the samples' list ids in the order they will be searched
    maple_placement_params pp{};
    // the score rows live in HBM ([K][ld] f64: kernels writing them straight into host memory moved 9 GB/s over PCIe, a
    // millisecond per 1 000 000-tip row); the row of the sample that is searched next is copied to the host by the copy engine
    // while the sample before it is being placed (hRow: two page-locked rows, taking turns), and what the placement in between
    // changed -- a handful of columns -- is patched into it from `hPatch`
    DevBuf<double> dTable;
    double *hRow[3] = {nullptr, nullptr, nullptr}; size_t capRow = 0;    // doubles per row buffer (this sample's, the next one's -- a traversal
                                       // made ahead may be reading it --, the one after that on its way)
    int32_t rowInBuf[3] = {-1, -1, -1};    // which row each buffer holds (or is receiving)
    int buf_of(int32_t row) const { for (int b = 0; b < 3; b++) if (rowInBuf[b] == row) return b; return -1; }
    int free_buf(int32_t keepA, int32_t keepB) const { for (int b = 0; b < 3; b++) if (rowInBuf[b] != keepA && rowInBuf[b] != keepB) return b; return -1; }
    double *hPatch = nullptr, *dPatch = nullptr; size_t capPatch = 0;   // page-locked (bytes), written by the scoring launch of a search: the changed columns' scores, then the changed leaves' flags
    void *hMinor = nullptr; uint8_t *dMinor = nullptr; size_t capMinor = 0;   // [K][ldL] u8 minor-sequence flags, page-locked, written by the kernel (1 MB per row)
    hipStream_t copyStream = nullptr;
    int64_t ld = 0, ldL = 0;           // row strides: the columns at scoring time + room for what the placements add; [ld - 1] = the root vector's score
    DevBuf<int32_t> dQ, dCols, dLists;
    std::vector<int32_t> dirtyCols, dirtyLeaves;     // columns whose list changed (or that are new) since the rows were made: every search scores them for its sample
    bool rootDirty = false;            // ... the root vector (its score sits at [ld - 1])
    long long refreshes = 0, refreshedPairs = 0;
    // the rows hold the scores of the branches an expansion under permissive rules reached (placement_host.h); every other column
    // holds PLACE_NO_SCORE, and a traversal that asks for one has the whole row scored first (counted)
    bool sparse = false;
    long long fallbacks = 0, expanded = 0, searches = 0;
    // The traversal of the NEXT announced sample, run by a host thread while the library's caller is busy with the sample just
    // searched (refinement, tree edit, updatePartials): speculative -- made on the tree as it is BEFORE that sample's placement --
    // and used only if the placement's patch touches no node the traversal visited (placement_host.h).
    struct Spec {
        std::thread th;
        int32_t row = -1, id = 0, status = -1, buf = 0;     // the row it is for (-1: none), its epoch in visitEpoch, 0 = outputs usable
        std::vector<int32_t> cols, lists, leafCols, leafLists;   // the changed columns as of its start
        std::vector<int32_t> hi; std::vector<double> hf; std::vector<uint8_t> hb;   // the traversal's outputs (PlaceOut over these)
        std::vector<int32_t> touched;                        // nodes patched since it set off
        bool rootTouched = false;
    } spec;
    std::vector<int32_t> visitEpoch;   // per node: the epoch of the last speculative traversal that visited it
    int32_t specSeq = 0;
    hipStream_t specStream = nullptr;
    DevBuf<int32_t> dSpecLists, dSpecLeaf;
    double *hSpecPatch = nullptr, *dSpecPatch = nullptr; size_t capSpecPatch = 0;   // bytes
    long long specUsed = 0, specDropped = 0;
    void join() { if (spec.th.joinable()) spec.th.join(); }
    DevBuf<uint8_t> dItems; DevBuf<unsigned long long> dCtr;
    void release()
    {
        join();
        if (hSpecPatch) (void)hipHostFree(hSpecPatch);
        hSpecPatch = nullptr; dSpecPatch = nullptr; capSpecPatch = 0;
        if (specStream) (void)hipStreamDestroy(specStream);
        specStream = nullptr;
        dSpecLists.release(); dSpecLeaf.release();
        for (double *&r : hRow) { if (r) (void)hipHostFree(r); r = nullptr; }
        if (hPatch) (void)hipHostFree(hPatch);
        if (hMinor) (void)hipHostFree(hMinor);
        if (copyStream) (void)hipStreamDestroy(copyStream);
        hPatch = nullptr; dPatch = nullptr; hMinor = nullptr; dMinor = nullptr; copyStream = nullptr; capRow = capPatch = capMinor = 0;
        dTable.release(); dQ.release(); dCols.release(); dLists.release(); dItems.release(); dCtr.release();
        active = false;
    }
};

struct maple_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // side stream: the dense scoring of the searches known to be whole-tree ones runs next to
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the lane searches of the others (maple_spr_search_batch)
    DevBuf<int32_t> z_ql;
    DevBuf<uint8_t> z_qt;
    DevBuf<double> z_qb;
    std::vector<hipEvent_t> evs;       // pairs (start, stop) of timed *_dev launches since the last reset
    size_t ev_used = 0;
    // what each timed launch was: kind (MAPLE_K_*), units of work (pairs scored / searches run) and the algorithmic
    // bytes of SURVEY 8d for the scoring kernels
    std::vector<int32_t> ev_kind;
    std::vector<double> ev_units, ev_bytes;
    std::vector<double> cand_bytes_prefix;       // per scored column of the uploaded tree: 8E + 8A + 8, summed (host)
    double scored_bytes_total = 0.0;
    std::string err;
    maple_tuning tuning{};             // maple_set_tuning
    maple_params params{};
    int32_t lRef = 0;
    std::vector<uint8_t> refIdx;
    DevModel dm{};                     // device pointers inside
    DevModel *d_model = nullptr;       // the same struct in device memory: what the kernels read
    bool model_set = false;
    double *d_siteRates = nullptr, *d_errorRates = nullptr, *d_cumRate = nullptr, *d_cumErr = nullptr;
    int32_t *d_cumBases = nullptr;
    double *d_rflec = nullptr;
    std::vector<double> h_cumRate, h_cumErr;
    // list arena
    uint2 *d_words = nullptr;
    double *d_aux = nullptr;
    int64_t cap_ent = 0, cap_aux = 0, cap_lists = 0;
    int64_t used_ent = 0, used_aux = 0;
    int64_t *d_ent_off = nullptr, *d_aux_off = nullptr;
    int32_t *d_n_ent = nullptr, *d_n_aux = nullptr;
    std::vector<int64_t> h_ent_off, h_aux_off;
    std::vector<int32_t> relocated;    // lists maple_lists_update moved to the end of the arena (maple_arena_release keeps their room)
    std::vector<int32_t> h_n_ent, h_n_aux;
    // mutation lists
    int32_t *d_mut3 = nullptr;
    int64_t *d_mut_off = nullptr;
    int32_t *d_mut_cnt = nullptr;
    int64_t cap_mut = 0, used_mut = 0, cap_mut_lists = 0;
    std::vector<int64_t> h_mut_off;
    std::vector<int32_t> h_mut_cnt;
    // staging / scratch
    DevBuf<int32_t> s_i32[8];
    DevBuf<double> s_f64[4];
    DevBuf<uint8_t> s_u8[4];
    DevBuf<int64_t> s_i64[6];
    DevBuf<uint2> s_words, s_pool_w;
    DevBuf<double> s_aux, s_pool_a;
    DevBuf<double> s_ais;
    // tree mirror (topology + list ids), maple_tree_upload
    DevTree dtree{};
    bool tree_set = false;
    DevBuf<int32_t> t_i32[9];
    DevBuf<double> t_dist;
    DevBuf<uint8_t> t_tip;
    DevBuf<uint8_t> t_nodes;           // NodeRec[n], 64-byte aligned
    std::vector<int32_t> h_tree_up, h_tree_lower;
    std::vector<double> h_tree_dist;
    std::vector<uint8_t> h_tree_tip;
    bool tree_has_mut = false;
    int32_t tree_max_ent = 0;          // longest genome list of the uploaded tree (entries)
    int32_t n_scored = 0;              // nodes with a probVectTotUp in the searches' depth-first order (trees with local references: by frame,
                                       // then depth-first): t_i32[8] = list ids, t_scored_col = preRank, t_scored_frame = frame
    DevBuf<int32_t> t_scored_col, t_scored_frame;
    // the same candidates in the searches' own depth-first order whatever the tree (host): list id, preRank, reference frame
    std::vector<int32_t> h_cand_ids, h_cand_rank, h_cand_frame;
    int over_hint_budget = -1; double over_hint_eff0 = -1.0;   // the budget and effectivelyNon0BLen the hints were taken under: another pair drops them
    std::vector<uint8_t> h_over_hint;  // per node: its SPR search ran over the wide-search budget the last time (error model: routing hint, dropped with the tree)
    int64_t cand_root_mark = -1, cand_root_top = -1;   // where the copies begin (an arena mark) and the list count right after them: stale copies
                                        // that are still the arena's last lists are released before new ones are made
    int64_t cand_root_end = -1;        // >= 0: s_cand_root is current (the root-frame copies are arena lists below this id); dropped with the
                                       // tree, by a maple_arena_release below them, by maple_arena_compact
    DevBuf<int32_t> s_frame_parent, s_frame_node;   // the frames' nesting on the device (per call of the SPR search)
    DevBuf<int32_t> t_cand_rank, s_cand_root;   // device: the ranks; the candidates' lists re-expressed in the root's frame (per call)
    DevBuf<int4> t_frame_chunks;       // trees with local references: the scored candidates in chunks of <= 64 within one frame
    int32_t n_frame_chunks = 0;
    DevBuf<uint8_t> s_tilebest;        // (query, 64-candidate tile) records of maple_append_queries_argmax_dev
    void *rccl_lib = nullptr;          // RCCL, loaded on first use (maple_comm_*)
    void *rccl_comm = nullptr;
    int rccl_world = 1, rccl_rank = 0;
    DevBuf<unsigned long long> s_comm_u64;
    DevBuf<long long> s_fan[6];        // per (query, frame) item of a nesting level: capacities, offsets, sizes (k_fan_*)
    DevBuf<uint8_t> s_fan_tmp;
    DevBuf<SScan> t_scan;              // the tree in the searches' depth-first order (search_dev.h), per effectivelyNon0BLen
    DevBuf<int32_t> t_scan_parent, t_cand_before, t_clade_visits;
    DevBuf<unsigned long long> s_fin_mask;    // bitmaps of the finite scores of the whole-tree searches' rows (FiniteRows, search_dev.h)
    DevBuf<int32_t> s_fin_prefix;
    bool scan_valid = false;
    double scan_eff = -1.0;
    std::vector<int32_t> h_depth;      // per node: distance from the root in branches
    std::vector<int32_t> h_clade;      // per node: nodes in its clade (itself included); filled on first use, dropped with the tree
    int32_t tree_max_depth = 0;
    // SPR search workspace
    DevBuf<uint8_t> s_search_ws, s_search_ws_big;
    DevBuf<uint8_t> s_search_out;
    DevBuf<int32_t> s_counter;
    DevBuf<double> s_cache;            // cached (query x node) scores of wide searches
    struct CandSet { int32_t n = 0, nFrames = 0; int32_t *lists = nullptr, *frame = nullptr; };
    std::vector<CandSet> candsets;     // resident candidate sets (maple_candset_create)
    // batched placement (maple_placement_search_batch)
    PlaceMeta *place = nullptr;
    PlaceAhead *ahead = nullptr;
    std::vector<int32_t> h_tree_c0, h_tree_c1, h_tree_mut, h_tree_totUp, h_tree_upRight, h_tree_upLeft;
    std::vector<NodeRec> h_nodes;      // host copy of the node records (host-side traversal of tiny placement batches)
    DevBuf<int32_t> p_i32[4];
    DevBuf<double> p_f64[2], p_score;
    PinBuf pin_place;                   // single-query placement: scores and minor-sequence flags on their way to the host
    PinBuf pin_res;                     // small per-launch results (a copy into pageable memory costs an extra ~10 us)
    DevBuf<int16_t> p_i16;
    DevBuf<uint8_t> p_u8, p_minor;
    DevBuf<uint32_t> p_from;
    int32_t *d_tile_counters = nullptr;    // ring of tile counters for the dynamically scheduled kernels
    int tile_counter_next = 0;
    void *upd = nullptr;               // UpdateScratch of maple_update_partials (update_host.h)
    std::vector<SearchOut> h_search_out;   // per-search results of the last maple_spr_search_batch on the host (kept: 24 MB of fresh
                                       // pages per call cost 5 ms at 200 000 searches)
    void *frontier = nullptr;          // FrontierScratch of the frontier tier of the SPR search (frontier.hip)
    void *witness = nullptr;           // WitnessScratch of the whole-tree searches' candidate filter (witness.hip)
    bool last_search_frontier_only = false;   // maple_spr_search_visited can report on the last maple_spr_search_batch
    bool nodes_current = false;        // the node records of the SPR search on the device follow every maple_tree_patch
    bool tree_stale = false;           // maple_tree_patch changed the host copy of the tree; the device tables of the SPR search
                                       // (and, for batches, of the placement search) are rebuilt from it before their next use
    // Staging of the small per-call argument columns of the batch operators: they are gathered in pinned host memory and go
    // to the device in ONE copy per call (a dozen separate copies from pageable memory cost ~0.2 ms per call, most of a
    // single-change updatePartials).  Two arenas used in turn: see stage_begin.
    uint8_t *stg_h[2] = {nullptr, nullptr}, *stg_d[2] = {nullptr, nullptr};
    size_t stg_cap[2] = {0, 0}, stg_used = 0, stg_flushed = 0;
    int stg_cur = 0;
    bool commit_pending = false;       // commit_lists left its copy kernel running on `stream`: entry points that launch on a
                                       // caller's stream wait for it first (settle)
    bool tolerate_fatal = false;       // maple_set_fatal_policy
    int trace_query = -1;
    DevBuf<int32_t> s_trace_i;
    DevBuf<double> s_trace_d;
};

enum { MAPLE_K_OTHER = 0, MAPLE_K_SPR_SCORE = 1, MAPLE_K_SPR_SEARCH = 2, MAPLE_K_SPR_REPLAY = 3, MAPLE_K_APPEND_QUERIES = 4,
       MAPLE_K_APPEND_PAIRS = 5, MAPLE_K_PLACE_SCORE = 6, MAPLE_K_FR_UPDATING = 7, MAPLE_K_FR_CACHED = 8, MAPLE_K_FR_REPLAY = 9,
       MAPLE_K_FR_WIDE = 10 };

static inline int fail(maple_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCK(c, call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail((c), MAPLE_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

static inline ArenaView view(const maple_ctx *c) { return ArenaView{c->d_words, c->d_aux, c->d_ent_off, c->d_aux_off, c->d_n_ent, c->d_n_aux}; }
static inline MutView mview(const maple_ctx *c) { return MutView{c->d_mut3, c->d_mut_off, c->d_mut_cnt}; }

__device__ inline ListRef list_ref(const ArenaView &a, int id)
{
    return ListRef{a.words + a.ent_off[id], a.aux + a.aux_off[id]};
}


// batches of at most this many items run one wavefront per item (maple_tuning.wavePerItemMax over the operator's own default)
static inline int wave_item_max(const maple_ctx *c, int dflt)
{
    return c->tuning.wavePerItemMax < 0 ? 0 : (c->tuning.wavePerItemMax > 0 ? c->tuning.wavePerItemMax : dflt);
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// one (start, stop) pair of HIP events from the context's pool, tagged with the kind of launch it brackets and the work it
// did (maple_timing_read_kind); defined in maple_hip.hip
__attribute__((visibility("hidden")))
int maple_internal_ev_pair(maple_ctx *c, hipEvent_t *a, hipEvent_t *b, int kind, double units, double bytes);
