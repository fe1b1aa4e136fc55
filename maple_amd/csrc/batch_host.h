// maple_amd/csrc/batch_host.h -- the small host-side helpers every translation unit with batch entry points shares (each gets its
// own copy: they only touch the context): launch geometry, the dispatch over the three model switches, list-id checks, the
// staging arena for a call's arguments, event pairs; and the declarations of the few larger ones that live in maple_hip.hip
// (hidden visibility: not part of the C ABI).  Included by maple_hip.hip and spr_batch.hip.
#pragma once
#include "ctx_host.h"

#include <algorithm>
#include <cstring>
#include <vector>

// the best (score, earliest visit rank, column) of one query over one tile of candidates: k_append_queries* / k_argmax_reduce
struct alignas(16) TileBest { double score; int32_t rank, idx; };

#ifndef MAPLE_BLOCK
#define MAPLE_BLOCK 256
#endif
#ifndef MAPLE_ZERO_DIST_BUDGET
#define MAPLE_ZERO_DIST_BUDGET 16      // traversal placements a search from a zero-length branch gets before the dense tier
#endif

static int grid_for(int n)
{
    int g = (n + MAPLE_BLOCK - 1) / MAPLE_BLOCK;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;      // 256 CUs x 8 workgroups, grid-stride beyond
    return g;
}

#define DISPATCH3(c, KERNEL, ...)                                                                          \
    do {                                                                                                  \
        const bool rv_ = (c)->dm.useRateVariation, u_ = (c)->dm.usingErrorRate, ss_ = (c)->dm.errorRateSiteSpecific; \
        if (!rv_ && !u_) KERNEL<false, false, false> __VA_ARGS__;                                          \
        else if (rv_ && !u_) KERNEL<true, false, false> __VA_ARGS__;                                       \
        else if (!rv_ && u_ && !ss_) KERNEL<false, true, false> __VA_ARGS__;                               \
        else if (!rv_ && u_ && ss_) KERNEL<false, true, true> __VA_ARGS__;                                 \
        else if (rv_ && u_ && !ss_) KERNEL<true, true, false> __VA_ARGS__;                                 \
        else KERNEL<true, true, true> __VA_ARGS__;                                                         \
    } while (0)


static int check_ids(maple_ctx *c, int32_t n, const int32_t *ids, bool allowNeg, const char *what)
{
    const int32_t nl = (int32_t)c->h_n_ent.size();
    for (int i = 0; i < n; i++)
        if (ids[i] >= nl || (ids[i] < 0 && !allowNeg))
            return fail(c, MAPLE_ERR_ARG, "%s[%d] = %d is not a list id (have %d)", what, i, ids[i], nl);
    return MAPLE_OK;
}

static inline int ev_pair(maple_ctx *c, hipEvent_t *a, hipEvent_t *b, int kind = 0, double units = 0.0, double bytes = 0.0)
{
    return maple_internal_ev_pair(c, a, b, kind, units, bytes);
}

template <class T> static int h2d(maple_ctx *c, DevBuf<T> &b, const T *src, size_t n)
{
    HIPCK(c, b.reserve(n ? n : 1));
    if (n) HIPCK(c, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return MAPLE_OK;
}

// Start staging the arguments of one batch call (at most `bytes` of them).  The two arenas alternate from call to call:
// every batch operator synchronises its stream at least once after its first stage_flush, so by the time an arena comes
// round again every copy out of it -- including a trailing asynchronous one -- has completed.
static int stage_begin(maple_ctx *c, size_t bytes)
{
    c->stg_cur ^= 1;
    const int k = c->stg_cur;
    bytes += 4096;
    if (bytes > c->stg_cap[k]) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (c->stg_h[k]) (void)hipHostFree(c->stg_h[k]);
        if (c->stg_d[k]) (void)hipFree(c->stg_d[k]);
        c->stg_h[k] = nullptr; c->stg_d[k] = nullptr; c->stg_cap[k] = 0;
        const size_t want = std::max(bytes * 2, (size_t)1 << 20);
        HIPCK(c, hipHostMalloc((void **)&c->stg_h[k], want, hipHostMallocDefault));
        HIPCK(c, hipMalloc((void **)&c->stg_d[k], want));
        c->stg_cap[k] = want;
    }
    c->stg_used = c->stg_flushed = 0;
    return MAPLE_OK;
}
// n items of `src` into the staging arena; returns where they will be on the device after stage_flush (null if out of room:
// stage_begin was given too small a bound)
template <class T> static T *stage_put(maple_ctx *c, const T *src, size_t n)
{
    const int k = c->stg_cur;
    const size_t off = (c->stg_used + 15) & ~(size_t)15, bytes = n * sizeof(T);
    if (off + bytes > c->stg_cap[k]) return nullptr;
    if (bytes) memcpy(c->stg_h[k] + off, src, bytes);
    c->stg_used = off + bytes;
    return (T *)(c->stg_d[k] + off);
}
static int stage_flush(maple_ctx *c)
{
    const int k = c->stg_cur;
    if (c->stg_used > c->stg_flushed)
        HIPCK(c, hipMemcpyAsync(c->stg_d[k] + c->stg_flushed, c->stg_h[k] + c->stg_flushed, c->stg_used - c->stg_flushed,
                                hipMemcpyHostToDevice, c->stream));
    c->stg_flushed = c->stg_used;
    return MAPLE_OK;
}
#define STAGE(var, c, src, n) auto *var = stage_put((c), (src), (size_t)(n)); if (!var) return fail((c), MAPLE_ERR_NOMEM, "argument staging overflow")

static int need_model(maple_ctx *c)
{
    if (!c->model_set) return fail(c, MAPLE_ERR_STATE, "maple_set_model has not been called");
    return MAPLE_OK;
}


// ---- defined in maple_hip.hip ---------------------------------------------------------------------------------------------
// scratch lists into the arena, ids handed out (-1 for None): see the definition
__attribute__((visibility("hidden")))
int commit_lists(maple_ctx *c, int32_t n, const int64_t *d_woff, const int64_t *d_aoff, int32_t *d_n_ent, int32_t *d_n_aux,
                 int32_t *outList, const uint2 *srcW = nullptr, const double *srcA = nullptr);
// one launch of the queries x candidates scoring kernel on stream s (k_append_queries / k_append_queries_lds)
__attribute__((visibility("hidden")))
int launch_append_queries(maple_ctx *c, hipStream_t s, int nQ, const int32_t *qList, int nC, const int32_t *cand, int isTip, double bLen,
                          double *out, long long ldOut, const int32_t *outCol, const uint8_t *qTip, const double *qBLen, int kind,
                          double algBytes, TileBest *tileBest = nullptr, const int32_t *visitRank = nullptr, const int4 *chunkTab = nullptr,
                          int nChunkTab = 0, int nF = 1, unsigned long long *finMask = nullptr, bool lanesOnly = false);
// one launch of the placement phase's scoring kernel (k_place_score, placement_host.h) on the context's stream
__attribute__((visibility("hidden")))
int launch_place_score(maple_ctx *c, int nQ, int nF, const int32_t *qFrameLists, int nC, const int32_t *cand, const int32_t *candFrame,
                       int isTip, double bLen, double *out, long long ldOut, const int32_t *outCol, const uint8_t *qTip, const double *qBLen,
                       int kind = MAPLE_K_PLACE_SCORE, double algBytes = 0.0);
// ---- defined in spr_batch.hip ----------------------------------------------------------------------------------------------
// the device tree once more from the host copy of its columns (after maple_tree_patch left tables stale)
__attribute__((visibility("hidden")))
int tree_rebuild_from_host(maple_ctx *c);
