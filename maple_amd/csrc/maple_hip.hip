// maple_amd/csrc/maple_hip.hip -- libmaple_hip.so: kernels + C ABI (include/maple_hip.h).
// gfx950 only.  One lane walks one (parent list, child list) pair; see genome_dev.h.
#include "../../include/maple_hip.h"
#include "genome_dev.h"
#include "search_dev.h"
#include "placement_dev.h"
#include "append_lds.h"
#include "wave_dev.h"
#include "wave_update.h"

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace maple;

// RCCL is used through dlopen only (maple_comm_*, below): types from the header, no link-time dependency
#include <dlfcn.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi g_rccl{};

#include "ctx_host.h"
#include "frontier.h"
#include "witness.h"

// =================================================================================================
// kernels
// =================================================================================================
#define MAPLE_BLOCK 256

// appendProbNode over arbitrary pairs ----------------------------------------------------------
// 120 VGPRs / no scratch at 4 waves per SIMD measured fastest (5 waves spills, 3 waves loses latency hiding).
#ifndef MAPLE_APPEND_WAVES
#define MAPLE_APPEND_WAVES 4
#endif
#define MAPLE_APPEND_ATTR __launch_bounds__(MAPLE_BLOCK) __attribute__((amdgpu_waves_per_eu(MAPLE_APPEND_WAVES, MAPLE_APPEND_WAVES)))
#define MAPLE_QLDS 192                 // query-list words staged in LDS per wavefront (longer lists are read from HBM/L2)
template <bool RV, bool U, bool SS>
__global__ MAPLE_APPEND_ATTR void k_append(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *pl,
                                           const int32_t *cl, const uint8_t *tip, const double *bl, double *out)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = append_walk(c, list_ref(av, pl[i]), list_ref(av, cl[i]), tip[i] != 0, bl[i]);
}

// Q queries x C candidates, query-major output out[q*C + k]: pair (q, k) is handled by one lane.  A tile is one query x
// 64 consecutive candidates and every WAVEFRONT pulls its next tile from an atomic counter, so there is no barrier
// anywhere and a wavefront that drew short lists never idles behind its workgroup's longest lane.  Tiles are numbered
// candidate-chunk-major: the ~4 000 wavefronts in flight sweep the same few candidate chunks (hot in L1/L2) with
// different queries.  Callers pass the candidates SORTED BY LIST LENGTH so that the 64 lanes of a wavefront finish
// together.  Measured on the 10 000-sample bench tree (256 queries x 14 878 branches), ms per launch:
//   static 256-candidate tiles, query-major 2.56 | chunk-major 2.29 | dynamic 64-candidate tiles, query-major 2.30 |
//   dynamic + chunk-major 1.85 | + candidates sorted by length 1.53.
// (Staging the query in LDS behind __syncthreads() was 1.4x slower; several queries per tile lost balance: 2.06 at 4.)
struct alignas(16) TileBest { double score; int32_t rank, idx; };

template <bool RV, bool U, bool SS>
__global__ MAPLE_APPEND_ATTR void k_append_queries(const DevModel *__restrict__ mp, ArenaView av, int nQ,
                                                   const int32_t *qList, int nC, const int32_t *cand, int isTip,
                                                   double bLen, double *out, long long ldOut, const int32_t *outCol,
                                                   const uint8_t *qTip, const double *qBLen, int *counter,
                                                   TileBest *tileBest, const int32_t *visitRank, unsigned long long *finMask)
{
    __shared__ Lds lds;
    __shared__ unsigned long long qlds[MAPLE_BLOCK / 64][MAPLE_QLDS];   // the tile's query list, one copy per wavefront
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x & 63;
    unsigned long long *myq = qlds[threadIdx.x >> 6];
    const int nChunks = (nC + 63) / 64;
    const long long tiles = (long long)nQ * nChunks;
    double tbScore = -INFINITY;
    int tbRank = 0x7fffffff, tbIdx = -1;
    for (;;) {
        int j = 0;
        if (lane == 0) j = atomicAdd(counter, 1);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= tiles) break;
        const int ch = j / nQ;
        const int q = j - ch * nQ;
        const int k = ch * 64 + lane;
        const int ql = qList[q];
        const int nq = av.n_ent[ql];
        const ListRef qref = list_ref(av, ql);
        const bool staged = nq <= MAPLE_QLDS;                             // wave-uniform
        if (staged) {
            // all 64 lanes walk the same query: its words go to LDS once per tile (no workgroup barrier: the LDS
            // pipeline serves one wavefront's requests in order) and every step's query load is a ds_read
            for (int i = lane; i < nq; i += 64) myq[i] = ((const unsigned long long *)qref.w)[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const int cl = k < nC ? cand[k] : -1;                             // -1: this column has no list (score unused)
        bool finite = false;
        if (cl >= 0) {
            const bool tipq = qTip ? qTip[q] != 0 : isTip != 0;
            const double blq = qBLen ? qBLen[q] : bLen;
            double lk;
            if (staged) {
                PairWalk<RV, U, SS> w(c, qref, tipq, blq, myq);
                w.start(list_ref(av, cl));
                while (!w.step()) {}
                lk = w.finish();
            } else lk = append_walk(c, list_ref(av, cl), qref, tipq, blq);
            if (!tileBest) { if (!finMask || lk > -INFINITY) out[(long long)q * ldOut + (outCol ? outCol[k] : k)] = lk; }   // (see the LDS kernel)
            else { tbScore = lk; tbRank = visitRank ? visitRank[k] : k; tbIdx = k; }
            finite = lk > -INFINITY;
        }
        if (finMask) {
            const unsigned long long fm = __ballot(finite);
            if (lane == 0) finMask[(long long)q * nChunks + ch] = fm;
        }
        if (tileBest) {
            // the wavefront reduction of north_star: best score of the tile's 64 candidates, exact ties to the EARLIEST visit
            // (the reference keeps the first of equal scores: strict >, M:7083 / 8065); one 16-byte record per (query, tile)
            // instead of 64 scores
            for (int m2 = 32; m2 >= 1; m2 >>= 1) {
                const double os = __shfl_xor(tbScore, m2, 64);
                const int orank = __shfl_xor(tbRank, m2, 64), oidx = __shfl_xor(tbIdx, m2, 64);
                if (os > tbScore || (os == tbScore && orank < tbRank)) { tbScore = os; tbRank = orank; tbIdx = oidx; }
            }
            if (lane == 0) tileBest[(long long)q * nChunks + ch] = TileBest{tbScore, tbRank, tbIdx};
            tbScore = -INFINITY; tbRank = 0x7fffffff; tbIdx = -1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// The same Q x C scoring with the tile's 64 candidate lists staged in LDS (append_lds.h): a workgroup of 16 wavefronts (one
// per CU) takes a unit = (chunk of 64 candidates, block of MAPLE_LDS_QB queries), copies the chunk's words and aux doubles into
// LDS with coalesced loads -- and, with per-site rates, the rate of every entry's last site next to it -- and its
// wavefronts then pull the block's queries from an LDS counter: one query x the 64 staged candidates per tile, candidate
// words / stored lengths / O vectors / site rates and the query's words and rates all read with ds_read.  Chunks too long for
// the LDS budget, and queries longer than the strip, are walked from global memory as before.
#ifndef MAPLE_ZERO_DIST_BUDGET
#define MAPLE_ZERO_DIST_BUDGET 16      // traversal placements a search from a zero-length branch gets before the dense tier
#endif
#define MAPLE_LDS_BLOCK 1024
#define MAPLE_LDS_CAPW 4096            // candidate words per staged chunk (32 KB, + 32 KB of site rates with rate variation)
#define MAPLE_LDS_CAPA 1536            // candidate aux doubles per staged chunk (12 KB)
#ifndef MAPLE_LDS_QB
#define MAPLE_LDS_QB 512               // queries per unit: 128 / 256 / 512 measured 549 / 544 / 538 ms per launch at 100k tips
#endif
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_LDS_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_append_queries_lds(const DevModel *__restrict__ mp, ArenaView av, int nQ, const int32_t *qList, int nC, const int32_t *cand,
                          int isTip, double bLen, double *out, long long ldOut, const int32_t *outCol, const uint8_t *qTip,
                          const double *qBLen, int *counter, TileBest *tileBest, const int32_t *visitRank,
                          const int4 *chunkTab, int nChunkTab, int nF, unsigned long long *finMask)
{
    // chunkTab (trees with MAT local references): the chunks are given as {first candidate, candidates (<= 64), reference
    // frame, -}, each within ONE frame, and query q's list is qList[q * nF + frame] -- the query expressed in that frame
    constexpr int NW = MAPLE_LDS_BLOCK / 64;
    __shared__ Lds lds;
    __shared__ int cwoff[65], caoff[65];
    __shared__ int sUnit, sNext, sStaged;
    extern __shared__ unsigned long long dynU64[];
    // dynamic LDS: candidate words | candidate aux | [candidate rates] | per-wavefront query words | [per-wavefront query rates]
    unsigned long long *cW = dynU64;
    double *cA = (double *)(cW + MAPLE_LDS_CAPW);
    double *cR = cA + MAPLE_LDS_CAPA;
    unsigned long long *qW = (unsigned long long *)(cR + (RV ? MAPLE_LDS_CAPW : 0));
    double *qR = (double *)(qW + NW * MAPLE_QLDS);
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nChunks = chunkTab ? nChunkTab : (nC + 63) / 64, nQB = (nQ + MAPLE_LDS_QB - 1) / MAPLE_LDS_QB;
    const long long units = (long long)nChunks * nQB;
    unsigned long long *myq = qW + wave * MAPLE_QLDS;
    double *myqR = qR + wave * MAPLE_QLDS;
    double tbScore = -INFINITY;
    int tbRank = 0x7fffffff, tbIdx = -1;
    for (;;) {
        if (tid == 0) sUnit = atomicAdd(counter, 1);
        __syncthreads();
        const int unit = sUnit;
        if (unit >= units) break;
        const int ch = unit / nQB, qb = unit - ch * nQB;
        int c0 = ch * 64, nCk = min(64, nC - ch * 64), frame = 0;
        if (chunkTab) { const int4 u = chunkTab[ch]; c0 = u.x; nCk = u.y; frame = u.z; }
        // this lane's candidate and where its list sits in the staged chunk
        const int k = c0 + lane;
        const int cl = lane < nCk ? cand[k] : -1;
        if (wave == 0) {
            int ne = cl >= 0 ? av.n_ent[cl] : 0, na = cl >= 0 ? av.n_aux[cl] : 0;
            int pw = ne, pa = na;                                          // inclusive prefix sums over the 64 lists
            for (int d = 1; d < 64; d <<= 1) {
                const int ow = __shfl_up(pw, d, 64), oa = __shfl_up(pa, d, 64);
                if (lane >= d) { pw += ow; pa += oa; }
            }
            cwoff[lane] = pw - ne; caoff[lane] = pa - na;
            if (lane == 63) { cwoff[64] = pw; caoff[64] = pa; sStaged = (pw <= MAPLE_LDS_CAPW && pa <= MAPLE_LDS_CAPA) ? 1 : 0; sNext = 0; }
        }
        __syncthreads();
        const bool stagedC = sStaged != 0;
        if (stagedC) {                                                     // 4 lists per wavefront, coalesced within a list
            constexpr int perWave = (64 + NW - 1) / NW;
            for (int i = wave * perWave; i < min(64, wave * perWave + perWave); i++) {
                if (i >= nCk) break;
                const int li = cand[c0 + i];
                const unsigned long long *sw = (const unsigned long long *)(av.words + av.ent_off[li]);
                const double *sa = av.aux + av.aux_off[li];
                const int w0 = cwoff[i], nw = cwoff[i + 1] - w0, a0 = caoff[i], na2 = caoff[i + 1] - a0;
                for (int j = lane; j < nw; j += 64) {
                    const unsigned long long w = sw[j];
                    cW[w0 + j] = w;
                    if (RV) cR[w0 + j] = c.rate((int)(uint32_t)w - 1);
                }
                for (int j = lane; j < na2; j += 64) cA[a0 + j] = sa[j];
            }
        }
        __syncthreads();
        const int myW = cwoff[lane], myA = caoff[lane];
        for (;;) {
            int qi = 0;
            if (lane == 0) qi = atomicAdd(&sNext, 1);
            qi = __builtin_amdgcn_readfirstlane(qi);
            const int q = qb * MAPLE_LDS_QB + qi;
            if (qi >= MAPLE_LDS_QB || q >= nQ) break;
            const int ql = chunkTab ? qList[(long long)q * nF + frame] : qList[q];
            const int nq = av.n_ent[ql];
            const ListRef qref = list_ref(av, ql);
            const bool stagedQ = nq <= MAPLE_QLDS;                          // wave-uniform
            if (stagedQ) {
                for (int i = lane; i < nq; i += 64) {
                    const unsigned long long w = ((const unsigned long long *)qref.w)[i];
                    myq[i] = w;
                    if (RV) myqR[i] = c.rate((int)(uint32_t)w - 1);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            bool finite = false;
            if (cl >= 0) {
                const bool tipq = qTip ? qTip[q] != 0 : isTip != 0;
                const double blq = qBLen ? qBLen[q] : bLen;
                const MemLG qL{(lds_u64p)myq, qref.aux, (lds_f64p)myqR};
                const MemG qG{(const unsigned long long *)qref.w, qref.aux};
                double lk;
                if (stagedC) {
                    const MemL pL{(lds_u64p)(cW + myW), (lds_f64p)(cA + myA), (lds_f64p)(cR + myW)};
                    lk = stagedQ ? append_walk_m(c, pL, qL, tipq, blq) : append_walk_m(c, pL, qG, tipq, blq);
                } else {
                    const ListRef pr = list_ref(av, cl);
                    const MemG pG{(const unsigned long long *)pr.w, pr.aux};
                    lk = stagedQ ? append_walk_m(c, pG, qL, tipq, blq) : append_walk_m(c, pG, qG, tipq, blq);
                }
                // finMask: which of the tile's 64 scores are finite goes out as ONE word per (query, tile) and only the finite
                // scores are stored -- the searches these rows are for are the ones whose scores are nearly all -inf (a mismatch
                // over a zero-length branch), and an 8-byte store into every line of a row was most of the kernel's HBM traffic
                if (!tileBest) { if (!finMask || lk > -INFINITY) out[(long long)q * ldOut + (outCol ? outCol[k] : k)] = lk; }
                else { tbScore = lk; tbRank = visitRank ? visitRank[k] : k; tbIdx = k; }
                finite = lk > -INFINITY;
            }
            if (finMask) {
                const unsigned long long fm = __ballot(finite);
                if (lane == 0) finMask[(long long)q * nChunks + ch] = fm;
            }
            if (tileBest) {
                for (int m2 = 32; m2 >= 1; m2 >>= 1) {
                    const double os = __shfl_xor(tbScore, m2, 64);
                    const int orank = __shfl_xor(tbRank, m2, 64), oidx = __shfl_xor(tbIdx, m2, 64);
                    if (os > tbScore || (os == tbScore && orank < tbRank)) { tbScore = os; tbRank = orank; tbIdx = oidx; }
                }
                if (lane == 0) tileBest[(long long)q * nChunks + ch] = TileBest{tbScore, tbRank, tbIdx};
                tbScore = -INFINITY; tbRank = 0x7fffffff; tbIdx = -1;
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                                   // nobody may still read the chunk when it is restaged
    }
}
template <bool RV> static size_t lds_kernel_dyn_bytes()
{
    constexpr int NW = MAPLE_LDS_BLOCK / 64;
    return (size_t)MAPLE_LDS_CAPW * 8 + (size_t)MAPLE_LDS_CAPA * 8 + (RV ? (size_t)MAPLE_LDS_CAPW * 8 : 0)
           + (size_t)NW * MAPLE_QLDS * 8 * (RV ? 2 : 1);
}

// per query: the best of its tiles (same order: score, then earliest visit)
__global__ __launch_bounds__(64) void k_argmax_reduce(int nQ, int nChunks, const TileBest *tb, double *bestScore, int32_t *bestIdx)
{
    const int q = blockIdx.x;
    if (q >= nQ) return;
    TileBest b{-INFINITY, 0x7fffffff, -1};
    for (int i = threadIdx.x; i < nChunks; i += 64) {
        const TileBest t = tb[(long long)q * nChunks + i];
        if (t.score > b.score || (t.score == b.score && t.rank < b.rank)) b = t;
    }
    for (int m2 = 32; m2 >= 1; m2 >>= 1) {
        const double os = __shfl_xor(b.score, m2, 64);
        const int orank = __shfl_xor(b.rank, m2, 64), oidx = __shfl_xor(b.idx, m2, 64);
        if (os > b.score || (os == b.score && orank < b.rank)) { b.score = os; b.rank = orank; b.idx = oidx; }
    }
    if (threadIdx.x == 0) { bestScore[q] = b.score; bestIdx[q] = b.idx; }
}

// per-item scratch placement for list-producing kernels
struct OutSpec {
    uint2 *words;                      // scratch words
    double *aux;                       // scratch aux
    const int64_t *woff;               // per item offset into words
    const int64_t *aoff;               // per item offset into aux
    int32_t *n_ent;                    // per item result: entries (or <0 status)
    int32_t *n_aux;
};

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_merge(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l1,
                                                       const double *b1, const uint8_t *t1, const int32_t *l2,
                                                       const double *b2, const uint8_t *t2, const uint8_t *ud,
                                                       const int32_t *nm1, const int32_t *nm2, OutSpec o, double *outLK)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Writer w;
        w.init(o.words + o.woff[i], o.aux + o.aoff[i]);
        double lk = 0.0;
        int r = merge_walk(c, list_ref(av, l1[i]), b1[i], t1[i] != 0, list_ref(av, l2[i]), b2[i], t2[i] != 0,
                           ud[i] != 0, outLK != nullptr, nm1 ? nm1[i] : 0, nm2 ? nm2[i] : 0, w, &lk);
        o.n_ent[i] = r;
        o.n_aux[i] = w.na;
        if (outLK) outLK[i] = lk;
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_blen(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *pl,
                                                      const int32_t *cl, const uint8_t *tip, double *ais,
                                                      const int64_t *aisOff, double *t, uint8_t *isFalse)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        bool f;
        t[i] = blen_walk(c, list_ref(av, pl[i]), list_ref(av, cl[i]), tip[i] != 0, ais + aisOff[i], 1, &f);
        isFalse[i] = f ? 1 : 0;
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_differ(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l1,
                                                        const int32_t *l2, uint8_t *out)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (l2[i] < 0) ? 1 : (differ_walk(c, list_ref(av, l1[i]), list_ref(av, l2[i])) ? 1 : 0);
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_root_prob(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l,
                                                           double *out)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = root_prob_walk(c, list_ref(av, l[i]));
}

// One query against a resident candidate set: candidate k is scored against the query's list in ITS reference frame
// (frameLists[frameIdx[k]]), so trees with MAT local references need one launch per query, not one per frame.
template <bool RV, bool U, bool SS>
__global__ MAPLE_APPEND_ATTR void k_append_candset(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *parent,
                                                   const int32_t *frameIdx, const int32_t *frameLists, int isTip, double bLen,
                                                   double *out)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = append_walk(c, list_ref(av, parent[i]), list_ref(av, frameLists[frameIdx[i]]), isTip != 0, bLen);
}

__global__ __launch_bounds__(MAPLE_BLOCK) void k_minor_candset(int lRef, ArenaView av, int n, const int32_t *l1,
                                                               const int32_t *frameIdx, const int32_t *frameLists,
                                                               int onlyIdentical, uint8_t *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (uint8_t)minor_walk(lRef, list_ref(av, l1[i]), list_ref(av, frameLists[frameIdx[i]]), onlyIdentical != 0);
}

__global__ __launch_bounds__(MAPLE_BLOCK) void k_minor(int lRef, ArenaView av, int n, const int32_t *l1, const int32_t *l2,
                                                       int onlyIdentical, uint8_t *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (uint8_t)minor_walk(lRef, list_ref(av, l1[i]), list_ref(av, l2[i]), onlyIdentical != 0);
}

__global__ __launch_bounds__(MAPLE_BLOCK) void k_pass(int lRef, ArenaView av, MutView mv, int n, const int32_t *l,
                                                      const int32_t *ml, const uint8_t *up, OutSpec o)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Writer w;
        w.init(o.words + o.woff[i], o.aux + o.aoff[i]);
        int id = ml[i];
        int r = pass_walk(lRef, list_ref(av, l[i]), mv.mut3 + 3 * mv.off[id], mv.cnt[id], up[i] != 0, w);
        o.n_ent[i] = r;
        o.n_aux[i] = w.na;
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_shorten(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l, OutSpec o)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Writer w;
        w.init(o.words + o.woff[i], o.aux + o.aoff[i]);
        o.n_ent[i] = shorten_walk(c, list_ref(av, l[i]), av.n_ent[l[i]], w);
        o.n_aux[i] = w.na;
    }
}

// One item of a level of updatePartials (update_host.h): mergeVectors, then what the reference does with the result, in
// one go -- no trip to the host between the three.  mode 0 (a lower list, M:5760-5800): shorten(), then
// areVectorsDifferent(new, old); mode 1 (probVectTotUp, M:5525-5557): shorten(); mode 2 (probVectUpRight / UpLeft,
// M:5559-5660): areVectorsDifferent(old, new), and shorten() only if they differ.  The merged list goes to scratch slot A,
// the shortened one to slot B (what is committed).  n_ent: entries of B, -1 = None, < -1 = fatal; flag: "different".
#define MAPLE_UPDATE_ITEM_ARGS                                                                                                  \
    const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l1, const double *b1, const uint8_t *t1,               \
        const int32_t *l2, const double *b2, const uint8_t *t2, const uint8_t *ud, const uint8_t *mode, const int32_t *old,     \
        uint2 *words, double *aux, const int64_t *woff, const int64_t *cap, int32_t *res3

template <bool RV, bool U, bool SS>
__device__ inline void update_item_lane(const Ctx<RV, U, SS> &c, const ArenaView &av, int n, int i, const int32_t *l1, const double *b1,
                                        const uint8_t *t1, const int32_t *l2, const double *b2, const uint8_t *t2, const uint8_t *ud,
                                        const uint8_t *mode, const int32_t *old, uint2 *words, double *aux, const int64_t *woff,
                                        const int64_t *cap, int32_t *res3)
{
    Writer wa, wb;
    wa.init(words + woff[i], aux + 5 * woff[i]);
    wb.init(words + woff[i] + cap[i], aux + 5 * (woff[i] + cap[i]));
    double lk = 0.0;
    const int r = merge_walk(c, list_ref(av, l1[i]), b1[i], t1[i] != 0, list_ref(av, l2[i]), b2[i], t2[i] != 0, ud[i] != 0, false, 0,
                             0, wa, &lk);
    int ne = r, na = 0, flag = 1;
    if (r >= 0) {
        const ListRef A{wa.w, wa.aux};
        if (mode[i] == 2 && old[i] >= 0) flag = differ_walk(c, list_ref(av, old[i]), A) ? 1 : 0;
        if (flag) {
            ne = shorten_walk(c, A, r, wb);
            na = wb.na;
            if (mode[i] == 0 && old[i] >= 0) flag = differ_walk(c, ListRef{wb.w, wb.aux}, list_ref(av, old[i])) ? 1 : 0;
        } else ne = 0;
    }
    res3[i] = ne; res3[n + i] = na; res3[2 * n + i] = flag;
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_update_items(MAPLE_UPDATE_ITEM_ARGS)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        update_item_lane(c, av, n, i, l1, b1, t1, l2, b2, t2, ud, mode, old, words, aux, woff, cap, res3);
}

// The same item by a whole wavefront (wave_update.h): what a level with a handful of items -- a single change walking up
// and down the tree -- waits for is one item's latency, not throughput.  One wavefront per workgroup, one item at a time;
// lists too long for the staged walk go through the one-lane code on lane 0.
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_update_items_wave(MAPLE_UPDATE_ITEM_ARGS)
{
    __shared__ Lds lds;
    __shared__ WaveUpdLds L;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int id1 = l1[i], id2 = l2[i], idOld = old[i];
        const int n1 = av.n_ent[id1], n2 = av.n_ent[id2], nOld = idOld >= 0 ? av.n_ent[idOld] : 0;
        if (n1 > MAPLE_WU_IN || n2 > MAPLE_WU_IN || nOld > MAPLE_WU_CAP) {
            if (lane == 0) update_item_lane(c, av, n, i, l1, b1, t1, l2, b2, t2, ud, mode, old, words, aux, woff, cap, res3);
            continue;
        }
        wave_sync();                                                       // the item before is done with the LDS
        const ListRef Lo = idOld >= 0 ? list_ref(av, idOld) : ListRef{nullptr, nullptr};
        if (idOld >= 0) {
            const unsigned long long *wo = (const unsigned long long *)Lo.w;
            for (int k = lane; k < nOld; k += 64) L.old[k] = wo[k];
        }
        int naA = 0;
        const int r = wave_merge(c, list_ref(av, id1), n1, b1[i], t1[i] != 0, list_ref(av, id2), n2, b2[i], t2[i] != 0, ud[i] != 0, L, naA);
        int ne = r, na = 0, flag = 1;
        if (r >= 0) {
            const int md = mode[i];
            if (md == 2 && idOld >= 0) flag = wave_differ(c, L.old, Lo.aux, nOld, L.m, L.maux, r) ? 1 : 0;
            if (flag) {
                ne = wave_shorten(c, L, r, words + woff[i] + cap[i], aux + 5 * (woff[i] + cap[i]), na);
                if (md == 0 && idOld >= 0) flag = wave_differ(c, L.in, L.baux, ne, L.old, Lo.aux, nOld) ? 1 : 0;
            } else ne = 0;
        }
        if (lane == 0) { res3[i] = ne; res3[n + i] = na; res3[2 * n + i] = flag; }
    }
}

// The explicit-pair operators by one wavefront per pair, for the few pairs of a single reference call (mergeVectors without
// the likelihood, estimateBranchLengthWithDerivative; appendProbNode: k_wave_append below): same results, a fifth of the wait.
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_merge_wave(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l1,
                                                   const double *b1, const uint8_t *t1, const int32_t *l2, const double *b2,
                                                   const uint8_t *t2, const uint8_t *ud, OutSpec o)
{
    __shared__ Lds lds;
    __shared__ WaveUpdLds L;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int id1 = l1[i], id2 = l2[i];
        const int n1 = av.n_ent[id1], n2 = av.n_ent[id2];
        uint2 *gw = o.words + o.woff[i];
        double *ga = o.aux + o.aoff[i];
        if (n1 > MAPLE_WU_IN || n2 > MAPLE_WU_IN) {
            if (lane == 0) {
                Writer w;
                w.init(gw, ga);
                double lk = 0.0;
                o.n_ent[i] = merge_walk(c, list_ref(av, id1), b1[i], t1[i] != 0, list_ref(av, id2), b2[i], t2[i] != 0, ud[i] != 0, false, 0, 0,
                                        w, &lk);
                o.n_aux[i] = w.na;
            }
            continue;
        }
        wave_sync();
        int na = 0;
        const int r = wave_merge(c, list_ref(av, id1), n1, b1[i], t1[i] != 0, list_ref(av, id2), n2, b2[i], t2[i] != 0, ud[i] != 0, L, na);
        if (r >= 0) {
            for (int k = lane; k < r; k += 64) { const unsigned long long w = L.m[k]; gw[k] = make_uint2((uint32_t)w, (uint32_t)(w >> 32)); }
            for (int k = lane; k < na; k += 64) ga[k] = L.maux[k];
        }
        if (lane == 0) { o.n_ent[i] = r; o.n_aux[i] = r >= 0 ? na : 0; }
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_blen_wave(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *pl,
                                                  const int32_t *cl, const uint8_t *tip, double *ais, const int64_t *aisOff,
                                                  double *t, uint8_t *isFalse)
{
    __shared__ Lds lds;
    __shared__ WaveLds W;
    __shared__ double terms[128];
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int idP = pl[i], idC = cl[i];
        const int nP = av.n_ent[idP], nC = av.n_ent[idC];
        bool f = false;
        double v = 0.0;
        if (nP > MAPLE_WAVE_CAPW || nC > MAPLE_WAVE_CAPW) {
            if (threadIdx.x == 0) v = blen_walk(c, list_ref(av, idP), list_ref(av, idC), tip[i] != 0, ais + aisOff[i], 1, &f);
        } else {
            wave_sync();
            v = wave_blen(c, list_ref(av, idP), nP, list_ref(av, idC), nC, tip[i] != 0, W, terms, &f);
        }
        if (threadIdx.x == 0) { t[i] = v; isFalse[i] = f ? 1 : 0; }
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_differ_wave(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l1, const int32_t *l2,
                                                    uint8_t *out)
{
    __shared__ Lds lds;
    __shared__ unsigned long long A[MAPLE_WU_CAP], B[MAPLE_WU_CAP];
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int id1 = l1[i], id2 = l2[i];
        if (id2 < 0) { if (lane == 0) out[i] = 1; continue; }
        const int n1 = av.n_ent[id1], n2 = av.n_ent[id2];
        const ListRef L1 = list_ref(av, id1), L2 = list_ref(av, id2);
        if (n1 > MAPLE_WU_CAP || n2 > MAPLE_WU_CAP) {
            if (lane == 0) out[i] = differ_walk(c, L1, L2) ? 1 : 0;
            continue;
        }
        wave_sync();
        const unsigned long long *w1 = (const unsigned long long *)L1.w, *w2 = (const unsigned long long *)L2.w;
        for (int k = lane; k < n1; k += 64) A[k] = w1[k];
        for (int k = lane; k < n2; k += 64) B[k] = w2[k];
        wave_sync();
        const bool d = wave_differ(c, A, L1.aux, n1, B, L2.aux, n2);
        if (lane == 0) out[i] = d ? 1 : 0;
    }
}

template <bool RV, bool U, bool SS>
__global__ void k_wave_append(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *pl, const int32_t *cl,
                              const uint8_t *tip, const double *bl, double *out);
#define MAPLE_WAVE_PAIRS_MAX 1024        // explicit-pair batches up to this size go one wavefront per pair

// shorten of ONE list per wavefront (wave_shorten, wave_update.h): for the handful of lists a single-query placement shortens
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_shorten_wave(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l, OutSpec o)
{
    __shared__ Lds lds;
    __shared__ WaveUpdLds L;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int id = l[i], ne = av.n_ent[id], na = av.n_aux[id];
        if (ne > MAPLE_WU_CAP || na > 5 * MAPLE_WU_CAP) {
            if (lane == 0) {
                Writer w;
                w.init(o.words + o.woff[i], o.aux + o.aoff[i]);
                o.n_ent[i] = shorten_walk(c, list_ref(av, id), ne, w);
                o.n_aux[i] = w.na;
            }
            continue;
        }
        wave_sync();
        const ListRef src = list_ref(av, id);
        const unsigned long long *sw = (const unsigned long long *)src.w;
        for (int k = lane; k < ne; k += 64) L.m[k] = sw[k];
        for (int k = lane; k < na; k += 64) L.maux[k] = src.aux[k];
        wave_sync();
        int naOut = 0;
        const int r = wave_shorten(c, L, ne, o.words + o.woff[i], o.aux + o.aoff[i], naOut);
        if (lane == 0) { o.n_ent[i] = r; o.n_aux[i] = naOut; }
    }
}

// rootVector: frames up (node..root), root_walk, frames down (root..node), shorten.
// Each item owns 3 scratch lists of `cap` entries: A, B (ping-pong) and the final output slot.
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_root_vector(const DevModel *__restrict__ mp, ArenaView av, MutView mv, int n,
                                                             const int32_t *l, const double *bl, const uint8_t *tip,
                                                             const int64_t *pathOff, const int32_t *pathMut,
                                                             const int64_t *capOff, OutSpec o)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int64_t cap = capOff[i + 1] - capOff[i];
        // o.woff[i] addresses 3*cap words; o.aoff[i] addresses 3*5*cap doubles
        uint2 *W[3] = {o.words + o.woff[i], o.words + o.woff[i] + cap, o.words + o.woff[i] + 2 * cap};
        double *A[3] = {o.aux + o.aoff[i], o.aux + o.aoff[i] + 5 * cap, o.aux + o.aoff[i] + 10 * cap};
        ListRef cur = list_ref(av, l[i]);
        int curN = av.n_ent[l[i]];
        int slot = 1;                  // next buffer to write (1 or 2); slot 0 is reserved for the result
        Writer w;
        for (int64_t k = pathOff[i]; k < pathOff[i + 1]; k++) {
            int id = pathMut[k];
            if (id < 0 || mv.cnt[id] == 0) continue;
            w.init(W[slot], A[slot]);
            curN = pass_walk(m.lRef, cur, mv.mut3 + 3 * mv.off[id], mv.cnt[id], true, w);
            cur = ListRef{W[slot], A[slot]};
            slot = 3 - slot;
        }
        w.init(W[slot], A[slot]);
        curN = root_walk(c, cur, bl[i], tip[i] != 0, w);
        cur = ListRef{W[slot], A[slot]};
        slot = 3 - slot;
        for (int64_t k = pathOff[i + 1] - 1; k >= pathOff[i]; k--) {
            int id = pathMut[k];
            if (id < 0 || mv.cnt[id] == 0) continue;
            w.init(W[slot], A[slot]);
            curN = pass_walk(m.lRef, cur, mv.mut3 + 3 * mv.off[id], mv.cnt[id], false, w);
            cur = ListRef{W[slot], A[slot]};
            slot = 3 - slot;
        }
        w.init(W[0], A[0]);
        o.n_ent[i] = shorten_walk(c, cur, curN, w);
        o.n_aux[i] = w.na;
    }
}

// evaluatePlacement (M:6790-6806): three branch-length solves around three merges, then one append.
// Each item owns 3 scratch lists (capacities capA/capB/capC packed back to back) and an `ais` strip.
#define MAPLE_EVALPLACE_ARGS                                                                                                    \
    const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *midTot, const int32_t *down, const int32_t *up,        \
        const double *dist, const int32_t *rem, const uint8_t *remTip, const uint8_t *fromTip1, uint2 *sw, double *sa,          \
        const int64_t *capOff, double *ais, const int64_t *aisOff, double *out4, int32_t *status, double *comp2

// comp2 (optional): what the caller compares the optimised placement with (M:8101-8187) -- appendProbNode of the node's
// lower list on its upper list at the branch's own length and at the sum of the two optimised halves.
template <bool RV, bool U, bool SS>
__device__ inline void evalplace_item_lane(const Ctx<RV, U, SS> &c, const ArenaView &av, int i, const int32_t *midTot, const int32_t *down,
                                           const int32_t *up, const double *dist, const int32_t *rem, const uint8_t *remTip,
                                           const uint8_t *fromTip1, uint2 *sw, double *sa, const int64_t *capOff, double *ais,
                                           const int64_t *aisOff, double *out4, int32_t *status, double *comp2)
{
    const DevModel &m = c.m;
    ListRef Lmid = list_ref(av, midTot[i]), Ldown = list_ref(av, down[i]), Lup = list_ref(av, up[i]), Lrem = list_ref(av, rem[i]);
    const int nDown = av.n_ent[down[i]], nUp = av.n_ent[up[i]], nRem = av.n_ent[rem[i]];
    const int64_t base = capOff[i];
    uint2 *wA = sw + base, *wB = wA + (nDown + nRem), *wC = wB + (nUp + nRem);
    double *aA = sa + 5 * base, *aB = aA + 5 * (int64_t)(nDown + nRem), *aC = aB + 5 * (int64_t)(nUp + nRem);
    double *myAis = ais + aisOff[i];
    const bool rt = remTip[i] != 0, ft = fromTip1[i] != 0;
    bool f;
    Writer w;
    status[i] = 0;
    double bestApp = blen_walk(c, Lmid, Lrem, rt, myAis, 1, &f);
    w.init(wA, aA);
    int r = merge_walk(c, Ldown, dist[i] / 2, ft, Lrem, bestApp, rt, false, false, 0, 0, w, nullptr);
    if (r < 0) { status[i] = -1; return; }
    ListRef midLower{wA, aA};
    double bestTop = blen_walk(c, Lup, midLower, false, myAis, 1, &f);
    w.init(wB, aB);
    r = merge_walk(c, Lup, bestTop, false, Lrem, bestApp, rt, true, false, 0, 0, w, nullptr);
    if (r == -1) {
        bestTop = m.defaultBLen * 0.1;
        w.init(wB, aB);
        r = merge_walk(c, Lup, bestTop, false, Lrem, bestApp, rt, true, false, 0, 0, w, nullptr);
    }
    if (r < 0) { status[i] = -1; return; }
    ListRef midTop{wB, aB};
    double bestBottom = blen_walk(c, midTop, Ldown, ft, myAis, 1, &f);
    w.init(wC, aC);
    r = merge_walk(c, Lup, bestTop, false, Ldown, bestBottom, ft, true, false, 0, 0, w, nullptr);
    if (r < 0) { status[i] = -1; return; }
    ListRef newMid{wC, aC};
    out4[i * 4 + 0] = append_walk(c, newMid, Lrem, rt, bestApp);
    out4[i * 4 + 1] = bestBottom;
    out4[i * 4 + 2] = bestTop;
    out4[i * 4 + 3] = bestApp;
    if (comp2) {
        comp2[2 * i] = append_walk(c, Lup, Ldown, ft, dist[i]);
        comp2[2 * i + 1] = append_walk(c, Lup, Ldown, ft, bestBottom + bestTop);
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(MAPLE_BLOCK) void k_evalplace(MAPLE_EVALPLACE_ARGS)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        evalplace_item_lane(c, av, i, midTot, down, up, dist, rem, remTip, fromTip1, sw, sa, capOff, ais, aisOff, out4, status, comp2);
}

// The same by a whole wavefront: the chain of seven list walks (three branch-length solves around three merges, one
// append) is what a single-query placement's refinement waits for -- 50 us per link for one lane, a few us for 64
// (wave_blen / wave_merge, wave_update.h; wave_append, wave_dev.h; every list in LDS).  One wavefront per workgroup.
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_evalplace_wave(MAPLE_EVALPLACE_ARGS)
{
    __shared__ Lds lds;
    __shared__ WaveUpdLds L;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    WaveLds &W = *reinterpret_cast<WaveLds *>(L.baux);                     // (baux and old are not used by the merges here)
    double *terms = reinterpret_cast<double *>(L.old);
    static_assert(sizeof(WaveLds) <= sizeof(L.baux) && 128 * sizeof(double) <= sizeof(L.old), "LDS aliases");
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int idM = midTot[i], idD = down[i], idU = up[i], idR = rem[i];
        const int nMid = av.n_ent[idM], nDown = av.n_ent[idD], nUp = av.n_ent[idU], nRem = av.n_ent[idR];
        if (nMid > MAPLE_WAVE_CAPW || nRem > MAPLE_WAVE_CAPW || nDown + nRem > MAPLE_WAVE_CAPW || nUp + nRem > MAPLE_WAVE_CAPW
            || nUp + nDown > MAPLE_WAVE_CAPW) {
            if (lane == 0)
                evalplace_item_lane(c, av, i, midTot, down, up, dist, rem, remTip, fromTip1, sw, sa, capOff, ais, aisOff, out4, status,
                                    comp2);
            continue;
        }
        wave_sync();
        const ListRef Lmid = list_ref(av, idM), Ldown = list_ref(av, idD), Lup = list_ref(av, idU), Lrem = list_ref(av, idR);
        const ListRef Lm{(const uint2 *)L.m, L.maux};                      // where the merges leave their result
        const bool rt = remTip[i] != 0, ft = fromTip1[i] != 0;
        const double d = dist[i];
        bool f;
        int na, st = 0;
        double bestApp = wave_blen(c, Lmid, nMid, Lrem, nRem, rt, W, terms, &f), bestTop = 0.0, bestBottom = 0.0, lk = 0.0;
        int r = wave_merge(c, Ldown, nDown, d / 2, ft, Lrem, nRem, bestApp, rt, false, L, na);
        if (r < 0) st = -1;
        if (!st) {
            bestTop = wave_blen(c, Lup, nUp, Lm, r, false, W, terms, &f);
            r = wave_merge(c, Lup, nUp, bestTop, false, Lrem, nRem, bestApp, rt, true, L, na);
            if (r == -1) {
                bestTop = m.defaultBLen * 0.1;
                r = wave_merge(c, Lup, nUp, bestTop, false, Lrem, nRem, bestApp, rt, true, L, na);
            }
            if (r < 0) st = -1;
        }
        if (!st) {
            bestBottom = wave_blen(c, Lm, r, Ldown, nDown, ft, W, terms, &f);
            r = wave_merge(c, Lup, nUp, bestTop, false, Ldown, nDown, bestBottom, ft, true, L, na);
            if (r < 0) st = -1;
        }
        if (!st) lk = wave_append(c, Lm, r, Lrem, nRem, rt, bestApp, W);
        if (lane == 0) {
            status[i] = st;
            if (!st) { out4[i * 4 + 0] = lk; out4[i * 4 + 1] = bestBottom; out4[i * 4 + 2] = bestTop; out4[i * 4 + 3] = bestApp; }
        }
        if (!st && comp2) {
            const double c0 = wave_append(c, Lup, nUp, Ldown, nDown, ft, d, W);
            const double c1 = wave_append(c, Lup, nUp, Ldown, nDown, ft, bestBottom + bestTop, W);
            if (lane == 0) { comp2[2 * i] = c0; comp2[2 * i + 1] = c1; }
        }
    }
}


// 248 VGPRs let only 2 wavefronts share a SIMD; the search is latency-bound, so capping it at 128 VGPRs (4 wavefronts,
// some state spilled to scratch) is faster: deep round 134 -> 103 ms (3 waves 115, 5 waves 144, 8 waves 186).
#ifndef MAPLE_SPR_ATTR
#define MAPLE_SPR_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
// SPR regraft search: one lane = one query (state machine in search_dev.h) ---------------------------
struct LaneBytes { size_t w, aux, h, st, best, ais, total; };
static LaneBytes lane_bytes(const WsLayout &L)
{
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    LaneBytes b;
    b.w = al((size_t)L.capW * sizeof(uint2));
    b.aux = al((size_t)L.capA * sizeof(double));
    b.h = al((size_t)L.capH * sizeof(TList));
    b.st = al((size_t)L.capS * sizeof(StackItem));
    b.best = al((size_t)L.capB * sizeof(BestRec));
    b.ais = al((size_t)L.capAis * sizeof(double));
    b.total = b.w + b.aux + b.h + b.st + b.best + b.ais;
    return b;
}

// ASSIST: with the wave-assisted lane searches compiled in.  A kernel of its own (k_spr_search_assisted): inlined next to the
// rest, that path costs every launch registers (with an error model the whole kernel spilled four times as many, and the
// replay launches, which never use it, slowed by a quarter).
template <bool RV, bool U, bool SS, bool ASSIST>
__device__ __forceinline__ void spr_search_impl(const DevModel *__restrict__ mp, ArenaView av, MutView mv, DevTree T, SearchParams P, int n,
                                                   const int32_t *nodes, WsLayout L, LaneBytes LB, uint8_t *wsBase,
                                                   int32_t *counter, SearchOut *out, uint2 *poolW, double *poolA,
                                                   unsigned long long *poolUsed, long long poolCapW, long long poolCapA,
                                                   int traceQuery, int32_t *trI, double *trD, int trCap, int32_t *trN,
                                                   int activeLanes, const double *cacheS, int budget, const int32_t *rTable, int nF,
                                                   const int32_t *cacheRow, int leanVisits, unsigned long long *ovfUsed, uint8_t *ovfBase,
                                                   long long ovfChunks, FiniteRows fin)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    // only the first `activeLanes` lanes of every wavefront search: with few queries it is better to spread them
    // over many wavefronts (a wavefront executes the union of its lanes' control paths) than to fill 64-wide waves
    // Cached (whole-tree) launches run ONE search per wavefront (activeLanes == 1): lane 0 is the state machine, and all 64
    // lanes join it whenever the search descends into a clade in the cached regime (wave_scan_clade, search_dev.h).
    const bool coop = T.scan != nullptr;
    // lane searches on a tree without local references: the lanes that do not search stay and help -- every cached-regime
    // placement score a searching lane needs is computed by all 64 (wave_append), one request after the other
    // (not compiled into the kernels of the error models, where it is not used: their site factors are large, and with the
    // wavefront-wide walk inlined next to them the whole kernel spilled four times as many registers)
    const bool assist = ASSIST && !coop && leanVisits != 0 && cacheS == nullptr;
    const int coopMax = leanVisits > 1 ? leanVisits - 1 : 0;                 // (leanVisits = 1 + the most requests served one by one)
    if (!coop && !assist && (int)threadIdx.x >= activeLanes) return;
    const bool searcher = (int)threadIdx.x < activeLanes;
    const size_t lane = (size_t)blockIdx.x * activeLanes + (searcher ? threadIdx.x : 0);
    LaneWs ws;
    // the lane's own workspace (a search that outgrows its list room carries on in chunks of a shared pool: LaneWs::reserve)
    auto setWs = [&](uint8_t *base, const LaneBytes &B, const WsLayout &Lx) {
        ws.w = (uint2 *)base;
        ws.aux = (double *)(base + B.w);
        ws.h = (TList *)(base + B.w + B.aux);
        ws.st = (StackItem *)(base + B.w + B.aux + B.h);
        ws.best = (BestRec *)(base + B.w + B.aux + B.h + B.st);
        ws.ais = (double *)(base + B.w + B.aux + B.h + B.st + B.best);
        ws.L = Lx;
    };
    setWs(wsBase + lane * LB.total, LB, L);
    if (ovfBase && ovfChunks > 0) {
        ws.ovfUsed = ovfUsed; ws.ovfChunks = ovfChunks;
        ws.ovfW = (uint2 *)ovfBase; ws.ovfA = (double *)(ovfBase + (size_t)ovfChunks * L.capW * sizeof(uint2));
    }
    Search<RV, U, SS, ASSIST> S(c, av, mv, T, P, ws);
    extern __shared__ double dynLds[];   // coop: per-depth (lastLK, failedPasses) slots of the clade scan; assisted lane searches:
    WaveLds &wl = *(WaveLds *)dynLds;    // the staging area of the wavefront-wide appendProbNode (wave_dev.h)
    double *slotLK = dynLds;
    int *slotFails = (int *)(dynLds + T.scanDepthCap);
    unsigned *slotOwner = (unsigned *)(slotFails + T.scanDepthCap);
    bool active = false, done = !searcher;
    int q = -1, node = -1, curRow = 0;
    double curLK = 0.0;
    for (;;) {
        S.wantScan = false;
        if (!done) do {                 // (`continue` below ends this pass of the state machine)
        if (!active) {
            q = atomicAdd(counter, 1);
            if (q >= n) { done = true; break; }
            node = nodes[q];
            setWs(wsBase + lane * LB.total, LB, L);                           // (the last search may have moved into the shared pool)
            ws.usedW = ws.usedA = ws.nH = ws.sp = ws.nB = 0;
            ws.overflow = 0;
            S.nAppend = 0;
            SearchOut &o = out[q];
            o.bestNode = -1; o.placement = -1; o.status = 0; o.nAppend = 0;
            o.bestScore = 0.0; o.improvement = 0.0; o.currentLK = 0.0;
            o.blen[0] = o.blen[1] = o.blen[2] = 0.0;
            o.rprWoff = o.rprAoff = -1; o.rprN = o.rprNA = 0;
            o.nShortList = o.nSteps = 0; o.tStep = o.tReplay = o.tRefine = 0;
            const int parent = T.nd[node].up;
            if (parent < 0) { o.status = 1; continue; }               // the root cannot be re-placed (M:9626)
            // current placement cost, M:9629-9646
            const int childIdx = (T.nd[parent].c0 == node) ? 0 : 1;
            int vectUp = S.opPass(S.treeList(childIdx == 0 ? T.nd[parent].upRight : T.nd[parent].upLeft), T.nd[node].mutId, false);
            if (!S.valid(vectUp)) { o.status = ws.overflow ? -3 : -1; continue; }
            curLK = append_walk(c, S.ref(vectUp), S.ref(S.treeList(T.nd[node].lower)), T.nd[node].isTip != 0, T.nd[node].dist);
            o.currentLK = curLK;
            if (!(curLK < P.thrPlacement || T.nd[node].dist != 0.0)) { o.status = 2; continue; }   // M:9674
            ws.usedW = ws.usedA = ws.nH = 0;
            const size_t row = cacheRow ? (size_t)cacheRow[q] : (size_t)q;
            curRow = (int)row;
            S.cached = cacheS ? cacheS + row * T.n : nullptr;             // this query's row of the (queries x nodes) score table
            S.finMask = (cacheS && fin.mask) ? fin.mask + row * fin.nWords : nullptr;
            S.rTable = (cacheS && rTable) ? rTable + row * nF : nullptr;  // and of the (queries x frames) removed lists
            S.fShort[0] = S.fShort[1] = S.fShort[2] = S.fShort[3] = -1;
            // A pruned node on a zero-length branch is searched with removedBLen = 0 (M:9644).  Without an error model a
            // mismatch over zero length is impossible (-inf, M:6663), -inf never counts as a failed pass, and the search
            // walks the whole tree by the reference's own rules (tools/wide_stats.py, 100 000-tip bench tree: all 37 536
            // whole-tree searches of a deep round sit on zero-length branches and none of the other 136 494 does): it goes
            // to the dense tier after a token budget instead of spending the full one here first.  With an error model the
            // mismatch has a finite cost and such searches end like any other (1 000 000-tip run), so no hint then.  (A
            // routing hint only: the dense tier runs the same search from the start.)
            S.budget = (!U && budget > MAPLE_ZERO_DIST_BUDGET && T.nd[node].dist == 0.0) ? MAPLE_ZERO_DIST_BUDGET : budget;
            S.overBudget = false;
            S.lean = assist && q != traceQuery;                            // (a traced query keeps to step(), which records its visits)
            S.wantApp = S.haveMail = S.stepOnce = false;
            S.trI = nullptr;
            if (q == traceQuery && trI) { S.trI = trI; S.trD = trD; S.trCap = trCap; S.trN = 0; }
            S.begin(parent, childIdx, curLK, T.nd[node].dist);
            active = true;
        } else if (ws.overflow) {
            // workspace exhausted: in the budgeted pass the search is simply handed to the batch path like a wide one (it
            // restarts there with 4x the room and allocates far less once scores are cached); otherwise the host retries
            // it with more
            out[q].status = (S.budget > 0 && !S.cached) ? -5 : -3;
            out[q].nAppend = ws.overflow;                             // (which capacity, for MAPLE_DEBUG)
            out[q].bestNode = -2;                                     // marks "handed over because it ran out of room"
            active = false;
        } else if (S.overBudget) {
            out[q].status = -5;                                       // a wide search: the host batch-scores it and re-runs it
            active = false;
        } else if (ws.sp > 0) {
#ifdef MAPLE_SPR_PROFILE
            const long long t0 = wall_clock64();
            const bool notUpd = !ws.st[ws.sp - 1].upd;
            const bool rep = (S.cached || S.lean) && notUpd && !S.stepOnce;
            if (rep) S.replayCached(); else { S.stepOnce = false; S.step(); if (!notUpd) out[q].nSteps++; }
            if (notUpd) out[q].tReplay += wall_clock64() - t0; else out[q].tStep += wall_clock64() - t0;
#else
            if ((S.cached || S.lean) && !ws.st[ws.sp - 1].upd && !S.stepOnce) S.replayCached();
            else { S.stepOnce = false; S.step(); }
#endif
        } else if (S.refineIdx < ws.nB) {
#ifdef MAPLE_SPR_PROFILE
            const long long t0 = wall_clock64();
            out[q].nShortList++;
            int r = S.refine(ws.best[S.refineIdx++]);
            out[q].tRefine += wall_clock64() - t0;
#else
            int r = S.refine(ws.best[S.refineIdx++]);
#endif
            if (r < 0 && !ws.overflow) {                              // the reference raises here; its worker swallows it (M:9703)
                SearchOut &o = out[q];
                o.status = -1; o.nAppend = S.nAppend;
                active = false;
            }
        } else {
            SearchOut &o = out[q];
            o.bestNode = S.bestNode; o.bestScore = S.bestScore;
            o.blen[0] = S.bl0; o.blen[1] = S.bl1; o.blen[2] = S.bl2;
            o.nAppend = S.nAppend;
#ifdef MAPLE_SPR_PROFILE
            o.rprN = (int32_t)(S.tWalk / 100); o.rprNA = (int32_t)(S.tRefSetup / 100);   // (profile: microseconds inside append_walk / list lookup)
            S.tWalk = S.tRefSetup = 0;
#endif
            if (S.trI) *trN = S.trN;
            if (poolW) {                                             // hand bestRemovedPartials out through the pool
                int hOut = S.hBestRpr;
                if (hOut <= -10 && S.rTable)                          // a frame-table list the reference shortened in place
                    for (int k = 0; k < 4; k++)
                        if (S.fShort[k] >= 0 && -(hOut + 10) == S.rTable[S.fShort[k]]) { hOut = S.opShortenCopy(hOut); break; }
                TList rp = S.L(hOut);
                long long ow = (long long)atomicAdd(&poolUsed[0], (unsigned long long)rp.n);
                long long oa = (long long)atomicAdd(&poolUsed[1], (unsigned long long)rp.na);
                if (ow + rp.n <= poolCapW && oa + rp.na <= poolCapA) {
                    for (int k = 0; k < rp.n; k++) poolW[ow + k] = rp.w[k];
                    for (int k = 0; k < rp.na; k++) poolA[oa + k] = rp.aux[k];
                    o.rprWoff = ow; o.rprAoff = oa; o.rprN = rp.n; o.rprNA = rp.na;
                } else o.status = -4;
            }
            // accept rule and the four "same place" vetoes, M:9681-9700
            if (S.bestScore + P.thrPlacement > curLK) {
                bool updated = true;
                int topNode = T.nd[node].up;
                if (S.bestNode == topNode) updated = false;
                while (T.nd[topNode].dist == 0.0 && T.nd[topNode].up >= 0) topNode = T.nd[topNode].up;
                if (S.bestNode == topNode && S.bl1 == 0.0) updated = false;
                const int par = T.nd[node].up;
                const int sib = (T.nd[par].c0 == node) ? T.nd[par].c1 : T.nd[par].c0;
                if (S.bestNode == sib) updated = false;
                if (T.nd[S.bestNode].up == sib && S.bl0 == 0.0) updated = false;
                if (updated) { o.improvement = S.bestScore - curLK; o.placement = S.bestNode; }
            }
            active = false;
        }
        } while (0);
        if constexpr (ASSIST) if (assist) {
            // ---- every lane is here: the scores the searching lanes asked for, one wavefront-wide walk each ----
            unsigned long long req = __ballot(searcher && !done && S.wantApp);
            if (__popcll(req) > coopMax) {
                // many lanes ask at once: each walks its own pair, all of them in lockstep (one walk's latency for all of
                // them, a third of the instructions per score of the wavefront-wide form); the wavefront-wide form is for
                // the few lanes still searching when the others are done
                if (searcher && !done && S.wantApp) {
                    const double v = append_walk(c, S.ref(S.treeList(T.nd[S.appT1].totUp)), S.ref(S.appHRpr), S.isRemovedTip, S.removedBLen);
                    S.mailScore = v; S.mailNode = S.appT1; S.haveMail = true; S.wantApp = false;
                }
                req = 0;
            }
            // (every asking lane looks its two lists up first, all of them at once: the look-ups are chains of dependent loads)
            TList tp{nullptr, nullptr, 0, 0}, tc{nullptr, nullptr, 0, 0};
            if (req && searcher && !done && S.wantApp) { tp = S.L(S.treeList(T.nd[S.appT1].totUp)); tc = S.L(S.appHRpr); }
            while (req) {
                const int r = (int)__ffsll((long long)req) - 1;
                req &= req - 1;
                auto bc64 = [&](unsigned long long x) {
                    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(x >> 32), r) << 32)
                           | (uint32_t)__builtin_amdgcn_readlane((int)x, r);
                };
                const ListRef P{(const uint2 *)bc64((unsigned long long)tp.w), (const double *)bc64((unsigned long long)tp.aux)};
                const ListRef Cq{(const uint2 *)bc64((unsigned long long)tc.w), (const double *)bc64((unsigned long long)tc.aux)};
                const int nP = __builtin_amdgcn_readlane(tp.n, r), nC = __builtin_amdgcn_readlane(tc.n, r);
                const bool tipq = __builtin_amdgcn_readlane(S.isRemovedTip ? 1 : 0, r) != 0;
                const double blq = __longlong_as_double((long long)bc64((unsigned long long)__double_as_longlong(S.removedBLen)));
                const double v = wave_append(c, P, nP, Cq, nC, tipq, blq, wl);
                if ((int)threadIdx.x == r) { S.mailScore = v; S.mailNode = S.appT1; S.haveMail = true; S.wantApp = false; }
            }
            if (!__ballot(!done)) break;
            continue;
        }
        if (!coop) {
            if (done) break;
            continue;
        }
        // ---- wave-level part: every lane is here, lane 0 decides ----
        if (__builtin_amdgcn_readfirstlane(S.wantScan ? 1 : 0)) {
            const int rowU = __builtin_amdgcn_readfirstlane(curRow);
            const double *cs = cacheS + (size_t)rowU * T.n;
            const int32_t *rT = rTable ? rTable + (size_t)rowU * nF : nullptr;
            ScanState st;
            st.best = readfirst_f64(S.bestLKdiff);
            st.nB = __builtin_amdgcn_readfirstlane(ws.nB);
            st.nApp = __builtin_amdgcn_readfirstlane(S.nAppend);
            st.overflow = 0;
            st.shortenSeed = false;
            for (int k = 0; k < 4; k++) st.fShort[k] = __builtin_amdgcn_readfirstlane(S.fShort[k]);
            const int hSeed = __builtin_amdgcn_readfirstlane(S.scanItem.hRpr);
            const unsigned long long brBits = (unsigned long long)ws.best;  // (lane 0's short list)
            BestRec *br = (BestRec *)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(brBits >> 32)) << 32)
                                      | (uint32_t)__builtin_amdgcn_readfirstlane((int)brBits));
            const int capBnow = __builtin_amdgcn_readfirstlane(ws.L.capB);
#ifdef MAPLE_SPR_PROFILE
            const long long tScan0 = wall_clock64();
#endif
            wave_scan_clade(T.scan, T.scanParent, cs, rT, __builtin_amdgcn_readfirstlane(S.scanRank),
                            __builtin_amdgcn_readfirstlane(S.scanFirstScored ? 1 : 0) != 0,
                            __builtin_amdgcn_readfirstlane(S.scanSeedFrame), hSeed, readfirst_f64(S.scanItem.lastLK),
                            __builtin_amdgcn_readfirstlane((int)S.scanItem.fails), P, br, capBnow, slotLK, slotFails,
                            slotOwner, T.scanDepthCap, st, fin.mask ? fin.mask + (size_t)rowU * fin.nWords : nullptr,
                            fin.mask ? fin.prefix + (size_t)rowU * (fin.nWords + 1) : nullptr, T.candBefore, T.cladeVisits);
#ifdef MAPLE_SPR_PROFILE
            if (searcher) out[q].tReplay += wall_clock64() - tScan0;
#endif
            if (searcher) {
                S.bestLKdiff = st.best; ws.nB = st.nB; S.nAppend = st.nApp;
                if (st.overflow && !ws.overflow) ws.overflow = st.overflow;
                for (int k = 0; k < 4; k++) S.fShort[k] = st.fShort[k];
                if (st.shortenSeed) S.opShortenInPlace(hSeed);
            }
            continue;
        }
        if (__builtin_amdgcn_readfirstlane(done ? 1 : 0)) break;
    }
}

#define MAPLE_SPR_KERNEL_ARGS const DevModel *__restrict__ mp, ArenaView av, MutView mv, DevTree T, SearchParams P, int n,            \
    const int32_t *nodes, WsLayout L, LaneBytes LB, uint8_t *wsBase, int32_t *counter, SearchOut *out, uint2 *poolW, double *poolA, \
    unsigned long long *poolUsed, long long poolCapW, long long poolCapA, int traceQuery, int32_t *trI, double *trD, int trCap,     \
    int32_t *trN, int activeLanes, const double *cacheS, int budget, const int32_t *rTable, int nF, const int32_t *cacheRow,        \
    int leanVisits, unsigned long long *ovfUsed, uint8_t *ovfBase, long long ovfChunks, FiniteRows fin
#define MAPLE_SPR_KERNEL_PASS mp, av, mv, T, P, n, nodes, L, LB, wsBase, counter, out, poolW, poolA, poolUsed, poolCapW, poolCapA,   \
    traceQuery, trI, trD, trCap, trN, activeLanes, cacheS, budget, rTable, nF, cacheRow, leanVisits, ovfUsed, ovfBase, ovfChunks, fin
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) MAPLE_SPR_ATTR void k_spr_search(MAPLE_SPR_KERNEL_ARGS)
{
    spr_search_impl<RV, U, SS, false>(MAPLE_SPR_KERNEL_PASS);
}
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) MAPLE_SPR_ATTR void k_spr_search_assisted(MAPLE_SPR_KERNEL_ARGS)
{
    spr_search_impl<RV, U, SS, true>(MAPLE_SPR_KERNEL_PASS);
}

// compaction of scratch lists into the arena: one wavefront per list, coalesced copies
__global__ __launch_bounds__(MAPLE_BLOCK) void k_commit(int n, const uint2 *sw, const double *sa, const int64_t *swoff,
                                                        const int64_t *saoff, const int32_t *n_ent, const int32_t *n_aux,
                                                        const int64_t *dst_w, const int64_t *dst_a, uint2 *words,
                                                        double *aux, const int32_t *rowId, int64_t *t_ent_off,
                                                        int64_t *t_aux_off, int32_t *t_n_ent, int32_t *t_n_aux)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < n; i += nwaves) {
        if (n_ent[i] < 0) continue;
        if (rowId && lane == 0 && rowId[i] >= 0) {                         // the new list's row of the list table
            const int r = rowId[i];
            t_ent_off[r] = dst_w[i]; t_aux_off[r] = dst_a[i]; t_n_ent[r] = n_ent[i]; t_n_aux[r] = n_aux[i];
        }
        const uint2 *s = sw + swoff[i];
        uint2 *d = words + dst_w[i];
        for (int k = lane; k < n_ent[i]; k += 64) d[k] = s[k];
        const double *s2 = sa + saoff[i];
        double *d2 = aux + dst_a[i];
        for (int k = lane; k < n_aux[i]; k += 64) d2[k] = s2[k];
    }
}

// =================================================================================================
// host side
// =================================================================================================
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

extern "C" int maple_abi_version(void) { return MAPLE_ABI_VERSION; }

extern "C" const char *maple_last_error(maple_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int maple_create(maple_ctx **out, int device, int32_t lRef, const uint8_t *refIdx, const double *rootFreqs4,
                            const maple_params *params, uint64_t arena_bytes)
{
    if (!out || !refIdx || !rootFreqs4 || !params || lRef <= 0) return MAPLE_ERR_ARG;
    *out = nullptr;
    maple_ctx *c = new maple_ctx();
    if (const char *e = getenv("MAPLE_DEBUG")) c->tuning.verbose = atoi(e) > 1 ? atoi(e) : 1;   // (the one environment variable: progress lines)
    c->device = device;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device >= ndev) {
        delete c;
        return MAPLE_ERR_HIP;          // no GPU: the product path fails loudly, there is no CPU fallback
    }
    if (hipSetDevice(device) != hipSuccess) { delete c; return MAPLE_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MAPLE_ERR_HIP; }
    c->lRef = lRef;
    c->refIdx.assign(refIdx, refIdx + lRef);
    c->params = *params;
    DevModel &m = c->dm;
    memset(&m, 0, sizeof m);
    m.lRef = lRef;
    for (int i = 0; i < 4; i++) m.rootFreqs[i] = rootFreqs4[i];
    for (int i = 0; i < 4; i++) m.rootFreqsLog[i] = log(rootFreqs4[i]);  // M:3679
    m.globalTotRate = -(double)lRef;                                   // M:3607
    m.minimumCarryOver = DBL_MIN * (1e50);                             // M:3623
    m.thresholdProb = params->thresholdProb;
    double t2 = params->thresholdProb * params->thresholdProb;         // M:3692-3693
    m.thresholdProb4 = t2 * t2;
    m.minBLenSensitivity = params->minBLenSensitivity;
    m.thresholdDiffForUpdate = params->thresholdDiffForUpdate;
    m.thresholdFoldChangeUpdate = params->thresholdFoldChangeUpdate;
    m.defaultBLen = params->defaultBLen;
    if (arena_bytes == 0) arena_bytes = 1ull << 30;
    c->cap_ent = (int64_t)(arena_bytes / 24);
    c->cap_aux = 2 * c->cap_ent;
    c->cap_lists = c->cap_ent / 4 + 1024;
    c->cap_mut = c->cap_ent / 16 + 4096;
    c->cap_mut_lists = c->cap_lists / 4 + 1024;
    bool ok = hipMalloc((void **)&c->d_words, (c->cap_ent + 64) * sizeof(uint2)) == hipSuccess
              && hipMalloc((void **)&c->d_aux, c->cap_aux * sizeof(double)) == hipSuccess
              && hipMalloc((void **)&c->d_ent_off, c->cap_lists * sizeof(int64_t)) == hipSuccess
              && hipMalloc((void **)&c->d_aux_off, c->cap_lists * sizeof(int64_t)) == hipSuccess
              && hipMalloc((void **)&c->d_n_ent, c->cap_lists * sizeof(int32_t)) == hipSuccess
              && hipMalloc((void **)&c->d_n_aux, c->cap_lists * sizeof(int32_t)) == hipSuccess
              && hipMalloc((void **)&c->d_mut3, c->cap_mut * 3 * sizeof(int32_t)) == hipSuccess
              && hipMalloc((void **)&c->d_mut_off, c->cap_mut_lists * sizeof(int64_t)) == hipSuccess
              && hipMalloc((void **)&c->d_mut_cnt, c->cap_mut_lists * sizeof(int32_t)) == hipSuccess
              && hipMalloc((void **)&c->d_cumRate, (lRef + 1) * sizeof(double)) == hipSuccess
              && hipMalloc((void **)&c->d_model, sizeof(DevModel)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_cumBases, (size_t)(lRef + 1) * 4 * sizeof(int32_t)) == hipSuccess;
    if (!ok) { maple_destroy(c); return MAPLE_ERR_NOMEM; }
    {                                                                  // cumulativeBases, M:3669-3674
        std::vector<int32_t> cb((size_t)(lRef + 1) * 4, 0);
        for (int i = 0; i < lRef; i++) {
            for (int k = 0; k < 4; k++) cb[(size_t)(i + 1) * 4 + k] = cb[(size_t)i * 4 + k];
            cb[(size_t)(i + 1) * 4 + refIdx[i]] += 1;
        }
        if (hipMemcpy(c->d_cumBases, cb.data(), cb.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            maple_destroy(c);
            return MAPLE_ERR_HIP;
        }
        m.cumulativeBases = c->d_cumBases;
    }
    *out = c;
    return MAPLE_OK;
}

static void update_scratch_free(maple_ctx *c);   // update_host.h

extern "C" int maple_set_tuning(maple_ctx *c, const maple_tuning *t)
{
    if (!c || !t) return MAPLE_ERR_ARG;
    const int32_t verbose = c->tuning.verbose;
    c->tuning = *t;
    if (!t->verbose && getenv("MAPLE_DEBUG")) c->tuning.verbose = verbose;   // (the environment variable keeps it on)
    return MAPLE_OK;
}

extern "C" int maple_destroy(maple_ctx *c)
{
    if (!c) return MAPLE_OK;
    // this context's device first, and nothing in flight on any of its streams, before anything is freed
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    (void)hipDeviceSynchronize();                                       // (the frontier tier's side stream lives in its scratch)
    update_scratch_free(c);
    frontier_scratch_free(c);
    witness_scratch_free(c);
    for (int k = 0; k < 2; k++) { if (c->stg_h[k]) (void)hipHostFree(c->stg_h[k]); if (c->stg_d[k]) (void)hipFree(c->stg_d[k]); }
    void *ptrs[] = {c->d_cumBases, c->d_rflec, c->d_model, c->d_words, c->d_aux, c->d_ent_off, c->d_aux_off, c->d_n_ent, c->d_n_aux, c->d_mut3, c->d_mut_off,
                    c->d_mut_cnt, c->d_cumRate, c->d_cumErr, c->d_siteRates, c->d_errorRates};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &b : c->s_i32) b.release();
    for (auto &b : c->s_f64) b.release();
    for (auto &b : c->s_u8) b.release();
    for (auto &b : c->s_i64) b.release();
    c->s_words.release(); c->s_aux.release(); c->s_ais.release(); c->s_pool_w.release(); c->s_pool_a.release();
    for (auto &b : c->t_i32) b.release();
    c->t_dist.release(); c->t_tip.release(); c->t_nodes.release(); c->t_scored_col.release(); c->t_scored_frame.release(); c->t_cand_rank.release(); c->s_cand_root.release(); c->s_frame_parent.release(); c->s_frame_node.release(); c->t_scan.release(); c->t_scan_parent.release(); c->t_cand_before.release(); c->t_clade_visits.release(); c->s_fin_mask.release(); c->s_fin_prefix.release(); c->s_tilebest.release(); c->s_comm_u64.release();
    if (c->d_tile_counters) (void)hipFree(c->d_tile_counters);
    c->s_search_ws.release(); c->s_search_ws_big.release(); c->s_search_out.release(); c->s_counter.release(); c->s_cache.release();
    for (auto &b : c->p_i32) b.release();
    for (auto &b : c->p_f64) b.release();
    c->pin_place.release(); c->pin_res.release();
    c->p_score.release(); c->p_i16.release(); c->p_u8.release(); c->p_minor.release(); c->p_from.release();
    if (c->place) {
        PlaceMeta &M = *c->place;
        M.d_scan.release(); M.d_frameOf.release(); M.d_candIdx.release(); M.d_leafIdx.release(); M.d_candList.release(); M.d_candFrame.release();
        M.d_leafList.release(); M.d_leafFrame.release();
        delete c->place;
    }
    for (hipEvent_t e : c->evs) (void)hipEventDestroy(e);
    for (auto &cs : c->candsets) { if (cs.lists) (void)hipFree(cs.lists); if (cs.frame) (void)hipFree(cs.frame); }
    if (c->rccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->rccl_comm);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    c->z_ql.release(); c->z_qt.release(); c->z_qb.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MAPLE_OK;
}

extern "C" int maple_set_model(maple_ctx *c, const double *Q16, const double *siteRates, int usingErrorRate,
                               double errorRateGlobal, const double *errorRates)
{
    if (!c || !Q16) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    DevModel &m = c->dm;
    const int lRef = c->lRef;
    for (int i = 0; i < 16; i++) m.Q[i] = Q16[i];
    m.useRateVariation = siteRates ? 1 : 0;
    m.usingErrorRate = usingErrorRate ? 1 : 0;
    m.errorRateSiteSpecific = (usingErrorRate && errorRates) ? 1 : 0;
    m.errorRate = errorRateGlobal;
    // cumulativeRate, M:6350-6370
    c->h_cumRate.assign(lRef + 1, 0.0);
    for (int i = 0; i < lRef; i++) {
        double nm = Q16[c->refIdx[i] * 5];
        c->h_cumRate[i + 1] = siteRates ? c->h_cumRate[i] + nm * siteRates[i] : c->h_cumRate[i] + nm;
    }
    HIPCK(c, hipMemcpyAsync(c->d_cumRate, c->h_cumRate.data(), (lRef + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    m.cumulativeRate = c->d_cumRate;
    if (siteRates) {
        if (!c->d_siteRates) HIPCK(c, hipMalloc((void **)&c->d_siteRates, lRef * sizeof(double)));
        HIPCK(c, hipMemcpyAsync(c->d_siteRates, siteRates, lRef * sizeof(double), hipMemcpyHostToDevice, c->stream));
        m.siteRates = c->d_siteRates;
    } else m.siteRates = nullptr;
    // cumulativeErrorRate / totError, M:6373-6390
    c->h_cumErr.clear();
    m.cumulativeErrorRate = nullptr;
    m.errorRates = nullptr;
    m.totError = 0.0;
    if (usingErrorRate) {
        if (errorRates) {
            c->h_cumErr.assign(lRef + 1, 0.0);
            for (int i = 0; i < lRef; i++) c->h_cumErr[i + 1] = c->h_cumErr[i] + errorRates[i];
            m.totError = -c->h_cumErr[lRef];
            if (!c->d_errorRates) HIPCK(c, hipMalloc((void **)&c->d_errorRates, lRef * sizeof(double)));
            if (!c->d_cumErr) HIPCK(c, hipMalloc((void **)&c->d_cumErr, (lRef + 1) * sizeof(double)));
            HIPCK(c, hipMemcpyAsync(c->d_errorRates, errorRates, lRef * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPCK(c, hipMemcpyAsync(c->d_cumErr, c->h_cumErr.data(), (lRef + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
            m.errorRates = c->d_errorRates;
            m.cumulativeErrorRate = c->d_cumErr;
        } else m.totError = -errorRateGlobal * lRef;
    }
    m.rootFreqsLogErrorCumulative = nullptr;
    if (usingErrorRate) {                                              // M:6380-6389 (note the 0.333333 of the table)
        std::vector<double> acc(lRef + 1, 0.0);
        for (int i = 0; i < lRef; i++) {
            double e = errorRates ? errorRates[i] : errorRateGlobal;
            acc[i + 1] = acc[i] + log(m.rootFreqs[c->refIdx[i]] * (1.0 - 1.33333 * e) + 0.333333 * e);
        }
        if (!c->d_rflec) HIPCK(c, hipMalloc((void **)&c->d_rflec, (lRef + 1) * sizeof(double)));
        HIPCK(c, hipMemcpy(c->d_rflec, acc.data(), (lRef + 1) * sizeof(double), hipMemcpyHostToDevice));
        m.rootFreqsLogErrorCumulative = c->d_rflec;
    }
    HIPCK(c, hipMemcpyAsync(c->d_model, &c->dm, sizeof(DevModel), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->model_set = true;
    return MAPLE_OK;
}

extern "C" int maple_get_model(maple_ctx *c, double *cumulativeRate, double *cumulativeErrorRate, double *totError)
{
    if (!c) return MAPLE_ERR_ARG;
    if (!c->model_set) return fail(c, MAPLE_ERR_STATE, "model not set");
    if (cumulativeRate) memcpy(cumulativeRate, c->h_cumRate.data(), c->h_cumRate.size() * sizeof(double));
    if (cumulativeErrorRate && !c->h_cumErr.empty())
        memcpy(cumulativeErrorRate, c->h_cumErr.data(), c->h_cumErr.size() * sizeof(double));
    if (totError) *totError = c->dm.totError;
    return MAPLE_OK;
}

// ---- arena --------------------------------------------------------------------------------------
static int push_list_rows(maple_ctx *c, int32_t n, const int64_t *ent_off_abs, const int64_t *aux_off_abs,
                          const int32_t *n_ent, const int32_t *n_aux)
{
    int64_t first = (int64_t)c->h_n_ent.size();
    if (first + n > c->cap_lists) return fail(c, MAPLE_ERR_NOMEM, "list table full (%lld)", (long long)c->cap_lists);
    c->h_ent_off.insert(c->h_ent_off.end(), ent_off_abs, ent_off_abs + n);
    c->h_aux_off.insert(c->h_aux_off.end(), aux_off_abs, aux_off_abs + n);
    c->h_n_ent.insert(c->h_n_ent.end(), n_ent, n_ent + n);
    c->h_n_aux.insert(c->h_n_aux.end(), n_aux, n_aux + n);
    HIPCK(c, hipMemcpyAsync(c->d_ent_off + first, ent_off_abs, n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->d_aux_off + first, aux_off_abs, n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->d_n_ent + first, n_ent, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->d_n_aux + first, n_aux, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));   // source vectors are caller temporaries
    return MAPLE_OK;
}

extern "C" int maple_lists_upload(maple_ctx *c, int32_t n, const int64_t *ent_off, const int32_t *pos,
                                  const uint32_t *meta, const int64_t *aux_off, const double *aux, int32_t *first_id)
{
    if (!c || n < 0 || !ent_off || !aux_off || !first_id) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    *first_id = (int32_t)c->h_n_ent.size();
    if (n == 0) return MAPLE_OK;
    const int64_t ne = ent_off[n] - ent_off[0], na = aux_off[n] - aux_off[0];
    if (c->used_ent + ne > c->cap_ent || c->used_aux + na > c->cap_aux)
        return fail(c, MAPLE_ERR_NOMEM, "arena full: need %lld entries / %lld aux", (long long)ne, (long long)na);
    std::vector<uint2> w((size_t)ne);
    for (int64_t k = 0; k < ne; k++) w[k] = make_uint2((uint32_t)pos[ent_off[0] + k], meta[ent_off[0] + k]);
    HIPCK(c, hipMemcpyAsync(c->d_words + c->used_ent, w.data(), ne * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    if (na) HIPCK(c, hipMemcpyAsync(c->d_aux + c->used_aux, aux + aux_off[0], na * sizeof(double), hipMemcpyHostToDevice, c->stream));
    std::vector<int64_t> eo(n), ao(n);
    std::vector<int32_t> cnt(n), cna(n);
    for (int i = 0; i < n; i++) {
        eo[i] = c->used_ent + (ent_off[i] - ent_off[0]);
        ao[i] = c->used_aux + (aux_off[i] - aux_off[0]);
        cnt[i] = (int32_t)(ent_off[i + 1] - ent_off[i]);
        cna[i] = (int32_t)(aux_off[i + 1] - aux_off[i]);
        if (cnt[i] <= 0) return fail(c, MAPLE_ERR_ARG, "list %d is empty", i);
    }
    int rc = push_list_rows(c, n, eo.data(), ao.data(), cnt.data(), cna.data());
    if (rc) return rc;
    c->used_ent += ne;
    c->used_aux += na;
    return MAPLE_OK;
}

static int check_ids(maple_ctx *c, int32_t n, const int32_t *ids, bool allowNeg, const char *what);

static int check_ids(maple_ctx *c, int32_t n, const int32_t *ids, bool allowNeg, const char *what)
{
    const int32_t nl = (int32_t)c->h_n_ent.size();
    for (int i = 0; i < n; i++)
        if (ids[i] >= nl || (ids[i] < 0 && !allowNeg))
            return fail(c, MAPLE_ERR_ARG, "%s[%d] = %d is not a list id (have %d)", what, i, ids[i], nl);
    return MAPLE_OK;
}

extern "C" int maple_lists_sizes(maple_ctx *c, int32_t n, const int32_t *ids, int32_t *n_ent, int32_t *n_aux)
{
    if (!c || n < 0 || !ids) return MAPLE_ERR_ARG;
    int rc = check_ids(c, n, ids, false, "ids");
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        if (n_ent) n_ent[i] = c->h_n_ent[ids[i]];
        if (n_aux) n_aux[i] = c->h_n_aux[ids[i]];
    }
    return MAPLE_OK;
}

extern "C" int maple_lists_download(maple_ctx *c, int32_t n, const int32_t *ids, const int64_t *ent_off, int32_t *pos,
                                    uint32_t *meta, const int64_t *aux_off, double *aux)
{
    if (!c || n < 0 || !ids || !ent_off || !aux_off) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int rc = check_ids(c, n, ids, false, "ids");
    if (rc) return rc;
    // lists that sit back to back in the arena AND in the caller's buffers (the usual case: a tree's lists, downloaded in id
    // order) move with one copy per run instead of one per list
    std::vector<uint2> w;
    int i = 0;
    while (i < n) {
        int j = i;
        int64_t ne = 0, na = 0;
        while (j < n) {
            const int id = ids[j];
            if (j > i) {
                const int pid = ids[j - 1];
                if (c->h_ent_off[id] != c->h_ent_off[pid] + c->h_n_ent[pid] || c->h_aux_off[id] != c->h_aux_off[pid] + c->h_n_aux[pid]
                    || ent_off[j] != ent_off[j - 1] + c->h_n_ent[pid] || aux_off[j] != aux_off[j - 1] + c->h_n_aux[pid])
                    break;
            }
            ne += c->h_n_ent[id]; na += c->h_n_aux[id];
            j++;
            if (ne > (int64_t)32 << 20) break;
        }
        const int id0 = ids[i];
        w.resize((size_t)ne);
        HIPCK(c, hipMemcpyAsync(w.data(), c->d_words + c->h_ent_off[id0], ne * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
        if (na) HIPCK(c, hipMemcpyAsync(aux + aux_off[i], c->d_aux + c->h_aux_off[id0], na * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        for (int64_t k = 0; k < ne; k++) { pos[ent_off[i] + k] = (int32_t)w[k].x; meta[ent_off[i] + k] = w[k].y; }
        i = j;
    }
    return MAPLE_OK;
}

extern "C" int maple_set_fatal_policy(maple_ctx *c, int tolerate)
{
    if (!c) return MAPLE_ERR_ARG;
    c->tolerate_fatal = tolerate != 0;
    return MAPLE_OK;
}

// A mark is the number of genome lists in its low 40 bits and the number of MAT mutation lists above: releasing it drops
// both kinds of temporaries.
extern "C" int maple_arena_mark(maple_ctx *c, int64_t *mark)
{
    if (!c || !mark) return MAPLE_ERR_ARG;
    *mark = (int64_t)c->h_n_ent.size() | ((int64_t)c->h_mut_cnt.size() << 40);
    return MAPLE_OK;
}

extern "C" int maple_arena_release(maple_ctx *c, int64_t markBoth)
{
    if (!c || markBoth < 0) return MAPLE_ERR_ARG;
    const int64_t mark = markBoth & (((int64_t)1 << 40) - 1), mmark = markBoth >> 40;
    if (mark > (int64_t)c->h_n_ent.size() || mmark > (int64_t)c->h_mut_cnt.size()) return MAPLE_ERR_ARG;
    if (mmark < (int64_t)c->h_mut_cnt.size()) {
        c->used_mut = c->h_mut_off[mmark];
        c->h_mut_off.resize(mmark); c->h_mut_cnt.resize(mmark);
    }
    if (mark == (int64_t)c->h_n_ent.size()) return MAPLE_OK;
    if (mark < c->cand_root_end) c->cand_root_end = -1;                 // (the candidates' root-frame copies go with the release)
    int64_t ue = c->h_ent_off[mark], ua = c->h_aux_off[mark];
    {   // a list below the mark that maple_lists_update moved to the end of the arena keeps its (new) room
        size_t k = 0;
        for (int32_t id : c->relocated) {
            if (id >= mark) continue;
            ue = std::max<int64_t>(ue, c->h_ent_off[id] + c->h_n_ent[id]);
            ua = std::max<int64_t>(ua, c->h_aux_off[id] + c->h_n_aux[id]);
            c->relocated[k++] = id;
        }
        c->relocated.resize(k);
    }
    c->used_ent = ue;
    c->used_aux = ua;
    c->h_ent_off.resize(mark); c->h_aux_off.resize(mark); c->h_n_ent.resize(mark); c->h_n_aux.resize(mark);
    if (c->place && c->place->rootVect >= mark) c->place->rootVect = -1;     // the cached root vector went with the release
    return MAPLE_OK;
}

static int settle(maple_ctx *c);
static int grid_for(int n);
// Keep only the lists `live` (ids in any order, no duplicates, -1 entries allowed and kept as -1): they are copied to the
// bottom of the arena in the order given and renumbered 0, 1, 2, ...; newIds[i] = the new id of live[i].  Everything else
// -- every other list id, every arena mark, resident candidate sets and the uploaded tree -- is gone: upload the tree again
// with the new ids.  What a long run of maple_update_partials / single-sample placements needs from time to time: every
// replaced list keeps its room until then.
extern "C" int maple_arena_compact(maple_ctx *c, int64_t nLive, const int32_t *live, int32_t *newIds)
{
    if (!c || nLive < 0 || (nLive && (!live || !newIds))) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    TRY(settle(c));
    const int64_t nl = (int64_t)c->h_n_ent.size();
    std::vector<uint8_t> seen((size_t)nl, 0);
    int64_t totE = 0, totA = 0, m = 0;
    for (int64_t i = 0; i < nLive; i++) {
        if (live[i] == -1) continue;
        if (live[i] < 0 || live[i] >= nl) return fail(c, MAPLE_ERR_ARG, "live[%lld] = %d is not a list id", (long long)i, live[i]);
        if (seen[live[i]]) return fail(c, MAPLE_ERR_ARG, "list %d is named twice", live[i]);
        seen[live[i]] = 1;
        totE += c->h_n_ent[live[i]]; totA += c->h_n_aux[live[i]];
        m++;
    }
    // gather into scratch (one wavefront per list), then one straight copy back to the bottom of the arena
    std::vector<int64_t> srcW((size_t)m), srcA((size_t)m), dstW((size_t)m), dstA((size_t)m);
    std::vector<int32_t> ne((size_t)m), na((size_t)m);
    int64_t e = 0, a = 0, k = 0;
    for (int64_t i = 0; i < nLive; i++) {
        if (live[i] == -1) { newIds[i] = -1; continue; }
        const int32_t id = live[i];
        srcW[k] = c->h_ent_off[id]; srcA[k] = c->h_aux_off[id]; dstW[k] = e; dstA[k] = a; ne[k] = c->h_n_ent[id]; na[k] = c->h_n_aux[id];
        e += ne[k]; a += na[k];
        newIds[i] = (int32_t)k++;
    }
    if (m) {
        HIPCK(c, c->s_words.reserve((size_t)totE));
        HIPCK(c, c->s_aux.reserve((size_t)std::max<int64_t>(totA, 1)));
        for (int b = 0; b < 6; b++) HIPCK(c, c->s_i64[b].reserve((size_t)m));
        HIPCK(c, c->s_i32[2].reserve((size_t)m));
        HIPCK(c, c->s_i32[3].reserve((size_t)m));
        HIPCK(c, hipMemcpyAsync(c->s_i64[0].p, srcW.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i64[1].p, srcA.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i64[2].p, dstW.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i64[3].p, dstA.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i32[2].p, ne.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i32[3].p, na.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        // (k_commit: list i of the source (words, aux) at swoff / saoff -> destination arrays at dst_w / dst_a)
        hipLaunchKernelGGL(k_commit, dim3(grid_for((int)std::min<int64_t>(m, 1 << 20) * 64)), dim3(MAPLE_BLOCK), 0, c->stream, (int)m, c->d_words,
                           c->d_aux, c->s_i64[0].p, c->s_i64[1].p, c->s_i32[2].p, c->s_i32[3].p, c->s_i64[2].p, c->s_i64[3].p, c->s_words.p,
                           c->s_aux.p, (const int32_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
        HIPCK(c, hipGetLastError());
        HIPCK(c, hipMemcpyAsync(c->d_words, c->s_words.p, (size_t)totE * sizeof(uint2), hipMemcpyDeviceToDevice, c->stream));
        if (totA) HIPCK(c, hipMemcpyAsync(c->d_aux, c->s_aux.p, (size_t)totA * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->d_ent_off, dstW.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->d_aux_off, dstA.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->d_n_ent, ne.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->d_n_aux, na.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    c->h_ent_off.assign(dstW.begin(), dstW.end()); c->h_aux_off.assign(dstA.begin(), dstA.end());
    c->h_n_ent.assign(ne.begin(), ne.end()); c->h_n_aux.assign(na.begin(), na.end());
    c->used_ent = totE; c->used_aux = totA;
    c->relocated.clear();
    c->tree_set = false; c->tree_stale = false; c->nodes_current = false; c->scan_valid = false; c->cand_root_end = -1;
    if (c->place) { c->place->valid = false; c->place->rootVect = -1; }
    for (auto &cs : c->candsets) {
        if (cs.lists) (void)hipFree(cs.lists);
        if (cs.frame) (void)hipFree(cs.frame);
        cs = maple_ctx::CandSet{};
    }
    return MAPLE_OK;
}

extern "C" int maple_arena_stats(maple_ctx *c, int64_t *n_lists, int64_t *n_entries, int64_t *n_aux, int64_t *cap_entries)
{
    if (!c) return MAPLE_ERR_ARG;
    if (n_lists) *n_lists = (int64_t)c->h_n_ent.size();
    if (n_entries) *n_entries = c->used_ent;
    if (n_aux) *n_aux = c->used_aux;
    if (cap_entries) *cap_entries = c->cap_ent;
    return MAPLE_OK;
}

extern "C" int maple_mutations_upload(maple_ctx *c, int32_t n, const int64_t *off, const int32_t *mut3, int32_t *first_id)
{
    if (!c || n < 0 || !off || !first_id) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    *first_id = (int32_t)c->h_mut_cnt.size();
    if (n == 0) return MAPLE_OK;
    const int64_t nm = off[n] - off[0];
    if (c->used_mut + nm > c->cap_mut || (int64_t)c->h_mut_cnt.size() + n > c->cap_mut_lists)
        return fail(c, MAPLE_ERR_NOMEM, "mutation arena full");
    if (nm) HIPCK(c, hipMemcpyAsync(c->d_mut3 + 3 * c->used_mut, mut3 + 3 * off[0], nm * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    std::vector<int64_t> o(n);
    std::vector<int32_t> cnt(n);
    for (int i = 0; i < n; i++) { o[i] = c->used_mut + (off[i] - off[0]); cnt[i] = (int32_t)(off[i + 1] - off[i]); }
    int64_t first = *first_id;
    HIPCK(c, hipMemcpyAsync(c->d_mut_off + first, o.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->d_mut_cnt + first, cnt.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->h_mut_off.insert(c->h_mut_off.end(), o.begin(), o.end());
    c->h_mut_cnt.insert(c->h_mut_cnt.end(), cnt.begin(), cnt.end());
    c->used_mut += nm;
    return MAPLE_OK;
}

// ---- helpers for batch calls -----------------------------------------------------------------------
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

// Move freshly produced scratch lists into the arena and hand out ids (or -1 for None).  d_woff / d_aoff: the per-item
// scratch offsets, already on the device.  One synchronisation (the sizes come back), then one staged copy (destinations and
// list ids) and the copy kernel, which also writes the new rows of the device-side list table; nothing waits for it.
static int commit_known(maple_ctx *c, int32_t n, const int64_t *d_woff, const int64_t *d_aoff, const int32_t *d_n_ent,
                        const int32_t *d_n_aux, const std::vector<int32_t> &ne, const std::vector<int32_t> &na, int32_t *outList,
                        const uint2 *srcW, const double *srcA);
static int commit_lists(maple_ctx *c, int32_t n, const int64_t *d_woff, const int64_t *d_aoff, int32_t *d_n_ent, int32_t *d_n_aux,
                        int32_t *outList, const uint2 *srcW = nullptr, const double *srcA = nullptr)
{
    HIPCK(c, c->pin_res.reserve((size_t)2 * n * sizeof(int32_t)));
    int32_t *h = (int32_t *)c->pin_res.p;
    HIPCK(c, hipMemcpyAsync(h, d_n_ent, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(h + n, d_n_aux, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const std::vector<int32_t> ne(h, h + n), na(h + n, h + 2 * (size_t)n);
    return commit_known(c, n, d_woff, d_aoff, d_n_ent, d_n_aux, ne, na, outList, srcW, srcA);
}

// ... with the sizes already on the host (ne[i] == -1: nothing to commit for item i)
static int commit_known(maple_ctx *c, int32_t n, const int64_t *d_woff, const int64_t *d_aoff, const int32_t *d_n_ent,
                        const int32_t *d_n_aux, const std::vector<int32_t> &ne, const std::vector<int32_t> &na, int32_t *outList,
                        const uint2 *srcW, const double *srcA)
{
    if (!srcW) { srcW = c->s_words.p; srcA = c->s_aux.p; }             // the batch operators' shared scratch
    std::vector<int64_t> dw(n, -1), da(n, -1);
    std::vector<int32_t> rowId(n, -1);
    int64_t ue = c->used_ent, ua = c->used_aux;
    const int64_t first = (int64_t)c->h_n_ent.size();
    int32_t next_id = (int32_t)first;
    for (int i = 0; i < n; i++) {
        if (ne[i] == -1) { outList[i] = -1; continue; }
        if (ne[i] < 0) {
            if (c->tolerate_fatal) { outList[i] = -2; continue; }
            return fail(c, MAPLE_ERR_FATAL, "item %d hit a state the reference treats as fatal (%d)", i, ne[i]);
        }
        dw[i] = ue; da[i] = ua;
        ue += ne[i]; ua += na[i];
        rowId[i] = next_id;
        outList[i] = next_id++;
    }
    if (ue > c->cap_ent || ua > c->cap_aux) return fail(c, MAPLE_ERR_NOMEM, "arena full while committing %d lists", n);
    if (next_id > c->cap_lists) return fail(c, MAPLE_ERR_NOMEM, "list table full (%lld)", (long long)c->cap_lists);
    for (int i = 0; i < n; i++) {
        if (rowId[i] < 0) continue;
        c->h_ent_off.push_back(dw[i]); c->h_aux_off.push_back(da[i]); c->h_n_ent.push_back(ne[i]); c->h_n_aux.push_back(na[i]);
    }
    STAGE(d_dw, c, dw.data(), n);
    STAGE(d_da, c, da.data(), n);
    STAGE(d_row, c, rowId.data(), n);
    TRY(stage_flush(c));
    int waves_per_block = MAPLE_BLOCK / 64;
    int g = (n + waves_per_block - 1) / waves_per_block;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_commit, dim3(g), dim3(MAPLE_BLOCK), 0, c->stream, n, srcW, srcA, d_woff, d_aoff, d_n_ent, d_n_aux, d_dw,
                       d_da, c->d_words, c->d_aux, d_row, c->d_ent_off, c->d_aux_off, c->d_n_ent, c->d_n_aux);
    HIPCK(c, hipGetLastError());
    c->used_ent = ue;
    c->used_aux = ua;
    c->commit_pending = true;
    return MAPLE_OK;
}

// lists committed on the library's stream are not yet visible to work on another stream: wait once
static int settle(maple_ctx *c)
{
    if (c->commit_pending) { HIPCK(c, hipStreamSynchronize(c->stream)); c->commit_pending = false; }
    return MAPLE_OK;
}

__global__ void k_set_rows(int n, const int32_t *ids, const int64_t *eo, const int64_t *ao, const int32_t *ne, const int32_t *na,
                           int64_t *t_ent_off, int64_t *t_aux_off, int32_t *t_n_ent, int32_t *t_n_aux)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int r = ids[i]; t_ent_off[r] = eo[i]; t_aux_off[r] = ao[i]; t_n_ent[r] = ne[i]; t_n_aux[r] = na[i]; }
}

// New contents for EXISTING lists: ids[i] keeps its number (every table that refers to it -- tree columns, candidate sets,
// an uploaded tree -- stays valid) and from now on names the new words.  A list that fits in the room of the old one is
// overwritten in place; otherwise it gets fresh room at the end of the arena (the old room is given back by the next
// maple_arena_release past it, like everything else in a bump arena).
extern "C" int maple_lists_update(maple_ctx *c, int32_t n, const int32_t *ids, const int64_t *ent_off, const int32_t *pos,
                                  const uint32_t *meta, const int64_t *aux_off, const double *aux)
{
    if (!c || n < 0 || !ids || !ent_off || !aux_off || (n && (!pos || !meta))) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(check_ids(c, n, ids, false, "list"));
    std::vector<int64_t> eo(n), ao(n);
    std::vector<int32_t> cnt(n), cna(n);
    int64_t ue = c->used_ent, ua = c->used_aux;
    for (int i = 0; i < n; i++) {
        cnt[i] = (int32_t)(ent_off[i + 1] - ent_off[i]);
        cna[i] = (int32_t)(aux_off[i + 1] - aux_off[i]);
        if (cnt[i] <= 0) return fail(c, MAPLE_ERR_ARG, "list %d is empty", i);
        const int32_t id = ids[i];
        if (cnt[i] <= c->h_n_ent[id] && cna[i] <= c->h_n_aux[id]) { eo[i] = c->h_ent_off[id]; ao[i] = c->h_aux_off[id]; }
        else { eo[i] = ue; ao[i] = ua; ue += cnt[i]; ua += cna[i]; }
    }
    if (ue > c->cap_ent || ua > c->cap_aux) return fail(c, MAPLE_ERR_NOMEM, "arena full while updating %d lists", n);
    // (maple_arena_release must not free their room; a list relocated again and again is listed once: the list is kept sorted)
    for (int i = 0; i < n; i++)
        if (eo[i] >= c->used_ent) {
            auto at = std::lower_bound(c->relocated.begin(), c->relocated.end(), ids[i]);
            if (at == c->relocated.end() || *at != ids[i]) c->relocated.insert(at, ids[i]);
        }
    TRY(settle(c));
    std::vector<uint2> w;
    for (int i = 0; i < n; i++) {
        w.resize((size_t)cnt[i]);
        for (int k = 0; k < cnt[i]; k++) w[k] = make_uint2((uint32_t)pos[ent_off[i] + k], meta[ent_off[i] + k]);
        HIPCK(c, hipMemcpyAsync(c->d_words + eo[i], w.data(), (size_t)cnt[i] * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
        if (cna[i]) HIPCK(c, hipMemcpyAsync(c->d_aux + ao[i], aux + aux_off[i], (size_t)cna[i] * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));                       // (w is reused)
        const int32_t id = ids[i];
        c->h_ent_off[id] = eo[i]; c->h_aux_off[id] = ao[i]; c->h_n_ent[id] = cnt[i]; c->h_n_aux[id] = cna[i];
    }
    TRY(stage_begin(c, (size_t)n * 32 + 256));
    STAGE(dids, c, ids, n); STAGE(deo, c, eo.data(), n); STAGE(dao, c, ao.data(), n); STAGE(dne, c, cnt.data(), n); STAGE(dna, c, cna.data(), n);
    TRY(stage_flush(c));
    hipLaunchKernelGGL(k_set_rows, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dids, deo, dao, dne, dna, c->d_ent_off,
                       c->d_aux_off, c->d_n_ent, c->d_n_aux);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->used_ent = ue;
    c->used_aux = ua;
    return MAPLE_OK;
}

// ---- batched operators -----------------------------------------------------------------------------
extern "C" int maple_append_batch(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl, const uint8_t *tip,
                                  const double *bl, double *out)
{
    if (!c || n < 0 || !pl || !cl || !tip || !bl || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, pl, false, "parentList"));
    TRY(check_ids(c, n, cl, false, "childList"));
    TRY(stage_begin(c, (size_t)n * 32 + 256));
    STAGE(dpl, c, pl, n); STAGE(dcl, c, cl, n); STAGE(dtip, c, tip, n); STAGE(dbl, c, bl, n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_f64[1].reserve(n));
    if (n <= wave_item_max(c, MAPLE_WAVE_PAIRS_MAX))
        DISPATCH3(c, k_wave_append, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dpl, dcl, dtip, dbl, c->s_f64[1].p));
    else
        DISPATCH3(c, k_append, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dpl, dcl, dtip, dbl, c->s_f64[1].p));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->s_f64[1].p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_merge_batch(maple_ctx *c, int32_t n, const int32_t *l1, const double *b1, const uint8_t *t1,
                                 const int32_t *l2, const double *b2, const uint8_t *t2, const uint8_t *ud,
                                 const int32_t *nm1, const int32_t *nm2, int32_t *outList, double *outLK)
{
    if (!c || n < 0 || !l1 || !b1 || !t1 || !l2 || !b2 || !t2 || !ud || !outList) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, l1, false, "list1"));
    TRY(check_ids(c, n, l2, false, "list2"));
    std::vector<int64_t> woff(n), aoff(n);
    int64_t tot = 0;
    for (int i = 0; i < n; i++) {
        woff[i] = tot; aoff[i] = 5 * tot;
        tot += (int64_t)c->h_n_ent[l1[i]] + c->h_n_ent[l2[i]];
    }
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    TRY(stage_begin(c, (size_t)n * 96 + 512));
    STAGE(dl1, c, l1, n); STAGE(dl2, c, l2, n); STAGE(db1, c, b1, n); STAGE(db2, c, b2, n);
    STAGE(dt1, c, t1, n); STAGE(dt2, c, t2, n); STAGE(dud, c, ud, n);
    const int32_t *dnm1 = nullptr, *dnm2 = nullptr;
    if (nm1) { STAGE(p1, c, nm1, n); dnm1 = p1; }
    if (nm2) { STAGE(p2, c, nm2, n); dnm2 = p2; }
    STAGE(dwo, c, woff.data(), n); STAGE(dao, c, aoff.data(), n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_i32[2].reserve(n));
    HIPCK(c, c->s_i32[3].reserve(n));
    double *dlk = nullptr;
    if (outLK) { HIPCK(c, c->s_f64[2].reserve(n)); dlk = c->s_f64[2].p; }
    OutSpec o{c->s_words.p, c->s_aux.p, dwo, dao, c->s_i32[2].p, c->s_i32[3].p};
    if (!outLK && n <= wave_item_max(c, MAPLE_WAVE_PAIRS_MAX))
        DISPATCH3(c, k_merge_wave, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dl1, db1, dt1, dl2, db2, dt2, dud, o));
    else
        DISPATCH3(c, k_merge, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dl1, db1, dt1, dl2, db2, dt2, dud, dnm1,
                                                                             dnm2, o, dlk));
    HIPCK(c, hipGetLastError());
    if (outLK) HIPCK(c, hipMemcpyAsync(outLK, dlk, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return commit_lists(c, n, dwo, dao, c->s_i32[2].p, c->s_i32[3].p, outList);
}

extern "C" int maple_blen_batch(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl, const uint8_t *tip,
                                double *t, uint8_t *isFalse)
{
    if (!c || n < 0 || !pl || !cl || !tip || !t || !isFalse) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, pl, false, "parentList"));
    TRY(check_ids(c, n, cl, false, "childList"));
    std::vector<int64_t> aoff(n);
    int64_t tot = 0;
    for (int i = 0; i < n; i++) { aoff[i] = tot; tot += (int64_t)c->h_n_ent[pl[i]] + c->h_n_ent[cl[i]]; }
    HIPCK(c, c->s_ais.reserve((size_t)tot));
    TRY(stage_begin(c, (size_t)n * 32 + 256));
    STAGE(dpl, c, pl, n); STAGE(dcl, c, cl, n); STAGE(dtip, c, tip, n); STAGE(dao, c, aoff.data(), n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_f64[0].reserve(n));
    HIPCK(c, c->s_u8[1].reserve(n));
    if (n <= wave_item_max(c, MAPLE_WAVE_PAIRS_MAX))
        DISPATCH3(c, k_blen_wave, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dpl, dcl, dtip, c->s_ais.p, dao, c->s_f64[0].p,
                                                             c->s_u8[1].p));
    else
        DISPATCH3(c, k_blen, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dpl, dcl, dtip, c->s_ais.p, dao,
                                                                            c->s_f64[0].p, c->s_u8[1].p));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(t, c->s_f64[0].p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(isFalse, c->s_u8[1].p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_differ_batch(maple_ctx *c, int32_t n, const int32_t *l1, const int32_t *l2, uint8_t *out)
{
    if (!c || n < 0 || !l1 || !l2 || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, l1, false, "list1"));
    TRY(check_ids(c, n, l2, true, "list2"));
    TRY(stage_begin(c, (size_t)n * 8 + 128));
    STAGE(dl1, c, l1, n); STAGE(dl2, c, l2, n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_u8[0].reserve(n));
    if (n <= wave_item_max(c, MAPLE_WAVE_PAIRS_MAX))
        DISPATCH3(c, k_differ_wave, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dl1, dl2, c->s_u8[0].p));
    else
        DISPATCH3(c, k_differ, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dl1, dl2, c->s_u8[0].p));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->s_u8[0].p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

// ---- resident candidate sets (placement of one query at a time against the whole tree) --------------------------
extern "C" int maple_candset_create(maple_ctx *c, int32_t n, const int32_t *lists, const int32_t *frameIdx, int32_t nFrames,
                                    int32_t *setId)
{
    if (!c || n <= 0 || !lists || !frameIdx || nFrames <= 0 || !setId) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    TRY(check_ids(c, n, lists, false, "lists"));
    for (int i = 0; i < n; i++)
        if (frameIdx[i] < 0 || frameIdx[i] >= nFrames) return fail(c, MAPLE_ERR_ARG, "frameIdx[%d] out of range", i);
    maple_ctx::CandSet cs;
    cs.n = n; cs.nFrames = nFrames;
    HIPCK(c, hipMalloc((void **)&cs.lists, n * sizeof(int32_t)));
    HIPCK(c, hipMalloc((void **)&cs.frame, n * sizeof(int32_t)));
    HIPCK(c, hipMemcpy(cs.lists, lists, n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(cs.frame, frameIdx, n * sizeof(int32_t), hipMemcpyHostToDevice));
    *setId = (int32_t)c->candsets.size();
    c->candsets.push_back(cs);
    return MAPLE_OK;
}

extern "C" int maple_candset_destroy(maple_ctx *c, int32_t setId)
{
    if (!c || setId < 0 || setId >= (int32_t)c->candsets.size()) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    auto &cs = c->candsets[setId];
    if (cs.lists) (void)hipFree(cs.lists);
    if (cs.frame) (void)hipFree(cs.frame);
    cs.lists = cs.frame = nullptr;
    cs.n = 0;
    return MAPLE_OK;
}

extern "C" int maple_append_candset(maple_ctx *c, int32_t setId, const int32_t *frameLists, int isTipC, double bLen, double *out)
{
    if (!c || setId < 0 || setId >= (int32_t)c->candsets.size() || !frameLists || !out) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    const auto &cs = c->candsets[setId];
    if (!cs.lists) return fail(c, MAPLE_ERR_ARG, "candidate set %d has been destroyed", setId);
    TRY(check_ids(c, cs.nFrames, frameLists, false, "frameLists"));
    TRY(h2d(c, c->s_i32[0], frameLists, (size_t)cs.nFrames));
    HIPCK(c, c->s_f64[0].reserve(cs.n));
    hipEvent_t e0, e1;
    TRY(ev_pair(c, &e0, &e1, MAPLE_K_OTHER, (double)cs.n, 0.0));
    HIPCK(c, hipEventRecord(e0, c->stream));
    DISPATCH3(c, k_append_candset, <<<grid_for(cs.n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), cs.n, cs.lists, cs.frame,
                                                                                  c->s_i32[0].p, isTipC, bLen, c->s_f64[0].p));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(e1, c->stream));
    HIPCK(c, hipMemcpyAsync(out, c->s_f64[0].p, cs.n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_minor_candset(maple_ctx *c, int32_t setId, const int32_t *frameLists, int onlyFindIdentical, uint8_t *out)
{
    if (!c || setId < 0 || setId >= (int32_t)c->candsets.size() || !frameLists || !out) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    const auto &cs = c->candsets[setId];
    if (!cs.lists) return fail(c, MAPLE_ERR_ARG, "candidate set %d has been destroyed", setId);
    TRY(check_ids(c, cs.nFrames, frameLists, false, "frameLists"));
    TRY(h2d(c, c->s_i32[0], frameLists, (size_t)cs.nFrames));
    HIPCK(c, c->s_u8[0].reserve(cs.n));
    hipLaunchKernelGGL(k_minor_candset, dim3(grid_for(cs.n)), dim3(MAPLE_BLOCK), 0, c->stream, c->lRef, view(c), cs.n, cs.lists,
                       cs.frame, c->s_i32[0].p, onlyFindIdentical, c->s_u8[0].p);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->s_u8[0].p, cs.n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_root_prob_batch(maple_ctx *c, int32_t n, const int32_t *l, double *out)
{
    if (!c || n < 0 || !l || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, l, false, "list"));
    TRY(h2d(c, c->s_i32[0], l, (size_t)n));
    HIPCK(c, c->s_f64[0].reserve(n));
    DISPATCH3(c, k_root_prob, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, c->s_i32[0].p, c->s_f64[0].p));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->s_f64[0].p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_minor_batch(maple_ctx *c, int32_t n, const int32_t *l1, const int32_t *l2, int onlyFindIdentical,
                                 uint8_t *out)
{
    if (!c || n < 0 || !l1 || !l2 || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(check_ids(c, n, l1, false, "list1"));
    TRY(check_ids(c, n, l2, false, "list2"));
    TRY(h2d(c, c->s_i32[0], l1, (size_t)n));
    TRY(h2d(c, c->s_i32[1], l2, (size_t)n));
    HIPCK(c, c->s_u8[0].reserve(n));
    hipLaunchKernelGGL(k_minor, dim3(grid_for(n)), dim3(MAPLE_BLOCK), 0, c->stream, c->lRef, view(c), n, c->s_i32[0].p,
                       c->s_i32[1].p, onlyFindIdentical, c->s_u8[0].p);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->s_u8[0].p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_pass_branch_batch(maple_ctx *c, int32_t n, const int32_t *l, const int32_t *ml, const uint8_t *up,
                                       int32_t *outList)
{
    if (!c || n < 0 || !l || !ml || !up || !outList) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(check_ids(c, n, l, false, "list"));
    const int32_t nml = (int32_t)c->h_mut_cnt.size();
    std::vector<int64_t> woff(n), aoff(n);
    int64_t tot = 0;
    for (int i = 0; i < n; i++) {
        if (ml[i] < 0 || ml[i] >= nml) return fail(c, MAPLE_ERR_ARG, "mutList[%d] = %d is not a mutation-list id", i, ml[i]);
        woff[i] = tot; aoff[i] = 5 * tot;
        tot += (int64_t)c->h_n_ent[l[i]] + 2 * (int64_t)c->h_mut_cnt[ml[i]];
    }
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    TRY(stage_begin(c, (size_t)n * 64 + 256));
    STAGE(dl, c, l, n); STAGE(dml, c, ml, n); STAGE(dup, c, up, n); STAGE(dwo, c, woff.data(), n); STAGE(dao, c, aoff.data(), n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_i32[2].reserve(n));
    HIPCK(c, c->s_i32[3].reserve(n));
    OutSpec o{c->s_words.p, c->s_aux.p, dwo, dao, c->s_i32[2].p, c->s_i32[3].p};
    hipLaunchKernelGGL(k_pass, dim3(grid_for(n)), dim3(MAPLE_BLOCK), 0, c->stream, c->lRef, view(c), mview(c), n, dl, dml, dup, o);
    HIPCK(c, hipGetLastError());
    return commit_lists(c, n, dwo, dao, c->s_i32[2].p, c->s_i32[3].p, outList);
}

extern "C" int maple_shorten_batch(maple_ctx *c, int32_t n, const int32_t *l, int32_t *outList)
{
    if (!c || n < 0 || !l || !outList) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, l, false, "list"));
    std::vector<int64_t> woff(n), aoff(n);
    int64_t tot = 0;
    for (int i = 0; i < n; i++) { woff[i] = tot; aoff[i] = 5 * tot; tot += c->h_n_ent[l[i]]; }
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    TRY(stage_begin(c, (size_t)n * 56 + 256));
    STAGE(dl, c, l, n); STAGE(dwo, c, woff.data(), n); STAGE(dao, c, aoff.data(), n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_i32[2].reserve(n));
    HIPCK(c, c->s_i32[3].reserve(n));
    OutSpec o{c->s_words.p, c->s_aux.p, dwo, dao, c->s_i32[2].p, c->s_i32[3].p};
    if (n <= wave_item_max(c, 1024))
        DISPATCH3(c, k_shorten_wave, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dl, o));
    else
        DISPATCH3(c, k_shorten, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dl, o));
    HIPCK(c, hipGetLastError());
    return commit_lists(c, n, dwo, dao, c->s_i32[2].p, c->s_i32[3].p, outList);
}

extern "C" int maple_root_vector_batch(maple_ctx *c, int32_t n, const int32_t *l, const double *bl, const uint8_t *tip,
                                       const int64_t *pathOff, const int32_t *pathMut, int32_t *outList)
{
    if (!c || n < 0 || !l || !bl || !tip || !pathOff || !outList) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, l, false, "list"));
    const int32_t nml = (int32_t)c->h_mut_cnt.size();
    std::vector<int64_t> capOff(n + 1), woff(n), aoff(n);
    int64_t tot = 0;
    for (int i = 0; i < n; i++) {
        int64_t cap = c->h_n_ent[l[i]];
        for (int64_t k = pathOff[i]; k < pathOff[i + 1]; k++) {
            if (pathMut[k] >= nml) return fail(c, MAPLE_ERR_ARG, "pathMutLists[%lld] is not a mutation-list id", (long long)k);
            if (pathMut[k] >= 0) cap += 4 * (int64_t)c->h_mut_cnt[pathMut[k]];
        }
        capOff[i] = tot; woff[i] = 3 * tot; aoff[i] = 15 * tot;
        tot += cap;
    }
    capOff[n] = tot;
    HIPCK(c, c->s_words.reserve((size_t)(3 * tot)));
    HIPCK(c, c->s_aux.reserve((size_t)(15 * tot)));
    const int64_t np = pathOff[n];
    TRY(stage_begin(c, (size_t)n * 96 + (size_t)np * 4 + 512));
    STAGE(dl, c, l, n); STAGE(dbl, c, bl, n); STAGE(dtip, c, tip, n); STAGE(dwo, c, woff.data(), n); STAGE(dao, c, aoff.data(), n);
    STAGE(dpo, c, pathOff, n + 1); STAGE(dpm, c, pathMut, np); STAGE(dco, c, capOff.data(), n + 1);
    TRY(stage_flush(c));
    HIPCK(c, c->s_i32[2].reserve(n));
    HIPCK(c, c->s_i32[3].reserve(n));
    OutSpec o{c->s_words.p, c->s_aux.p, dwo, dao, c->s_i32[2].p, c->s_i32[3].p};
    DISPATCH3(c, k_root_vector, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), mview(c), n, dl, dbl, dtip, dpo, dpm,
                                                                               dco, o));
    HIPCK(c, hipGetLastError());
    return commit_lists(c, n, dwo, dao, c->s_i32[2].p, c->s_i32[3].p, outList);
}

// comp2 (optional, 2 doubles per item): see k_evalplace
static int evaluate_placement_items(maple_ctx *c, int32_t n, const int32_t *midTot, const int32_t *down, const int32_t *up,
                                    const double *dist, const int32_t *rem, const uint8_t *remTip, const uint8_t *fromTip1,
                                    double *out4, double *comp2)
{
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, midTot, false, "midTot"));
    TRY(check_ids(c, n, down, false, "downVect"));
    TRY(check_ids(c, n, up, false, "upVect"));
    TRY(check_ids(c, n, rem, false, "removedPartials"));
    std::vector<int64_t> capOff(n), aisOff(n);
    int64_t tot = 0, totA = 0;
    for (int i = 0; i < n; i++) {
        int64_t nd = c->h_n_ent[down[i]], nu = c->h_n_ent[up[i]], nr = c->h_n_ent[rem[i]], nm = c->h_n_ent[midTot[i]];
        capOff[i] = tot;
        tot += (nd + nr) + (nu + nr) + (nu + nd);
        aisOff[i] = totA;
        int64_t mx = nm + nr;
        if (nu + nd + nr > mx) mx = nu + nd + nr;
        if (nu + nr + nd > mx) mx = nu + nr + nd;
        totA += mx;
    }
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    HIPCK(c, c->s_ais.reserve((size_t)totA));
    TRY(stage_begin(c, (size_t)n * 64 + 1024));
    STAGE(dMid, c, midTot, n); STAGE(dDown, c, down, n); STAGE(dUp, c, up, n); STAGE(dRem, c, rem, n); STAGE(dDist, c, dist, n);
    STAGE(dRt, c, remTip, n); STAGE(dFt, c, fromTip1, n); STAGE(dCap, c, capOff.data(), n); STAGE(dAis, c, aisOff.data(), n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_f64[1].reserve((size_t)6 * n));
    HIPCK(c, c->s_i32[4].reserve(n));
    double *d4 = c->s_f64[1].p, *d2 = comp2 ? d4 + (size_t)4 * n : nullptr;
    // a handful of items (a single query's short list) wait for ONE item's chain of walks: a wavefront per item
    if (n <= wave_item_max(c, 2048))
        DISPATCH3(c, k_evalplace_wave, <<<n, 64, 0, c->stream>>>(c->d_model, view(c), n, dMid, dDown, dUp, dDist, dRem, dRt, dFt, c->s_words.p,
                                                                  c->s_aux.p, dCap, c->s_ais.p, dAis, d4, c->s_i32[4].p, d2));
    else
        DISPATCH3(c, k_evalplace, <<<grid_for(n), MAPLE_BLOCK, 0, c->stream>>>(c->d_model, view(c), n, dMid, dDown, dUp, dDist, dRem, dRt, dFt,
                                                                                 c->s_words.p, c->s_aux.p, dCap, c->s_ais.p, dAis, d4,
                                                                                 c->s_i32[4].p, d2));
    HIPCK(c, hipGetLastError());
    const size_t nD = (size_t)(comp2 ? 6 : 4) * n;
    HIPCK(c, c->pin_res.reserve(nD * sizeof(double) + (size_t)n * sizeof(int32_t)));
    double *hD = (double *)c->pin_res.p;
    int32_t *st = (int32_t *)(hD + nD);
    HIPCK(c, hipMemcpyAsync(hD, d4, nD * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(st, c->s_i32[4].p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    memcpy(out4, hD, (size_t)4 * n * sizeof(double));
    if (comp2) memcpy(comp2, hD + (size_t)4 * n, (size_t)2 * n * sizeof(double));
    for (int i = 0; i < n; i++)
        if (st[i]) return fail(c, MAPLE_ERR_FATAL, "evaluatePlacement item %d: a merge returned None", i);
    return MAPLE_OK;
}

extern "C" int maple_evaluate_placement_batch(maple_ctx *c, int32_t n, const int32_t *midTot, const int32_t *down,
                                              const int32_t *up, const double *dist, const int32_t *rem,
                                              const uint8_t *remTip, const uint8_t *fromTip1, double *out4)
{
    if (!c || n < 0 || !midTot || !down || !up || !dist || !rem || !remTip || !fromTip1 || !out4) return MAPLE_ERR_ARG;
    return evaluate_placement_items(c, n, midTot, down, up, dist, rem, remTip, fromTip1, out4, nullptr);
}

// ---- device-resident forms ---------------------------------------------------------------------------
int maple_internal_ev_pair(maple_ctx *c, hipEvent_t *a, hipEvent_t *b, int kind, double units, double bytes)
{
    if (c->ev_used >= 65536) c->ev_used = 0;      // nobody is reading these timings: recycle the event pairs
    const size_t slot = c->ev_used / 2;
    if (c->ev_kind.size() <= slot) { c->ev_kind.resize(slot + 1); c->ev_units.resize(slot + 1); c->ev_bytes.resize(slot + 1); }
    c->ev_kind[slot] = kind; c->ev_units[slot] = units; c->ev_bytes[slot] = bytes;
    if (c->ev_used + 2 > c->evs.size()) {
        hipEvent_t e0, e1;
        HIPCK(c, hipEventCreate(&e0));
        HIPCK(c, hipEventCreate(&e1));
        c->evs.push_back(e0);
        c->evs.push_back(e1);
    }
    *a = c->evs[c->ev_used];
    *b = c->evs[c->ev_used + 1];
    c->ev_used += 2;
    return MAPLE_OK;
}
// one launch of k_append_queries on stream s (timed with an event pair): out[q * ldOut + (outCol ? outCol[k] : k)]
static int launch_append_queries(maple_ctx *c, hipStream_t s, int nQ, const int32_t *qList, int nC, const int32_t *cand,
                                 int isTip, double bLen, double *out, long long ldOut, const int32_t *outCol,
                                 const uint8_t *qTip, const double *qBLen, int kind, double algBytes, TileBest *tileBest = nullptr,
                                 const int32_t *visitRank = nullptr, const int4 *chunkTab = nullptr, int nChunkTab = 0, int nF = 1,
                                 unsigned long long *finMask = nullptr)
{
    const long long tiles = (long long)nQ * (chunkTab ? nChunkTab : (nC + 63) / 64);
    if (tiles > 0x7fffffffLL - (1 << 20)) return fail(c, MAPLE_ERR_ARG, "nQ x nC too large for one launch");
    if (!c->d_tile_counters) HIPCK(c, hipMalloc((void **)&c->d_tile_counters, 64 * sizeof(int32_t)));
    int32_t *counter = c->d_tile_counters + (c->tile_counter_next++ & 63);
    HIPCK(c, hipMemsetAsync(counter, 0, sizeof(int32_t), s));
    const long long waves = (tiles + 3) / 4;
    const int grid = waves < 256 * MAPLE_APPEND_WAVES ? (int)waves : 256 * MAPLE_APPEND_WAVES;   // workgroups of 4 wavefronts, MAPLE_APPEND_WAVES per CU = the occupancy limit
    hipEvent_t e0, e1;
    TRY(ev_pair(c, &e0, &e1, kind, (double)nQ * (double)nC, algBytes));
    HIPCK(c, hipEventRecord(e0, s));
    if (chunkTab || nQ >= 32) {
        // enough queries to reuse a staged candidate chunk: the LDS kernel, one workgroup of 16 wavefronts per CU
        const long long units = (long long)(chunkTab ? nChunkTab : (nC + 63) / 64) * ((nQ + MAPLE_LDS_QB - 1) / MAPLE_LDS_QB);
        const int gridL = units < 256 ? (int)units : 256;
        const bool rv_ = c->dm.useRateVariation;
        const size_t dyn = rv_ ? lds_kernel_dyn_bytes<true>() : lds_kernel_dyn_bytes<false>();
        static bool attrSet = false;
        if (!attrSet) {                                                // more than 64 KB of LDS per workgroup has to be asked for
#define MAPLE_SET_LDS(RV_, U_, SS_) HIPCK(c, hipFuncSetAttribute((const void *)k_append_queries_lds<RV_, U_, SS_>, \
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kernel_dyn_bytes<RV_>()))
            MAPLE_SET_LDS(false, false, false); MAPLE_SET_LDS(true, false, false); MAPLE_SET_LDS(false, true, false);
            MAPLE_SET_LDS(false, true, true); MAPLE_SET_LDS(true, true, false); MAPLE_SET_LDS(true, true, true);
#undef MAPLE_SET_LDS
            attrSet = true;
        }
        DISPATCH3(c, k_append_queries_lds, <<<gridL, MAPLE_LDS_BLOCK, dyn, s>>>(c->d_model, view(c), nQ, qList, nC, cand, isTip, bLen, out,
                                                                              ldOut, outCol, qTip, qBLen, counter, tileBest, visitRank,
                                                                              chunkTab, nChunkTab, nF, finMask));
    } else
    DISPATCH3(c, k_append_queries, <<<grid, MAPLE_BLOCK, 0, s>>>(c->d_model, view(c), nQ, qList, nC, cand, isTip, bLen, out, ldOut,
                                                                  outCol, qTip, qBLen, counter, tileBest, visitRank, finMask));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(e1, s));
    return MAPLE_OK;
}

extern "C" int maple_append_batch_dev(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl, const uint8_t *tip,
                                      const double *bl, double *out, void *stream)
{
    if (!c || n < 0 || !pl || !cl || !tip || !bl || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(settle(c));
    hipStream_t s = (hipStream_t)stream;                               // the caller's stream, verbatim (NULL = the legacy default stream)
    hipEvent_t e0, e1;
    TRY(ev_pair(c, &e0, &e1, MAPLE_K_APPEND_PAIRS, (double)n, 0.0));
    HIPCK(c, hipEventRecord(e0, s));
    DISPATCH3(c, k_append, <<<grid_for(n), MAPLE_BLOCK, 0, s>>>(c->d_model, view(c), n, pl, cl, tip, bl, out));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(e1, s));
    return MAPLE_OK;
}

extern "C" int maple_append_queries_dev(maple_ctx *c, int32_t nQ, const int32_t *qList_dev, int32_t nC,
                                        const int32_t *cand_dev, int isTipC, double bLen, double *out_dev, void *stream)
{
    if (!c || nQ < 0 || nC < 0 || !qList_dev || !cand_dev || !out_dev) return MAPLE_ERR_ARG;
    if (nQ == 0 || nC == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(settle(c));
    return launch_append_queries(c, (hipStream_t)stream, nQ, qList_dev, nC, cand_dev, isTipC, bLen, out_dev,
                                 nC, nullptr, nullptr, nullptr, MAPLE_K_APPEND_QUERIES, 0.0);
}

// Q queries x C candidates without the score matrix: per query the best score and the candidate that has it (exact
// ties to the smallest visitRank, or to the smallest index when visitRank is NULL).
extern "C" int maple_append_queries_argmax_dev(maple_ctx *c, int32_t nQ, const int32_t *qList_dev, int32_t nC,
                                               const int32_t *cand_dev, const int32_t *visitRank_dev, int isTipC, double bLen,
                                               double *bestScore_dev, int32_t *bestIdx_dev, void *stream)
{
    if (!c || nQ < 0 || nC < 0 || !qList_dev || !cand_dev || !bestScore_dev || !bestIdx_dev) return MAPLE_ERR_ARG;
    if (nQ == 0 || nC == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(settle(c));
    const int nChunks = (nC + 63) / 64;
    HIPCK(c, c->s_tilebest.reserve((size_t)nQ * nChunks * sizeof(TileBest)));
    TileBest *tb = (TileBest *)c->s_tilebest.p;
    TRY(launch_append_queries(c, (hipStream_t)stream, nQ, qList_dev, nC, cand_dev, isTipC, bLen, nullptr, 0, nullptr, nullptr, nullptr,
                              MAPLE_K_APPEND_QUERIES, 0.0, tb, visitRank_dev));
    hipLaunchKernelGGL(k_argmax_reduce, dim3(nQ), dim3(64), 0, (hipStream_t)stream, nQ, nChunks, tb, bestScore_dev, bestIdx_dev);
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}

// ---- RCCL: the arg-max over the ranks' candidate shards (SURVEY 8b / 8e level 2) -----------------------------------------
// RCCL is loaded at run time (dlopen: the copy the process already has, e.g. PyTorch's, is reused), so the library itself
// links against nothing but the HIP runtime.  The communicator's unique id travels out of band (the host broadcasts the
// 128 bytes, e.g. with torch.distributed).
static int rccl_load(maple_ctx *c)
{
    if (c->rccl_lib) return MAPLE_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(c, MAPLE_ERR_STATE, "RCCL (librccl.so) cannot be loaded: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(c, MAPLE_ERR_STATE, "librccl.so lacks the expected entry points");
    c->rccl_lib = h;
    return MAPLE_OK;
}
#define NCCLCK(c, call)                                                                                     \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
            return fail((c), MAPLE_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
    } while (0)

extern "C" int maple_comm_unique_id(maple_ctx *c, uint8_t *id128)
{
    if (!c || !id128) return MAPLE_ERR_ARG;
    TRY(rccl_load(c));
    ncclUniqueId id;
    NCCLCK(c, g_rccl.GetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return MAPLE_OK;
}

extern "C" int maple_comm_init(maple_ctx *c, int32_t world, int32_t rank, const uint8_t *id128)
{
    if (!c || world < 1 || rank < 0 || rank >= world || !id128) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    TRY(rccl_load(c));
    if (c->rccl_comm) { NCCLCK(c, g_rccl.CommDestroy((ncclComm_t)c->rccl_comm)); c->rccl_comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm;
    NCCLCK(c, g_rccl.CommInitRank(&comm, world, id, rank));
    c->rccl_comm = comm; c->rccl_world = world; c->rccl_rank = rank;
    return MAPLE_OK;
}

// order-preserving image of a double in an unsigned 64-bit integer (so that an integer max IS the float max), and back
__device__ inline unsigned long long f64_key(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ void k_argmax_pack(int n, const double *score, unsigned long long *key)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = f64_key(score[i]);
}
// after the max all-reduce of the keys: ranks that hold the winning score offer their visit index, the others "infinity"
__global__ void k_argmax_offer(int n, const double *score, const unsigned long long *best, const int32_t *idx, unsigned long long *offer)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) offer[i] = (f64_key(score[i]) == best[i]) ? (unsigned long long)(uint32_t)idx[i] : ~0ull;
}
__global__ void k_argmax_unpack(int n, const unsigned long long *best, const unsigned long long *offer, double *score, int32_t *idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned long long k = best[i];
        const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        score[i] = __longlong_as_double((long long)b);
        idx[i] = (int32_t)(uint32_t)offer[i];
    }
}

// In place, on `stream`: score[i] = max over ranks, idx[i] = the SMALLEST idx among the ranks that hold that score (the
// earliest depth-first visit wins an exact tie, like the reference's strict >, M:7083 / 8065).  Two all-reduces of n
// 8-byte words over xGMI (max of the order-preserving keys, then min of the offered indices).
extern "C" int maple_argmax_allreduce_dev(maple_ctx *c, int32_t n, double *score_dev, int32_t *idx_dev, void *stream)
{
    if (!c || n < 0 || !score_dev || !idx_dev) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    if (!c->rccl_comm) return fail(c, MAPLE_ERR_STATE, "maple_comm_init has not been called");
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, c->s_comm_u64.reserve((size_t)2 * n));
    unsigned long long *key = c->s_comm_u64.p, *offer = key + n;
    hipStream_t s = (hipStream_t)stream;
    const int g = (n + 255) / 256;
    hipLaunchKernelGGL(k_argmax_pack, dim3(g), dim3(256), 0, s, n, score_dev, key);
    NCCLCK(c, g_rccl.AllReduce(key, key, (size_t)n, ncclUint64, ncclMax, (ncclComm_t)c->rccl_comm, s));
    hipLaunchKernelGGL(k_argmax_offer, dim3(g), dim3(256), 0, s, n, score_dev, key, idx_dev, offer);
    NCCLCK(c, g_rccl.AllReduce(offer, offer, (size_t)n, ncclUint64, ncclMin, (ncclComm_t)c->rccl_comm, s));
    hipLaunchKernelGGL(k_argmax_unpack, dim3(g), dim3(256), 0, s, n, key, offer, score_dev, idx_dev);
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}

static int tree_rebuild_from_host(maple_ctx *c);
#include "placement_host.h"
#include "update_host.h"
#include "rebuild_host.h"

// MAT reference frames of the uploaded tree: frame 0 is the root's reference, every node whose branch carries mutations
// opens a new one for its clade.  Frames are numbered by nesting depth (a parent frame always has a smaller index).
static int compute_frames(maple_ctx *c)
{
    PlaceMeta &M = *c->place;
    const int32_t n = c->dtree.n, root = c->dtree.root;
    const auto &up = c->h_tree_up;
    const auto &c0 = c->h_tree_c0, &c1 = c->h_tree_c1, &mut = c->h_tree_mut;
    std::vector<int32_t> &order = M.order;
    std::vector<int32_t> depth(n, 0), fdepth;
    order.clear();
    order.reserve(n);
    std::vector<int32_t> st{root};
    M.frameOf.assign(n, -1);
    M.frameNode.assign(1, -1);
    M.frameParent.assign(1, -1);
    fdepth.assign(1, 0);
    M.maxDepth = 0;
    while (!st.empty()) {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        const int32_t pf = up[v] < 0 || v == root ? 0 : M.frameOf[up[v]];
        if (v != root) depth[v] = depth[up[v]] + 1;
        if (depth[v] > M.maxDepth) M.maxDepth = depth[v];
        if (mut[v] >= 0) {
            M.frameOf[v] = (int32_t)M.frameNode.size();
            M.frameNode.push_back(v);
            M.frameParent.push_back(pf);
            fdepth.push_back(fdepth[pf] + 1);
        } else M.frameOf[v] = pf;
        if (c0[v] >= 0) { st.push_back(c0[v]); st.push_back(c1[v]); }
    }
    // renumber frames by nesting depth so that a level is a contiguous range (parents always in earlier levels)
    const int32_t nF = (int32_t)M.frameNode.size();
    std::vector<int32_t> perm(nF), inv(nF);
    for (int i = 0; i < nF; i++) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return fdepth[a] < fdepth[b]; });
    for (int i = 0; i < nF; i++) inv[perm[i]] = i;
    std::vector<int32_t> fn(nF), fp(nF);
    M.levelStart.clear();
    for (int i = 0; i < nF; i++) {
        fn[i] = M.frameNode[perm[i]];
        fp[i] = M.frameParent[perm[i]] < 0 ? -1 : inv[M.frameParent[perm[i]]];
        if (i > 0 && fdepth[perm[i]] != fdepth[perm[i - 1]]) M.levelStart.push_back(i);
    }
    M.levelStart.push_back(nF);
    M.frameNode.swap(fn);
    M.frameParent.swap(fp);
    for (auto &f : M.frameOf) if (f >= 0) f = inv[f];
    M.nF = nF;
    for (auto &f : M.frameOf) if (f < 0) f = 0;                           // nodes not reachable from the root
    return MAPLE_OK;
}

// ---- tree mirror + SPR search ------------------------------------------------------------------------
extern "C" int maple_tree_upload(maple_ctx *c, int32_t n, int32_t root, const int32_t *up, const int32_t *child0,
                                 const int32_t *child1, const double *dist, const uint8_t *isTip, const int32_t *lower,
                                 const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp, const int32_t *mutList)
{
    if (!c || n <= 0 || root < 0 || root >= n || !up || !child0 || !child1 || !dist || !isTip || !lower || !upRight || !upLeft
        || !totUp || !mutList)
        return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    TRY(check_ids(c, n, lower, true, "lower"));
    TRY(check_ids(c, n, upRight, true, "upRight"));
    TRY(check_ids(c, n, upLeft, true, "upLeft"));
    TRY(check_ids(c, n, totUp, true, "totUp"));
    const int32_t nml = (int32_t)c->h_mut_cnt.size();
    for (int i = 0; i < n; i++)
        if (mutList[i] >= nml) return fail(c, MAPLE_ERR_ARG, "mutList[%d] is not a mutation-list id", i);
    // the topology arrays are trusted by everything below (depth-first orders, frames, the kernels): check them here
    if (up[root] >= 0) return fail(c, MAPLE_ERR_ARG, "the root (%d) has a parent", root);
    for (int i = 0; i < n; i++) {
        if (up[i] < -1 || up[i] >= n || child0[i] < -1 || child0[i] >= n || child1[i] < -1 || child1[i] >= n)
            return fail(c, MAPLE_ERR_ARG, "node %d: up / child index out of range", i);
        if ((child0[i] >= 0) != (child1[i] >= 0)) return fail(c, MAPLE_ERR_ARG, "node %d has exactly one child", i);
        if (child0[i] >= 0 && (child0[i] == child1[i] || child0[i] == i || child1[i] == i))
            return fail(c, MAPLE_ERR_ARG, "node %d: malformed children", i);
    }
    {   // every node reachable from the root must be the child its parent says it is, and be reached once (no cycles)
        std::vector<uint8_t> seen((size_t)n, 0);
        std::vector<int32_t> st{root};
        while (!st.empty()) {
            const int32_t v = st.back();
            st.pop_back();
            if (seen[v]) return fail(c, MAPLE_ERR_ARG, "node %d is reached twice from the root (cycle or shared child)", v);
            seen[v] = 1;
            if (child0[v] >= 0) {
                if (up[child0[v]] != v || up[child1[v]] != v) return fail(c, MAPLE_ERR_ARG, "children of node %d do not point back to it", v);
                st.push_back(child0[v]); st.push_back(child1[v]);
            }
        }
    }
    const int32_t *src[9] = {up, child0, child1, lower, upRight, upLeft, totUp, mutList, nullptr};
    for (int k = 0; k < 8; k++) TRY(h2d(c, c->t_i32[k], src[k], (size_t)n));
    TRY(h2d(c, c->t_dist, dist, (size_t)n));
    TRY(h2d(c, c->t_tip, isTip, (size_t)n));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->h_tree_up.assign(up, up + n);
    c->h_tree_lower.assign(lower, lower + n);
    c->h_tree_dist.assign(dist, dist + n);
    c->h_tree_tip.assign(isTip, isTip + n);
    c->h_tree_c0.assign(child0, child0 + n);
    c->h_tree_c1.assign(child1, child1 + n);
    c->h_tree_mut.assign(mutList, mutList + n);
    c->h_tree_totUp.assign(totUp, totUp + n);
    c->h_tree_upRight.assign(upRight, upRight + n);
    c->h_tree_upLeft.assign(upLeft, upLeft + n);
    if (!c->place) c->place = new PlaceMeta();
    c->place->valid = false;
    c->place->rootVect = -1;
    c->dtree.n = n; c->dtree.root = root;
    TRY(compute_frames(c));
    const PlaceMeta &F = *c->place;
    std::vector<NodeRec> recs((size_t)n);
    for (int i = 0; i < n; i++) {
        NodeRec &r = recs[i];
        memset(&r, 0, sizeof r);
        r.up = up[i]; r.c0 = child0[i]; r.c1 = child1[i];
        r.lower = lower[i]; r.upRight = upRight[i]; r.upLeft = upLeft[i]; r.totUp = totUp[i];
        r.mutId = mutList[i]; r.dist = dist[i]; r.isTip = isTip[i];
        r.upIsRoot = (up[i] >= 0 && up[up[i]] < 0) ? 1 : 0;
        r.whichChild = (up[i] >= 0 && child1[up[i]] == i) ? 1 : 0;
        r.preRank = i;
        r.frameOf = F.frameOf[i];
        r.c0Frame = child0[i] >= 0 ? F.frameOf[child0[i]] : r.frameOf;
        r.c1Frame = child1[i] >= 0 ? F.frameOf[child1[i]] : r.frameOf;
        r.upFrame = up[i] >= 0 ? F.frameOf[up[i]] : r.frameOf;
    }
    {   // depth-first ranks in the order the searches descend (the child pushed last, child 1, is visited first)
        std::vector<int32_t> st;
        std::vector<uint8_t> seen((size_t)n, 0);
        int32_t next = 0;
        if (root >= 0 && root < n) st.push_back(root);
        while (!st.empty()) {
            const int32_t v = st.back();
            st.pop_back();
            if (v < 0 || v >= n || seen[v]) continue;
            seen[v] = 1;
            recs[v].preRank = next++;
            if (child0[v] >= 0) { st.push_back(child0[v]); st.push_back(child1[v]); }
        }
        for (int i = 0; i < n; i++) if (!seen[i]) recs[i].preRank = next++;   // nodes not reachable from the root
        std::vector<int32_t> byRank((size_t)n, 0);
        for (int i = 0; i < n; i++) byRank[recs[i].preRank] = i;
        c->h_depth.assign((size_t)n, 0);
        c->h_clade.clear();
        for (int r = 0; r < n; r++) {                                          // parents precede their children in rank order
            const int v = byRank[r], u = up[v];
            if (u >= 0 && seen[v] && recs[u].preRank < r) c->h_depth[v] = c->h_depth[u] + 1;
        }
    }
    HIPCK(c, c->t_nodes.reserve(((size_t)n + (size_t)n / 8 + 1024) * sizeof(NodeRec) + 64));   // (room for the nodes patches add)
    uint8_t *aligned = (uint8_t *)(((uintptr_t)c->t_nodes.p + 63) & ~(uintptr_t)63);
    HIPCK(c, hipMemcpy(aligned, recs.data(), (size_t)n * sizeof(NodeRec), hipMemcpyHostToDevice));
    c->h_nodes = recs;
    c->nodes_current = true;
    DevTree &T = c->dtree;
    T.n = n; T.root = root;
    T.nd = (const NodeRec *)aligned;
    T.totUp = c->t_i32[6].p;
    c->scan_valid = false;
    c->cand_root_end = -1;
    T.scan = nullptr; T.scanParent = nullptr; T.scanDepthCap = 0;
    c->tree_has_mut = false;
    c->tree_max_ent = 0;
    for (int i = 0; i < n; i++) {
        if (mutList[i] >= 0) c->tree_has_mut = true;
        for (const int32_t *col : {lower, upRight, upLeft, totUp})
            if (col[i] >= 0) c->tree_max_ent = std::max(c->tree_max_ent, c->h_n_ent[col[i]]);
    }
    {   // The dense scoring of the whole-tree searches takes its candidates in the searches' own depth-first order: the 64
        // scores of a tile then land next to each other in the search's row of the score table (a contiguous 512-byte
        // store instead of 64 partial-line stores, which WRITE_SIZE counts 4x), and neighbours in the tree have lists of
        // similar length anyway.  Measured at 100 000 tips: 1 017 -> 940 ms per round against candidates sorted by length.
        std::vector<int32_t> col;
        for (int i = 0; i < n; i++) if (totUp[i] >= 0) col.push_back(i);
        if (c->tree_has_mut)                                          // by reference frame, then depth-first: a chunk of 64
            std::stable_sort(col.begin(), col.end(), [&](int a, int b) {     // candidates shares ONE copy of the query
                return recs[a].frameOf != recs[b].frameOf ? recs[a].frameOf < recs[b].frameOf : recs[a].preRank < recs[b].preRank; });
        else
            std::stable_sort(col.begin(), col.end(), [&](int a, int b) { return recs[a].preRank < recs[b].preRank; });
        {   // (in rank order whatever the tree: what the rows that come with bitmaps are indexed by, FiniteRows)
            std::vector<int32_t> byRank(col);
            if (c->tree_has_mut) std::stable_sort(byRank.begin(), byRank.end(), [&](int a, int b) { return recs[a].preRank < recs[b].preRank; });
            c->h_cand_ids.resize(byRank.size()); c->h_cand_rank.resize(byRank.size()); c->h_cand_frame.resize(byRank.size());
            for (size_t i = 0; i < byRank.size(); i++) {
                c->h_cand_ids[i] = totUp[byRank[i]]; c->h_cand_rank[i] = recs[byRank[i]].preRank; c->h_cand_frame[i] = recs[byRank[i]].frameOf;
            }
            TRY(h2d(c, c->t_cand_rank, c->h_cand_rank.data(), c->h_cand_rank.size()));
        }
        std::vector<int32_t> ids(col.size()), rank(col.size()), fr(col.size());
        for (size_t i = 0; i < col.size(); i++) { ids[i] = totUp[col[i]]; rank[i] = recs[col[i]].preRank; fr[i] = recs[col[i]].frameOf; }
        c->n_frame_chunks = 0;
        if (c->tree_has_mut) {
            std::vector<int4> chunks;
            for (size_t i = 0; i < col.size();) {
                size_t j = i;
                while (j < col.size() && j - i < 64 && fr[j] == fr[i]) j++;
                chunks.push_back(make_int4((int)i, (int)(j - i), fr[i], 0));
                i = j;
            }
            TRY(h2d(c, c->t_frame_chunks, chunks.data(), chunks.size()));
            c->n_frame_chunks = (int32_t)chunks.size();
        }
        c->n_scored = (int32_t)col.size();
        c->scored_bytes_total = 0.0;                                       // 8 E + 8 A + 8 (result) per candidate
        for (size_t i = 0; i < col.size(); i++) c->scored_bytes_total += 8.0 * c->h_n_ent[ids[i]] + 8.0 * c->h_n_aux[ids[i]] + 8.0;
        TRY(h2d(c, c->t_i32[8], ids.data(), ids.size()));
        TRY(h2d(c, c->t_scored_col, rank.data(), rank.size()));
        TRY(h2d(c, c->t_scored_frame, fr.data(), fr.size()));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    c->tree_set = true;
    c->tree_stale = false;
    return MAPLE_OK;
}

// every device table rebuilt from the host's own copy of the tree (after maple_tree_patch, before a search that needs them)
static int tree_rebuild_from_host(maple_ctx *c)
{
    const std::vector<int32_t> up = c->h_tree_up, c0 = c->h_tree_c0, c1 = c->h_tree_c1, lower = c->h_tree_lower,
                               upRight = c->h_tree_upRight, upLeft = c->h_tree_upLeft, totUp = c->h_tree_totUp, mut = c->h_tree_mut;
    const std::vector<double> dist = c->h_tree_dist;
    const std::vector<uint8_t> tip = c->h_tree_tip;
    return maple_tree_upload(c, (int32_t)up.size(), c->dtree.root, up.data(), c0.data(), c1.data(), dist.data(), tip.data(),
                             lower.data(), upRight.data(), upLeft.data(), totUp.data(), mut.data());
}

// A local change of the uploaded tree -- what placeSampleOnTree (M:8300-8722) and the updatePartials after it leave behind:
// a few nodes with new relatives, branch lengths or list ids, one or two new nodes.  nodes[i] gets the record
// (up, child0, child1, dist, isTip, lower, upRight, upLeft, totUp)[i]; ids >= the old node count are new nodes (all of them
// must be listed; nTotal = the new count).  Mutation lists (MAT reference nodes) and the root do not change this way:
// re-upload the tree for that.  The host copy of the tree and the candidate / leaf columns of the placement search are
// updated in place (a few 4-byte writes); the linearised tables of the batched placement search and of the SPR search are
// only marked stale and are rebuilt from the host copy before their next use -- so the serial phase (one placement, one
// patch, one placement, ...) never pays for the whole tree.
extern "C" int maple_tree_patch(maple_ctx *c, int32_t nTotal, int32_t nTouched, const int32_t *nodes, const int32_t *up,
                                const int32_t *child0, const int32_t *child1, const double *dist, const uint8_t *isTip,
                                const int32_t *lower, const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp)
{
    if (!c || nTouched < 0 || (nTouched && (!nodes || !up || !child0 || !child1 || !dist || !isTip || !lower || !upRight || !upLeft || !totUp)))
        return MAPLE_ERR_ARG;
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    HIPCK(c, hipSetDevice(c->device));
    const int32_t nOld = (int32_t)c->h_tree_up.size(), root = c->dtree.root;
    if (nTotal < nOld) return fail(c, MAPLE_ERR_ARG, "a patch cannot remove nodes (%d < %d)", nTotal, nOld);
    const int32_t nl = (int32_t)c->h_n_ent.size();
    std::vector<uint8_t> seenNew((size_t)(nTotal - nOld), 0);
    for (int i = 0; i < nTouched; i++) {
        const int v = nodes[i];
        if (v < 0 || v >= nTotal) return fail(c, MAPLE_ERR_ARG, "nodes[%d] = %d is not a node", i, v);
        if (v >= nOld) seenNew[v - nOld] = 1;
        if (up[i] < -1 || up[i] >= nTotal || child0[i] < -1 || child0[i] >= nTotal || child1[i] < -1 || child1[i] >= nTotal
            || (child0[i] < 0) != (child1[i] < 0))
            return fail(c, MAPLE_ERR_ARG, "node %d: relative out of range, or only one child", v);
        if ((up[i] < 0) != (v == root)) return fail(c, MAPLE_ERR_ARG, "node %d: the root cannot change in a patch", v);
        for (int32_t id : {lower[i], upRight[i], upLeft[i], totUp[i]})
            if (id < -1 || id >= nl) return fail(c, MAPLE_ERR_ARG, "node %d: %d is not a list id", v, id);
    }
    for (size_t k = 0; k < seenNew.size(); k++)
        if (!seenNew[k]) return fail(c, MAPLE_ERR_ARG, "new node %d is not in the patch", nOld + (int)k);
    {   // relatives must point back at each other in the tree AS PATCHED -- checked before anything changes, so that a
        // malformed patch leaves the library's copy as it was (the traversals trust these columns)
        std::vector<int32_t> at((size_t)nTotal, -1);
        for (int i = 0; i < nTouched; i++) at[nodes[i]] = i;
        auto upOf = [&](int v) { return at[v] >= 0 ? up[at[v]] : c->h_tree_up[v]; };
        auto c0Of = [&](int v) { return at[v] >= 0 ? child0[at[v]] : c->h_tree_c0[v]; };
        auto c1Of = [&](int v) { return at[v] >= 0 ? child1[at[v]] : c->h_tree_c1[v]; };
        for (int i = 0; i < nTouched; i++) {
            const int v = nodes[i];
            for (int32_t ch : {child0[i], child1[i]})
                if (ch >= 0 && (ch == v || upOf(ch) != v)) return fail(c, MAPLE_ERR_ARG, "node %d: child %d does not point back to it", v, ch);
            if (child0[i] >= 0 && child0[i] == child1[i]) return fail(c, MAPLE_ERR_ARG, "node %d: the same child twice", v);
            if (up[i] >= 0 && c0Of(up[i]) != v && c1Of(up[i]) != v)
                return fail(c, MAPLE_ERR_ARG, "node %d is not a child of its parent %d", v, up[i]);
        }
    }
    // ---- the host copy
    for (auto *vec : {&c->h_tree_up, &c->h_tree_c0, &c->h_tree_c1, &c->h_tree_lower, &c->h_tree_upRight, &c->h_tree_upLeft,
                      &c->h_tree_totUp, &c->h_tree_mut})
        vec->resize((size_t)nTotal, -1);
    c->h_tree_dist.resize((size_t)nTotal, 0.0);
    c->h_tree_tip.resize((size_t)nTotal, 0);
    for (int i = 0; i < nTouched; i++) {
        const int v = nodes[i];
        c->h_tree_up[v] = up[i]; c->h_tree_c0[v] = child0[i]; c->h_tree_c1[v] = child1[i];
        c->h_tree_dist[v] = dist[i]; c->h_tree_tip[v] = isTip[i];
        c->h_tree_lower[v] = lower[i]; c->h_tree_upRight[v] = upRight[i]; c->h_tree_upLeft[v] = upLeft[i]; c->h_tree_totUp[v] = totUp[i];
    }
    c->dtree.n = nTotal;
    c->tree_stale = true;
    c->scan_valid = false;
    c->cand_root_end = -1;
    c->h_clade.clear();
    // ---- the node records of the SPR search (search_dev.h): the touched nodes and their relatives are rewritten in place, so
    // that a small batch of searches -- the re-search of a proposed move before it is applied, M:9470-9484 -- can run on the
    // patched tree at once (frontier tier, no tree-sized table); everything tree-sized (depth-first orders, score columns)
    // waits for the rebuild
    if (c->nodes_current && !c->tree_has_mut && c->dtree.nd) {
        const size_t capNodes = (c->t_nodes.cap - 64) / sizeof(NodeRec);
        if ((size_t)nTotal > capNodes) c->nodes_current = false;
        else {
            c->h_nodes.resize((size_t)nTotal);
            std::vector<int32_t> redo;
            for (int i = 0; i < nTouched; i++) {
                const int v = nodes[i];
                redo.push_back(v);
                for (int32_t w : {c->h_tree_up[v], c->h_tree_c0[v], c->h_tree_c1[v]}) if (w >= 0) redo.push_back(w);
            }
            std::sort(redo.begin(), redo.end());
            redo.erase(std::unique(redo.begin(), redo.end()), redo.end());
            NodeRec *dn = const_cast<NodeRec *>(c->dtree.nd);
            for (int32_t v : redo) {
                NodeRec &r = c->h_nodes[v];
                const int32_t keepRank = v < nOld ? r.preRank : 0;
                memset(&r, 0, sizeof r);
                r.up = c->h_tree_up[v]; r.c0 = c->h_tree_c0[v]; r.c1 = c->h_tree_c1[v];
                r.lower = c->h_tree_lower[v]; r.upRight = c->h_tree_upRight[v]; r.upLeft = c->h_tree_upLeft[v]; r.totUp = c->h_tree_totUp[v];
                r.mutId = -1; r.dist = c->h_tree_dist[v]; r.isTip = c->h_tree_tip[v];
                r.upIsRoot = (r.up >= 0 && c->h_tree_up[r.up] < 0) ? 1 : 0;
                r.whichChild = (r.up >= 0 && c->h_tree_c1[r.up] == v) ? 1 : 0;
                r.preRank = keepRank;                                          // (stale: only the tree-sized tables use it)
                HIPCK(c, hipMemcpyAsync(dn + v, &r, sizeof(NodeRec), hipMemcpyHostToDevice, c->stream));
            }
            HIPCK(c, hipStreamSynchronize(c->stream));
        }
    } else c->nodes_current = false;
    PlaceMeta &M = *c->place;
    if (!M.valid) return MAPLE_OK;                                        // nothing of the placement search to keep up to date
    // ---- the placement search's columns
    M.scanStale = true;
    M.frameOf.resize((size_t)nTotal, -1);
    M.h_candIdx.resize((size_t)nTotal, -1);
    M.h_leafIdx.resize((size_t)nTotal, -1);
    for (bool again = true; again;) {                                     // a new node lives in its parent's reference frame
        again = false;
        for (int i = 0; i < nTouched; i++) {
            const int v = nodes[i];
            if (M.frameOf[v] >= 0) continue;
            const int u = c->h_tree_up[v];
            if (u >= 0 && M.frameOf[u] >= 0) { M.frameOf[v] = M.frameOf[u]; again = true; }
        }
    }
    for (int i = 0; i < nTouched; i++)
        if (M.frameOf[nodes[i]] < 0) return fail(c, MAPLE_ERR_ARG, "new node %d is not attached to the tree", nodes[i]);
    auto poke = [&](DevBuf<int32_t> &b, size_t at, int32_t value) -> int {
        if (at >= b.cap) { M.valid = false; return MAPLE_OK; }            // out of room: the next search rebuilds everything
        HIPCK(c, hipMemcpyAsync(b.p + at, &value, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));                       // (`value` is a local)
        return MAPLE_OK;
    };
    for (int i = 0; i < nTouched && M.valid; i++) {
        const int v = nodes[i];
        if (v == root && lower[i] != -1) M.rootVect = -1;                 // (recomputed by the next search if the root's list changed)
        const bool cand = v != root && up[i] >= 0 && dist[i] > M.effNon0 && totUp[i] >= 0;        // M:8049
        int col = M.h_candIdx[v];
        if (col >= 0 && !cand) M.h_candIdx[v] = -1;                       // (the column stays and is scored for nothing)
        else if (col >= 0) { if (M.h_candList[col] != totUp[i]) { M.h_candList[col] = totUp[i]; TRY(poke(M.d_candList, col, totUp[i])); } }
        else if (cand) {
            col = (int)M.cand.size();                                     // the column of the root vector moves up by one
            M.cand.push_back(v);
            M.h_candIdx[v] = col;
            const int32_t rootFrame = M.h_candFrame.back();
            M.h_candList.back() = totUp[i]; M.h_candFrame.back() = M.frameOf[v];
            M.h_candList.push_back(-1); M.h_candFrame.push_back(rootFrame);
            TRY(poke(M.d_candList, col, totUp[i]));
            TRY(poke(M.d_candFrame, col, M.frameOf[v]));
            TRY(poke(M.d_candFrame, col + 1, rootFrame));
            if ((size_t)col + 1 >= M.d_candList.cap) M.valid = false;
        }
        const bool leaf = child0[i] < 0;
        int lc = M.h_leafIdx[v];
        if (lc >= 0 && !leaf) M.h_leafIdx[v] = -1;
        else if (leaf) {
            if (lower[i] < 0) return fail(c, MAPLE_ERR_STATE, "leaf %d has no lower genome list", v);
            if (lc >= 0) { if (M.h_leafList[lc] != lower[i]) { M.h_leafList[lc] = lower[i]; TRY(poke(M.d_leafList, lc, lower[i])); } }
            else {
                lc = (int)M.leaves.size();
                M.leaves.push_back(v);
                M.h_leafIdx[v] = lc;
                M.h_leafList.push_back(lower[i]); M.h_leafFrame.push_back(M.frameOf[v]);
                TRY(poke(M.d_leafList, lc, lower[i]));
                TRY(poke(M.d_leafFrame, lc, M.frameOf[v]));
            }
        }
    }
    return MAPLE_OK;
}

#ifndef MAPLE_WIDE_BUDGET_DEFAULT
#define MAPLE_WIDE_BUDGET_DEFAULT 256
#endif
// ---- the removed lists of a batch of whole-tree searches in every MAT reference frame of one nesting level, on the device --
// R[k * nF + f] = list id of query k's removed list expressed in frame f (-1: not yet).  One item per (query, frame of the
// level): the list in the parent frame goes down through the mutations of the frame's node (passGenomeListThroughBranch,
// M:7119 / 7342).  Sizes, scratch offsets, arena offsets and the rows of the list table are all produced here (two prefix
// sums per level); the host only learns the totals.
__global__ __launch_bounds__(MAPLE_BLOCK) void k_fan_cap(long long nItems, int nF, int a, int nL, const int32_t *R,
                                                         const int32_t *frameParent, const int32_t *frameMut, ArenaView av,
                                                         MutView mv, long long *cap)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nItems; t += (long long)gridDim.x * blockDim.x) {
        const long long k = t / nL;
        const int f = a + (int)(t - k * nL);
        long long cp = 0;
        if (R[k * nF + f] < 0) {
            const int src = R[k * nF + frameParent[f]];
            if (src >= 0) cp = (long long)av.n_ent[src] + 2ll * mv.cnt[frameMut[f]];
        }
        cap[t] = cp;
    }
}
__global__ __launch_bounds__(MAPLE_BLOCK) void k_fan_pass(int lRef, ArenaView av, MutView mv, long long nItems, int nF, int a, int nL,
                                                          const int32_t *R, const int32_t *frameParent, const int32_t *frameMut,
                                                          const long long *cap, const long long *woff, uint2 *words, double *aux,
                                                          long long *ne, long long *na)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nItems; t += (long long)gridDim.x * blockDim.x) {
        long long e = 0, x = 0;
        if (cap[t] > 0) {
            const long long k = t / nL;
            const int f = a + (int)(t - k * nL);
            Writer w;
            w.init(words + woff[t], aux + 5 * woff[t]);
            const int id = frameMut[f];
            e = pass_walk(lRef, list_ref(av, R[k * nF + frameParent[f]]), mv.mut3 + 3 * mv.off[id], mv.cnt[id], false, w);
            x = w.na;
        }
        ne[t] = e; na[t] = x;
    }
}
// one wavefront per item: scratch -> arena, the list's row of the list table, its id into R
__global__ __launch_bounds__(MAPLE_BLOCK) void k_fan_commit(long long nItems, int nF, int a, int nL, int32_t *R, const long long *woff,
                                                            const long long *ne, const long long *na, const long long *de,
                                                            const long long *da, long long baseE, long long baseA, int32_t firstId,
                                                            const uint2 *sw, const double *sa, uint2 *words, double *aux,
                                                            int64_t *t_ent_off, int64_t *t_aux_off, int32_t *t_n_ent, int32_t *t_n_aux)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long t = wave; t < nItems; t += nwaves) {
        const long long e = ne[t], x = na[t];
        if (lane == 0) {
            t_ent_off[firstId + t] = baseE + de[t]; t_aux_off[firstId + t] = baseA + da[t];
            t_n_ent[firstId + t] = (int32_t)e; t_n_aux[firstId + t] = (int32_t)x;
            if (e > 0) { const long long k = t / nL; R[k * nF + a + (int)(t - k * nL)] = firstId + (int32_t)t; }
        }
        const uint2 *s1 = sw + woff[t];
        uint2 *d1 = words + baseE + de[t];
        for (long long i = lane; i < e; i += 64) d1[i] = s1[i];
        const double *s2 = sa + 5 * woff[t];
        double *d2 = aux + baseA + da[t];
        for (long long i = lane; i < x; i += 64) d2[i] = s2[i];
    }
}

static int fan_out_level(maple_ctx *c, int m, int nF, int a, int b, int32_t *dR, const int32_t *dFrameParent, const int32_t *dFrameMut,
                         DevBuf<long long> *buf /* [6] */, DevBuf<uint8_t> &tmp, double *bytesRead)
{
    const int nL = b - a;
    const long long nItems = (long long)m * nL;
    if (nItems <= 0) return MAPLE_OK;
    if (nItems > 0x7fffffffLL) return fail(c, MAPLE_ERR_ARG, "too many (query, frame) items in one level");
    for (int i = 0; i < 6; i++) HIPCK(c, buf[i].reserve((size_t)nItems + 1));
    long long *cap = buf[0].p, *woff = buf[1].p, *ne = buf[2].p, *na = buf[3].p, *de = buf[4].p, *da = buf[5].p;
    const int grid = (int)std::min<long long>((nItems + MAPLE_BLOCK - 1) / MAPLE_BLOCK, 256 * 8);
    hipLaunchKernelGGL(k_fan_cap, dim3(grid), dim3(MAPLE_BLOCK), 0, c->stream, nItems, nF, a, nL, dR, dFrameParent, dFrameMut, view(c),
                       mview(c), cap);
    size_t tb = 0;
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cap, woff, (int)nItems, c->stream));
    HIPCK(c, tmp.reserve(tb + 256));
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cap, woff, (int)nItems, c->stream));
    long long lastOff = 0, lastCap = 0;
    HIPCK(c, hipMemcpyAsync(&lastOff, woff + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&lastCap, cap + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const long long tot = lastOff + lastCap;
    if (tot == 0) return MAPLE_OK;
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    hipLaunchKernelGGL(k_fan_pass, dim3(grid), dim3(MAPLE_BLOCK), 0, c->stream, c->lRef, view(c), mview(c), nItems, nF, a, nL, dR,
                       dFrameParent, dFrameMut, cap, woff, c->s_words.p, c->s_aux.p, ne, na);
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, ne, de, (int)nItems, c->stream));
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, na, da, (int)nItems, c->stream));
    long long tail[4] = {0, 0, 0, 0};
    HIPCK(c, hipMemcpyAsync(&tail[0], de + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&tail[1], ne + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&tail[2], da + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&tail[3], na + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const long long totE = tail[0] + tail[1], totA = tail[2] + tail[3];
    const int64_t first = (int64_t)c->h_n_ent.size();
    if (c->used_ent + totE > c->cap_ent || c->used_aux + totA > c->cap_aux)
        return fail(c, MAPLE_ERR_NOMEM, "arena full while expressing %d removed lists in %d reference frames", m, nL);
    if (first + nItems > c->cap_lists) return fail(c, MAPLE_ERR_NOMEM, "list table full (%lld)", (long long)c->cap_lists);
    const int gridW = (int)std::min<long long>((nItems + MAPLE_BLOCK / 64 - 1) / (MAPLE_BLOCK / 64), 256 * 8);
    hipLaunchKernelGGL(k_fan_commit, dim3(gridW), dim3(MAPLE_BLOCK), 0, c->stream, nItems, nF, a, nL, dR, woff, ne, na, de, da,
                       (long long)c->used_ent, (long long)c->used_aux, (int32_t)first, c->s_words.p, c->s_aux.p, c->d_words, c->d_aux,
                       c->d_ent_off, c->d_aux_off, c->d_n_ent, c->d_n_aux);
    HIPCK(c, hipGetLastError());
    // the host's copy of the new rows
    c->h_ent_off.resize((size_t)(first + nItems)); c->h_aux_off.resize((size_t)(first + nItems));
    c->h_n_ent.resize((size_t)(first + nItems)); c->h_n_aux.resize((size_t)(first + nItems));
    HIPCK(c, hipMemcpyAsync(c->h_ent_off.data() + first, c->d_ent_off + first, (size_t)nItems * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(c->h_aux_off.data() + first, c->d_aux_off + first, (size_t)nItems * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(c->h_n_ent.data() + first, c->d_n_ent + first, (size_t)nItems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(c->h_n_aux.data() + first, c->d_n_aux + first, (size_t)nItems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->used_ent += totE;
    c->used_aux += totA;
    *bytesRead += 8.0 * (double)totE + 8.0 * (double)totA;
    return MAPLE_OK;
}

// The tree in the searches' own depth-first order (SScan, search_dev.h) -- clade sizes, depths and the per-node facts the
// cached-regime descent tests -- plus what rows with a bitmap of their finite scores need (FiniteRows).  Per uploaded tree and
// effectivelyNon0BLen.
static int build_scan_tables(maple_ctx *c, const SearchParams &P)
{

    // the tree in the searches' own depth-first order (SScan, search_dev.h): clade sizes, depths and the per-node facts
    // the cached-regime descent tests
    const int32_t nT = c->dtree.n;
    std::vector<int32_t> byRank(nT, -1);
    for (int i = 0; i < nT; i++) byRank[c->h_nodes[i].preRank] = i;
    std::vector<SScan> sc((size_t)nT);
    std::vector<int32_t> size(nT, 1), depth(nT, 0);
    // Node slots the root does not reach (a tree read from a file keeps the slots of collapsed nodes, their `up` still
    // naming a live node) rank behind every reachable node and belong to no clade: counted into their stale parent's
    // clade they made the scan of that parent -- and of every ancestor -- run past the clade's end.
    std::vector<uint8_t> reach(nT, 0);
    {
        std::vector<int32_t> stk{c->dtree.root};
        while (!stk.empty()) {
            const int v = stk.back();
            stk.pop_back();
            reach[v] = 1;
            if (c->h_tree_c0[v] >= 0) { stk.push_back(c->h_tree_c0[v]); stk.push_back(c->h_tree_c1[v]); }
        }
    }
    int32_t maxDepth = 0;
    for (int r = 0; r < nT; r++) {                                      // parents precede their clades in rank order
        const int v = byRank[r];
        const int u = c->h_tree_up[v];
        if (reach[v] && u >= 0 && v != c->dtree.root && c->h_nodes[u].preRank < r) depth[v] = depth[u] + 1;
        maxDepth = std::max(maxDepth, depth[v]);
    }
    for (int r = nT - 1; r >= 0; r--) {
        const int v = byRank[r];
        const int u = c->h_tree_up[v];
        if (reach[v] && u >= 0 && v != c->dtree.root && c->h_nodes[u].preRank < r) size[u] += size[v];
    }
    for (int r = 0; r < nT; r++) {
        const int v = byRank[r];
        const NodeRec &nr = c->h_nodes[v];
        uint32_t fl = 0;
        if (nr.up >= 0 && (nr.dist > P.effNon0 || nr.upIsRoot)) fl |= SS_SCORED;
        if (nr.totUp >= 0) fl |= SS_TOTUP;
        if (nr.c0 >= 0) fl |= SS_INNER;
        if (nr.up >= 0 && (nr.whichChild ? c->h_nodes[nr.up].upLeft : c->h_nodes[nr.up].upRight) >= 0) fl |= SS_ENTER;
        sc[r] = SScan{v, size[v], depth[v], ((uint32_t)nr.frameOf << 4) | fl};
    }
    std::vector<int32_t> prank((size_t)nT, 0);
    for (int r = 0; r < nT; r++) { const int u = c->h_tree_up[byRank[r]]; prank[r] = u >= 0 ? c->h_nodes[u].preRank : 0; }
    // for rows that come with a bitmap of their finite scores (FiniteRows, search_dev.h): candidates before each rank, and
    // what a clade adds to the count of candidate placements when it is walked with every score -inf
    std::vector<int32_t> candBefore((size_t)nT + 1, 0), cladeVisits((size_t)nT, 0);
    for (int r = 0; r < nT; r++)
        candBefore[r + 1] = candBefore[r] + ((sc[r].ff & SS_TOTUP) ? 1 : 0);   // (the order of the dense kernel's candidates)
    for (int r = nT - 1; r >= 1; r--) {
        const int v = byRank[r];
        if (!reach[v]) continue;
        const uint32_t fl = sc[r].ff & 15u;
        if (!(fl & SS_ENTER)) continue;                                 // never pushed: neither it nor its clade is visited
        const bool scored = fl & SS_SCORED, counts = scored && (fl & SS_TOTUP), dropped = scored && !(fl & SS_TOTUP);
        const int add = (counts ? 1 : 0) + ((!dropped && (fl & SS_INNER)) ? cladeVisits[r] : 0);
        cladeVisits[prank[r]] += add;
    }
    TRY(h2d(c, c->t_cand_before, candBefore.data(), candBefore.size()));
    TRY(h2d(c, c->t_clade_visits, cladeVisits.data(), cladeVisits.size()));
    TRY(h2d(c, c->t_scan, sc.data(), sc.size()));
    TRY(h2d(c, c->t_scan_parent, prank.data(), prank.size()));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->tree_max_depth = maxDepth;
    c->scan_eff = P.effNon0;
    c->scan_valid = true;
    return MAPLE_OK;
}

// finite scores before each word of every row's bitmap (FiniteRows, search_dev.h): one wavefront per row
__global__ __launch_bounds__(64) void k_finite_prefix(int nRows, int nWords, const unsigned long long *mask, int32_t *prefix)
{
    const int lane = threadIdx.x;
    for (int row = blockIdx.x; row < nRows; row += gridDim.x) {
        const unsigned long long *m = mask + (size_t)row * nWords;
        int32_t *p = prefix + (size_t)row * (nWords + 1);
        int run = 0;
        for (int base = 0; base <= nWords; base += 64) {
            const int w = base + lane;
            const int cnt = w < nWords ? __popcll(m[w]) : 0;
            int incl = cnt;
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
            if (w <= nWords) p[w] = run + incl - cnt;
            run += __shfl(incl, 63, 64);
        }
    }
}

// Trees with MAT local references: lists up the chain of their frames (passGenomeListThroughBranch, one enclosing frame per
// round, batched) until every one is written in the ROOT's frame.  ids / fr: list ids and their frames, both updated in place.
static int lists_to_root_frame(maple_ctx *c, std::vector<int32_t> &ids, std::vector<int32_t> fr)
{
    const PlaceMeta &Fm = *c->place;
    std::vector<int32_t> who, src, ml, out;
    std::vector<uint8_t> dir;
    for (;;) {
        who.clear(); src.clear(); ml.clear();
        for (size_t k = 0; k < ids.size(); k++)
            if (fr[k] != 0 && ids[k] >= 0) { who.push_back((int32_t)k); src.push_back(ids[k]); ml.push_back(c->h_tree_mut[Fm.frameNode[fr[k]]]); }
        if (who.empty()) return MAPLE_OK;
        dir.assign(who.size(), 1);
        out.resize(who.size());
        TRY(maple_pass_branch_batch(c, (int32_t)who.size(), src.data(), ml.data(), dir.data(), out.data()));
        for (size_t i = 0; i < who.size(); i++) { ids[who[i]] = out[i]; fr[who[i]] = Fm.frameParent[fr[who[i]]]; }
    }
}
// ... the candidates' copies: made once per tree, kept in the arena until the tree changes or a release of the caller's takes them
static int ensure_cand_root(maple_ctx *c)
{
    if (c->cand_root_end >= 0) return MAPLE_OK;
    std::vector<int32_t> candRoot(c->h_cand_ids);
    TRY(lists_to_root_frame(c, candRoot, c->h_cand_frame));
    TRY(h2d(c, c->s_cand_root, candRoot.data(), candRoot.size()));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->cand_root_end = (int64_t)c->h_n_ent.size();
    return MAPLE_OK;
}

extern "C" int maple_spr_search_batch(maple_ctx *c, int32_t n, const int32_t *nodes, const maple_search_params *sp,
                                      int32_t ws_entries_per_lane, int32_t *bestNode, double *bestScore, double *blen3,
                                      int32_t *placement, double *improvement, double *currentLK, int32_t *nAppend,
                                      int32_t *status, int32_t *outRprList)
{
    if (!c || n < 0 || !nodes || !sp || !bestNode || !bestScore || !blen3 || !placement || !improvement || !currentLK
        || !nAppend || !status)
        return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    for (int i = 0; i < n; i++)
        if (nodes[i] < 0 || nodes[i] >= c->dtree.n) return fail(c, MAPLE_ERR_ARG, "nodes[%d] = %d is not a node", i, nodes[i]);
    // The tree was patched since its tables were built.  A small batch (the re-search of proposed moves before they are
    // applied) runs on the patched node records alone, through the frontier tier with no hand-over to the dense tier -- unless
    // one of its searches is a whole-tree search by construction (a zero-length branch without an error model); everything
    // else rebuilds the tables first.
    bool patchedOnly = false;
    c->last_search_frontier_only = false;
    if (c->tree_stale) {
        patchedOnly = c->nodes_current && !c->tree_has_mut && n <= 64 && sp->searchTier == 0 && c->trace_query < 0 && !outRprList;
        for (int i = 0; i < n && patchedOnly; i++)
            if (sp->wideSearchBudget >= 0 && !c->dm.usingErrorRate && c->h_tree_dist[nodes[i]] == 0.0) patchedOnly = false;
        if (!patchedOnly) TRY(tree_rebuild_from_host(c));
    }
    const bool dbgT = c->tuning.verbose != 0;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration_cast<std::chrono::microseconds>(b - a).count() * 1e-3;
    };
    const auto tStart = tnow();
    SearchParams P;
    P.strict = sp->strictTopologyStopRules; P.allowedFails = sp->allowedFailsTopology;
    P.thrLKtopology = sp->thresholdLogLKtopology; P.thrPlacement = sp->thresholdTopologyPlacement;
    P.thrOptTopo = sp->thresholdLogLKoptimizationTopology; P.thrConsec = sp->thresholdLogLKconsecutivePlacement;
    P.effNon0 = sp->effectivelyNon0BLen;
    HIPCK(c, c->s_search_out.reserve((size_t)n * sizeof(SearchOut)));
    HIPCK(c, c->s_counter.reserve(8));
    std::vector<SearchOut> &ho = c->h_search_out;
    ho.assign((size_t)n, SearchOut{});
    std::vector<int32_t> todo(nodes, nodes + n), slot(n);
    for (int i = 0; i < n; i++) slot[i] = i;
    // output pool for bestRemovedPartials: a list re-expressed in another frame stays close to its original size
    long long poolCapW = 0, poolCapA = 0;
    uint2 *poolW = nullptr;
    double *poolA = nullptr;
    if (outRprList) {
        for (int i = 0; i < n; i++) {
            int lid = c->h_tree_lower[nodes[i]];
            long long ne = lid >= 0 ? c->h_n_ent[lid] : 0, na = lid >= 0 ? c->h_n_aux[lid] : 0;
            poolCapW += 3 * ne + 64; poolCapA += 3 * na + 5 * ne + 64;
        }
        // its own buffers: the batch operators' scratch is reused by the per-frame passes of the wide searches
        HIPCK(c, c->s_pool_w.reserve((size_t)poolCapW));
        HIPCK(c, c->s_pool_a.reserve((size_t)poolCapA));
        poolW = c->s_pool_w.p; poolA = c->s_pool_a.p;
    }
    HIPCK(c, hipMemsetAsync(c->s_counter.p, 0, 8 * sizeof(int32_t), c->stream));
    unsigned long long *poolUsed = (unsigned long long *)(c->s_counter.p + 2);
    // per-lane list workspace: a search near the root of a tree with long lists (rate variation: many O vectors) merges
    // lists of several hundred entries a few hundred times before it is handed over or done
    const int capW0 = ws_entries_per_lane > 0 ? ws_entries_per_lane : std::max(16384, 64 * c->tree_max_ent);
    // Runs the searches `todo` (results into ho[slot[]]).  Queries whose per-lane workspace overflowed (status -3) are
    // re-run with 8x the workspace, twice at most.  cacheS (optional) = row-major (|todo| x T.n) cached scores.
    bool heavyQueries = false;
    // Lane searches assisted by their wavefront (k_spr_search, wave_dev.h): without an
    // error model -- with one, every search from a zero-length branch also runs to the budget here (no routing hint), the lane
    // tier is then bound by its throughput, not by its longest search, and 24 lanes walking in lockstep do better (100 000
    // tips, full model: 460 ms against 541).
    bool assistOK = !c->dm.usingErrorRate;
    const bool assistFew = !c->dm.usingErrorRate;      // few searching lanes per wavefront, every request served by all 64 lanes
    // (the frontier tier, frontier.hip / frontier_upd.hip; searchTier 1 keeps every search in the one-lane kernels)
    const bool useFrontier = sp->searchTier == 0 && c->trace_query < 0;
    const std::vector<int32_t> *rowOverride = nullptr;                 // rows of the score table the next cached launch reads
    FiniteRows finRows{nullptr, nullptr, 0};                           // ... and, where the rows come with one, the bitmap of their finite scores
    const int finWords = (c->n_scored + 63) / 64;                      // (words per row: one per tile of 64 candidates of the dense kernel)
    auto fin_reserve = [&](size_t rows) -> int {
        HIPCK(c, c->s_fin_mask.reserve_exact(std::max(rows * (size_t)finWords, c->s_fin_mask.cap)));
        HIPCK(c, c->s_fin_prefix.reserve_exact(std::max(rows * (size_t)(finWords + 1), c->s_fin_prefix.cap)));
        return MAPLE_OK;
    };
    auto fin_prefix = [&](hipStream_t st, size_t row0, size_t rows) -> int {   // (after the dense launch that wrote those rows' bitmaps)
        if (!rows) return MAPLE_OK;
        k_finite_prefix<<<(int)std::min<size_t>(rows, 4096), 64, 0, st>>>((int)rows, finWords, c->s_fin_mask.p + row0 * finWords,
                                                                          c->s_fin_prefix.p + row0 * (finWords + 1));
        HIPCK(c, hipGetLastError());
        return MAPLE_OK;
    };
    std::function<int()> afterLaunch;                                  // called once, right after the next search kernel is queued
    auto run_queries = [&](std::vector<int32_t> todo, std::vector<int32_t> slot, const double *cacheS, int budgetNow,
                           const int32_t *rTable, int nF) -> int {
        // the few cached (whole-tree) searches get room up front; more when the budgeted pass already ran out of it
        int capW = cacheS ? (heavyQueries ? 8 : 4) * capW0 : capW0;
        std::vector<int32_t> rows(todo.size());                        // row of each query in the cache / frame tables
        for (size_t k = 0; k < rows.size(); k++) rows[k] = rowOverride ? (*rowOverride)[k] : (int32_t)k;
        if ((int)c->h_depth.size() >= c->dtree.n) {
            // Lanes pull searches from a counter, so a launch ends one search after the last one is pulled: the expensive
            // searches go first.  The expensive ones are those near the root (long lists: an updating step there merges
            // several hundred entries; measured up to 100 ms of updating steps in one search of the 100 000-tip tree
            // against 4 ms on average) -- nodes in order of depth.
            std::vector<int32_t> ord(todo.size());
            for (size_t k = 0; k < ord.size(); k++) ord[k] = (int32_t)k;
            if (cacheS) {
                // ... for whole-tree searches, those that take a large clade out of the tree: what its removal changes reaches far,
                // the search updates lists for hundreds of steps (measured: the 11 searches of the 100 000-tip tree that used to
                // come back for more workspace sit at depths 17-22, with removed lists of ordinary length)
                if ((int)c->h_clade.size() != c->dtree.n) {
                    const int nT = c->dtree.n;
                    c->h_clade.assign((size_t)nT, 1);
                    std::vector<int32_t> order, stk{c->dtree.root};
                    while (!stk.empty()) {
                        const int v = stk.back();
                        stk.pop_back();
                        order.push_back(v);
                        if (c->h_tree_c0[v] >= 0) { stk.push_back(c->h_tree_c0[v]); stk.push_back(c->h_tree_c1[v]); }
                    }
                    for (size_t k = order.size(); k-- > 1;) c->h_clade[c->h_tree_up[order[k]]] += c->h_clade[order[k]];
                }
                std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return c->h_clade[todo[a]] > c->h_clade[todo[b]]; });
            } else
            std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return c->h_depth[todo[a]] < c->h_depth[todo[b]]; });
            std::vector<int32_t> t2(todo.size()), s2(todo.size()), r2(todo.size());
            for (size_t k = 0; k < ord.size(); k++) { t2[k] = todo[ord[k]]; s2[k] = slot[ord[k]]; r2[k] = rows[ord[k]]; }
            todo.swap(t2); slot.swap(s2); rows.swap(r2);
        }
        for (int attempt = 0; attempt < 3 && !todo.empty(); attempt++, capW *= 8) {
            const int m = (int)todo.size();
            WsLayout L;
            L.capW = capW;
            L.capA = 5 * L.capW;                                        // O-vector-heavy lists (rate variation) carry up to 4-5 aux doubles per entry
            L.capH = L.capW / 8 + 256;
            L.capS = 1024 * (attempt + 1);
            L.capB = (cacheS ? 4096 : 1024) * (attempt + 1);             // whole-tree (cached) searches short-list far more branches
            L.capAis = 8192 * (attempt + 1);
            LaneBytes LB = lane_bytes(L);
            // lanes: one query per lane while they last; at most 4 wavefronts per SIMD (the kernel's occupancy) and a
            // workspace footprint bounded to ~96 GB of the 288 GB
            long long wsBudget = 96ll << 30;
            {   // ... and to 70 % of what is free on the device right now (plus what this buffer already holds)
                size_t freeB = 0, totalB = 0;
                if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                    const long long avail = (long long)((double)(freeB + c->s_search_ws.cap) * 0.7);
                    if (avail < wsBudget) wsBudget = avail;
                }
            }
            long long maxLanes = wsBudget / (long long)LB.total;
            if (maxLanes > 4096 * 64) maxLanes = 4096 * 64;
            // lanes pull searches from a counter: more wavefronts than the GPU holds at once (4 per SIMD) only cost workspace
            if (cacheS && maxLanes > 8192) maxLanes = 8192;
            if (maxLanes < 64) maxLanes = 64;
            const int lanesWanted = (int)(m < maxLanes ? m : maxLanes);
            // searching lanes per wavefront (measured at 20k queries: the long searches of a non-strict round like 2 lanes,
            // 42 vs 46 ms with 1; the short ones of a strict round like 1, 31 vs 37 ms with 2; more is always worse)
            // (at 200k queries with the budget of a 100 000-tip tree: 2 / 4 / 8 / 13 / 24 lanes -> 825 / 645 / 512 / 424 / 417 ms)
            const int lanesDiv = P.strict ? 32768 : (lanesWanted > 65536 ? 8192 : 16384);
            int activeLanes = (lanesWanted + lanesDiv - 1) / lanesDiv;
            if (activeLanes < 1) activeLanes = 1;
            if (activeLanes > 64) activeLanes = 64;
            // wave-assisted lane searches (below): the wavefront serves its lanes' score requests one after the other, so few
            // searching lanes per wavefront (100 000 tips, budget 2 132: 2 / 4 / 6 / 8 / 16 / 24 lanes -> 298 / 281 / 279 / 289 /
            // 394 / 345 ms; without the assistance 375)
            if (!cacheS && assistOK && assistFew && activeLanes > 4) activeLanes = 4;
            int nWaves = (lanesWanted + activeLanes - 1) / activeLanes;
            if (nWaves > 8192) nWaves = 8192;
            const int lanes = nWaves * activeLanes;
            const auto tWs0 = tnow();
            {   // grow-only and at least doubling (a 20 GB hipMalloc costs ~0.7 s): batches of slowly growing size must not
                // reallocate every time
                size_t need = (size_t)lanes * LB.total;
                if (need > c->s_search_ws.cap) {
                    const size_t top = (size_t)maxLanes * LB.total;
                    need = std::min(std::max(need, 2 * c->s_search_ws.cap), std::max(top, need));
                }
                HIPCK(c, c->s_search_ws.reserve_exact(need));
            }
            if (dbgT) fprintf(stderr, "[maple]   workspace %d lanes x %zu B: reserve %.1f ms\n", lanes, (size_t)LB.total, tms(tWs0, tnow()));
            HIPCK(c, hipMemsetAsync(c->s_counter.p, 0, sizeof(int32_t), c->stream));
            TRY(h2d(c, c->s_i32[0], todo.data(), (size_t)m));
            if (cacheS) TRY(h2d(c, c->s_i32[1], rows.data(), (size_t)m));
            HIPCK(c, c->s_search_out.reserve((size_t)m * sizeof(SearchOut)));
            SearchOut *dout = (SearchOut *)c->s_search_out.p;
            // cached (whole-tree) searches descend by scanning the tree in their own depth-first order; the per-depth slots
            // of every searching lane live in LDS (deeper trees fall back to popping one node at a time)
            DevTree Tk = c->dtree;
            Tk.candBefore = c->t_cand_before.p;
            Tk.cladeVisits = c->t_clade_visits.p;
            size_t dynLds = 0;
            int launchLanes = activeLanes, launchWaves = nWaves;
            if (cacheS && c->scan_valid && !c->tuning.noCladeScan && (size_t)(c->tree_max_depth + 2) * 16 <= (48u << 10)) {
                Tk.scan = c->t_scan.p;
                Tk.scanParent = c->t_scan_parent.p;

                Tk.scanDepthCap = c->tree_max_depth + 2;
                dynLds = ((size_t)Tk.scanDepthCap * 16 + 15) & ~(size_t)15;
                launchLanes = 1;                                         // one search per wavefront, 64 lanes per clade scan
                launchWaves = lanes;                                     // (the workspace is sized for `lanes` searches at a time)
            } else {
                Tk.scan = nullptr; Tk.scanParent = nullptr; Tk.scanDepthCap = 0;
                if (!cacheS && assistOK) dynLds = sizeof(WaveLds);
            }
            // searches that update lists for hundreds of steps outgrow the per-lane list room; they carry on in chunks (one lane's
            // worth each) of a pool the launch shares instead of coming back for a second launch with 8x the room
            long long ovfChunks = 0;
            if (cacheS && attempt == 0) {
                ovfChunks = std::min<long long>(1024, std::max<long long>(64, m / 16));
                HIPCK(c, c->s_search_ws_big.reserve_exact((size_t)ovfChunks * ((size_t)L.capW * sizeof(uint2) + (size_t)L.capA * sizeof(double))));
                HIPCK(c, hipMemsetAsync(c->s_counter.p + 6, 0, 2 * sizeof(int32_t), c->stream));
            }
            const int coopMaxHost = 8;                                      // (see k_spr_search: requests served one by one)
            hipEvent_t e0, e1;
            TRY(ev_pair(c, &e0, &e1, cacheS ? MAPLE_K_SPR_REPLAY : MAPLE_K_SPR_SEARCH, (double)m, 0.0));
            const size_t slotEv = c->ev_used / 2 - 1;                       // (this launch's timing record: filled in below)
            HIPCK(c, hipEventRecord(e0, c->stream));
#define MAPLE_SPR_LAUNCH_ARGS <<<launchWaves, 64, dynLds, c->stream>>>(c->d_model, view(c), mview(c), Tk, P, m, c->s_i32[0].p,          \
                                                                         L, LB, c->s_search_ws.p, c->s_counter.p, dout, poolW,     \
                                                                         poolA, poolUsed, poolCapW, poolCapA,                      \
                                                                         attempt == 0 ? c->trace_query : -1, c->s_trace_i.p,       \
                                                                         c->s_trace_d.p, 4096, c->s_trace_i.p ? c->s_trace_i.p + 4 * 4096 : nullptr, \
                                                                         launchLanes, cacheS, budgetNow, rTable, nF,               \
                                                                         cacheS ? c->s_i32[1].p : nullptr,                          \
                                                                         assistOK ? 1 + coopMaxHost : 0, (unsigned long long *)(c->s_counter.p + 6),          \
                                                                         ovfChunks ? c->s_search_ws_big.p : nullptr, ovfChunks, cacheS ? finRows : FiniteRows{nullptr, nullptr, 0})
            if (!cacheS && assistOK) DISPATCH3(c, k_spr_search_assisted, MAPLE_SPR_LAUNCH_ARGS);
            else DISPATCH3(c, k_spr_search, MAPLE_SPR_LAUNCH_ARGS);
#undef MAPLE_SPR_LAUNCH_ARGS
            HIPCK(c, hipGetLastError());
            HIPCK(c, hipEventRecord(e1, c->stream));
            if (afterLaunch) {                                             // (work for the side stream, queued behind this launch)
                std::function<int()> f;
                f.swap(afterLaunch);
                TRY(f());
            }
            std::vector<SearchOut> part(m);
            HIPCK(c, hipMemcpyAsync(part.data(), dout, (size_t)m * sizeof(SearchOut), hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
            std::vector<int32_t> todo2, slot2, rows2;
            {   // what this launch did, for maple_timing_read_kind: candidate placements it scored itself (lane searches that
                // finished: each reads a candidate list and writes a score, SURVEY 8d, plus its removed list once) or replayed
                // from the score table (8 bytes each)
                const double meanCand = c->n_scored ? c->scored_bytes_total / c->n_scored : 0.0;
                double units = 0.0, bytes = 0.0;
                for (int k = 0; k < m; k++) {
                    if (part[k].status != 0 && part[k].status != -1) continue;
                    units += part[k].nAppend;
                    const int32_t l = c->h_tree_lower[todo[k]];
                    const double qb = l >= 0 ? 8.0 * c->h_n_ent[l] + 8.0 * c->h_n_aux[l] : 0.0;
                    bytes += cacheS ? 8.0 * part[k].nAppend + qb : meanCand * part[k].nAppend + qb;
                }
                c->ev_units[slotEv] = units; c->ev_bytes[slotEv] = bytes;
            }
            for (int k = 0; k < m; k++) {
                ho[slot[k]] = part[k];
                if (part[k].status == -3 && attempt < 2) {
                    todo2.push_back(todo[k]); slot2.push_back(slot[k]); rows2.push_back(rows[k]);
                    if (dbgT) fprintf(stderr, "[maple]   node %d ran out of workspace (capacity kind %d): position %d of the launch, depth %d, lower list %d entries\n", todo[k], part[k].nAppend, k, c->h_depth[todo[k]], c->h_tree_lower[todo[k]] >= 0 ? c->h_n_ent[c->h_tree_lower[todo[k]]] : -1);
                }
            }
            if (dbgT) fprintf(stderr, "[maple] search launch: %d queries, %zu retried with more workspace\n", m, todo2.size());
            todo.swap(todo2);
            slot.swap(slot2);
            rows.swap(rows2);
            // (the budget stays: a retried search that turns out to be wide still goes to the batch path)
        }
        return MAPLE_OK;
    };
    // Wide searches (the non-strict rounds let ~1 query in 5 walk the whole tree): in the cached regime the score of a
    // branch is a pure function of (query, branch), so those queries are scored against every branch by the batch
    // kernel (k_append_queries) and the state machine then only replays the traversal over the cached scores.
    // With MAT local references the removed list is first expressed in every reference frame (below).
    // A search that scores more branches than the budget is handed to the dense path, which costs it one appendProbNode per
    // branch of the tree -- so the budget that pays grows with the tree: 1/64 of the scored branches, measured best at 10 000
    // tips (256: 91 ms per round; 128: 100, 384: 97) and at 100 000 (2 048: 1.08 s; 256: 2.69, 1 024: 1.13, 4 096: 1.20)
    int wideBudget = sp->wideSearchBudget == 0 ? std::max(MAPLE_WIDE_BUDGET_DEFAULT, std::min(8192, c->n_scored / 64))
                                               : sp->wideSearchBudget;
    if (patchedOnly) wideBudget = -1;                                   // (no tree-sized table is current)
    // (a long search costs the wave-assisted lane tier a tenth of what it cost one lane: twice the budget pays -- 100 000 tips:
    // 2 132 / 3 072 / 4 096 / 6 144 -> 693 / 688 / 668 / 692 ms per round; 10 000 tips: 256 / 384 / 512 -> 75 / 72 / 72)
    // (with an error model too, since the frontier tier: 100 000 tips, budget 2 132 / 3 000 / 4 264 / 8 528 -> 532 / 447 / 420 / 441 ms
    // per round -- the searches between 2 000 and 4 000 items are the ones with the longest removed lists, which the dense kernel
    // walks slowest: 28 116 rows take it 260 ms, 25 785 rows 114)
    // (1 000 000 tips, 8 192 / 16 384: 2.32 / 2.73 s per 131 072 searches -- the pools of the longer searches are reallocated on
    // the way: with an error model the doubling stops at 8 192)
    if (sp->wideSearchBudget == 0) wideBudget = c->dm.usingErrorRate ? std::min(2 * wideBudget, std::max(wideBudget, 8192)) : 2 * wideBudget;
    const bool hybrid = wideBudget > 0;
    if (hybrid && !(c->scan_valid && c->scan_eff == P.effNon0) && !c->tuning.noCladeScan) TRY(build_scan_tables(c, P));
    // rows of the score table come with the bitmap of their finite scores (FiniteRows) when the tables that go with it exist
    bool useFin = hybrid && c->scan_valid && !c->tree_has_mut;         // (trees with local references: only for the rows of the searches
                                                                       // known beforehand, below)
    // Without an error model the whole-tree searches are known before anything runs: they are the ones that start from a
    // zero-length branch (the routing hint in the kernel gives those 16 placements and sends them on).  Their dense scoring
    // needs nothing from the lane searches, so it is launched first, on a side stream, and shares the GPU with them -- the
    // lane launch is latency-bound and spends its second half on a thinning tail.  Rows of the score table: the predicted
    // searches in order, then up to `preSpare` searches that run over their budget unannounced.
    std::vector<int32_t> preIdx, preRowOf;
    int preSpare = 0;
    const int nTpre = c->dtree.n;
    // Trees with MAT local references: appendProbNode does not depend on the frame its two lists are written in (the same sites
    // need work, with the same nucleotides, lengths and rates, in the same order), so the rows of these searches are made in the
    // ROOT's frame: every candidate list and every removed list re-expressed there once per call (passGenomeListThroughBranch up
    // the chain of frames), the witness filter and the pair walks as on a plain tree.
    const bool matPre = c->tree_has_mut;
    int64_t preMark = -1;                                              // (the re-expressed lists live until the results are in)
    std::vector<int32_t> preFrameParent, preFrameNode;
    if (hybrid && !c->dm.usingErrorRate && wideBudget > 16 && (!matPre || (c->scan_valid && c->place && !c->tuning.noCladeScan && !c->tuning.denseWideScoring && !c->tuning.wideOutsideFrontier))) {
        {   // a node on a zero-length branch is searched at all only if its current placement is bad enough (M:9674): the
            // kernel's own test, on the same appendProbNode, for all of them at once
            std::vector<int32_t> zi, pl, cl;
            std::vector<uint8_t> tp;
            for (int i = 0; i < n; i++) {
                const int v = nodes[i], u = c->h_tree_up[v];
                if (c->h_tree_dist[v] != 0.0 || u < 0) continue;
                if (matPre && c->h_tree_mut[v] >= 0) continue;          // (a reference node itself: the frame-by-frame path, below)
                const int32_t vu = c->h_tree_c0[u] == v ? c->h_tree_upRight[u] : c->h_tree_upLeft[u];
                if (vu < 0 || c->h_tree_lower[v] < 0) continue;
                zi.push_back(i); pl.push_back(vu); cl.push_back(c->h_tree_lower[v]); tp.push_back(c->h_tree_tip[v]);
            }
            if (zi.size() >= 64) {
                std::vector<double> bl(zi.size(), 0.0), cur(zi.size());
                const int rc = maple_append_batch(c, (int32_t)zi.size(), pl.data(), cl.data(), tp.data(), bl.data(), cur.data());
                if (rc != MAPLE_OK) return rc;
                for (size_t k = 0; k < zi.size(); k++) if (cur[k] < P.thrPlacement) preIdx.push_back(zi[k]);
            }
        }
        size_t freeB = 0, totalB = 0;
        size_t budgetB = (size_t)4ull << 30;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess)
            budgetB = std::max(budgetB, std::min((freeB + c->s_cache.cap * sizeof(double)) / 2, (size_t)96ull << 30));
        const size_t rowsMax = budgetB / ((size_t)nTpre * sizeof(double));
        if (preIdx.size() < 64 || preIdx.size() > rowsMax) preIdx.clear();
        else {
            preSpare = (int)std::min<size_t>(4096, rowsMax - preIdx.size());
            const int mZ = (int)preIdx.size();
            HIPCK(c, c->s_cache.reserve_exact((size_t)(mZ + preSpare) * nTpre));
            TRY(fin_reserve((size_t)mZ + preSpare));
            if (!c->stream2) {
                HIPCK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                HIPCK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIPCK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            std::vector<int32_t> ql(mZ);
            std::vector<uint8_t> qt(mZ);
            std::vector<double> qb(mZ);
            preRowOf.assign(n, -1);
            double qBytes = 0.0;
            for (int k = 0; k < mZ; k++) {
                const int node = nodes[preIdx[k]];
                ql[k] = c->h_tree_lower[node]; qt[k] = c->h_tree_tip[node]; qb[k] = c->h_tree_dist[node];
                preRowOf[preIdx[k]] = k;
                qBytes += 8.0 * c->h_n_ent[ql[k]] + 8.0 * c->h_n_aux[ql[k]];
                // (the witness filter writes -inf by omission ON THE PROOF that the pair is searched with removedBLen = 0 and no
                // error model -- witness.hip: a row with another length must never get here)
                if (qb[k] != 0.0) return fail(c, MAPLE_ERR_FATAL, "a search with removedBLen %g among the searches of the witness filter", qb[k]);
            }
            if (matPre) {
                const PlaceMeta &Fm = *c->place;
                // (the candidates' copies are made once per tree; the removed lists' copies live for this call)
                TRY(ensure_cand_root(c));
                TRY(maple_arena_mark(c, &preMark));
                std::vector<int32_t> qf(mZ);
                for (int k = 0; k < mZ; k++) qf[k] = Fm.frameOf[nodes[preIdx[k]]];
                TRY(lists_to_root_frame(c, ql, qf));
                preFrameParent = Fm.frameParent;
                preFrameNode = Fm.frameNode;
                useFin = true;
            }
            HIPCK(c, c->z_ql.reserve(mZ)); HIPCK(c, c->z_qt.reserve(mZ)); HIPCK(c, c->z_qb.reserve(mZ));
            HIPCK(c, hipEventRecord(c->ev_fork, c->stream));               // (everything the tree tables wait for)
            HIPCK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            HIPCK(c, hipMemcpyAsync(c->z_ql.p, ql.data(), (size_t)mZ * sizeof(int32_t), hipMemcpyHostToDevice, c->stream2));
            HIPCK(c, hipMemcpyAsync(c->z_qt.p, qt.data(), (size_t)mZ, hipMemcpyHostToDevice, c->stream2));
            HIPCK(c, hipMemcpyAsync(c->z_qb.p, qb.data(), (size_t)mZ * sizeof(double), hipMemcpyHostToDevice, c->stream2));
            HIPCK(c, hipStreamSynchronize(c->stream2));                     // (the three vectors are locals)
            // queued BEHIND the lane launch: a workgroup of the dense kernel wants most of a compute unit's LDS, so it starts
            // where the lane searches have thinned out -- launched first it would hold them off instead (measured: no overlap)
            afterLaunch = [c, mZ, nTpre, qBytes, fin_prefix, useFin, finWords, dbgT, matPre]() -> int {
                // (every one of these searches has removedBLen = 0 and there is no error model: only the pairs the witness
                // filter cannot rule out are walked -- witness.hip)
                if (useFin && !c->tuning.denseWideScoring) {
                    long long pairs = 0;
                    TRY(witness_score(c, c->stream2, mZ, c->z_ql.p, c->z_qt.p, c->z_qb.p, c->n_scored, matPre ? c->s_cand_root.p : c->t_i32[8].p,
                                      matPre ? c->t_cand_rank.p : c->t_scored_col.p,
                                      c->s_cache.p, nTpre, c->s_fin_mask.p, finWords,
                                      c->n_scored ? c->scored_bytes_total / c->n_scored : 0.0, qBytes, &pairs));
                    if (dbgT) fprintf(stderr, "[maple] witness filter: %lld of %lld (search, branch) pairs walked\n", pairs, (long long)mZ * c->n_scored);
                    TRY(fin_prefix(c->stream2, 0, (size_t)mZ));
                    HIPCK(c, hipEventRecord(c->ev_join, c->stream2));
                    return MAPLE_OK;
                }
                TRY(launch_append_queries(c, c->stream2, mZ, c->z_ql.p, c->n_scored, c->t_i32[8].p, 0, 0.0, c->s_cache.p, nTpre,
                                          c->t_scored_col.p, c->z_qt.p, c->z_qb.p, MAPLE_K_SPR_SCORE,
                                          (double)mZ * c->scored_bytes_total + qBytes, nullptr, nullptr, nullptr, 0, 1, useFin ? c->s_fin_mask.p : nullptr));
                if (useFin) TRY(fin_prefix(c->stream2, 0, (size_t)mZ));
                HIPCK(c, hipEventRecord(c->ev_join, c->stream2));
                return MAPLE_OK;
            };
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %d searches from zero-length branches to be scored on the side stream\n", tms(tStart, tnow()), mZ);
        }
    }
    // Frontier tier (frontier.hip): every search of the batch expanded level by level, one lane per (search, branch) item,
    // then replayed exactly -- for trees without MAT local references.  What it hands back (a search that would edit its
    // removed list in place, touches the root while still updating lists, or overflows a pool) runs one lane per search.
    if (useFrontier) {
        if (afterLaunch) { std::function<int()> f; f.swap(afterLaunch); TRY(f()); }   // (the side-stream scoring starts alongside)
        FrontierStats fs;
        // An item of the frontier tier costs about what half a (search, branch) pair costs the dense tier on full walks (1.1e9
        // items/s against 2.3e9 pairs/s), so a search only pays for a row of the whole tree once it has expanded half a tree's
        // worth of items; the searches from zero-length branches (whole-tree searches without an error model) never start here.
        // (With an error model there are no searches known to be whole-tree ones beforehand, a fifth of all searches is long, and
        // half a tree's worth of items each does not fit any pool -- 5e8 items at 100 000 tips and counting: those searches
        // leave at the lane tiers' budget.)
        const int frontierBudget = !hybrid ? 0 : (c->dm.usingErrorRate ? wideBudget
                                                                       : std::max(wideBudget, sp->wideSearchBudget == 0 ? c->n_scored / 2 : 0));
        // the searches scored on the side stream stay in the tier: their updating steps run with everybody else's, their clades
        // in the cached regime are scanned over the rows (k_fr_replay_wide)
        FrontierWide fw{nullptr, nullptr, FiniteRows{nullptr, nullptr, 0}, nullptr, 0};
        if (!preIdx.empty() && !c->tuning.wideOutsideFrontier) {
            fw.rowOf = preRowOf.data(); fw.cacheS = c->s_cache.p; fw.rowsReady = c->ev_join;
            if (useFin) fw.fin = FiniteRows{c->s_fin_mask.p, c->s_fin_prefix.p, finWords};
            if (matPre) {                                               // the frames' nesting, for the clade scans' short lists
                TRY(h2d(c, c->s_frame_parent, preFrameParent.data(), preFrameParent.size()));
                TRY(h2d(c, c->s_frame_node, preFrameNode.data(), preFrameNode.size()));
                fw.frameParent = c->s_frame_parent.p; fw.frameNode = c->s_frame_node.p; fw.nFrames = (int)preFrameParent.size();
            }
        }
        TRY(frontier_search(c, P, n, todo.data(), frontierBudget, (hybrid && wideBudget > MAPLE_ZERO_DIST_BUDGET) ? MAPLE_ZERO_DIST_BUDGET : (1 << 30),
                            ho.data(), poolW, poolA, poolUsed, poolCapW, poolCapA, &fs, 0, fw.rowOf ? &fw : nullptr));
        std::vector<int32_t> todoFb, slotFb;
        for (int i = 0; i < n; i++)
            if (ho[i].status == FR_STATUS_FALLBACK) { todoFb.push_back(todo[i]); slotFb.push_back(i); }
        if (dbgT)
            fprintf(stderr, "[maple] t=%.1f ms: frontier tier done: %d levels, %lld updating + %lld cached items, %lld temporary lists "
                            "(%lld words, %lld aux), %lld refined records, %zu searches handed back%s\n", tms(tStart, tnow()), fs.levels,
                    fs.itemsUpdating, fs.itemsCached, fs.tempLists, fs.tempWords, fs.tempAux, fs.records, todoFb.size(),
                    fs.overflow ? " (a pool overflowed)" : "");
        c->last_search_frontier_only = todoFb.empty() && !hybrid;
        if (!todoFb.empty()) TRY(run_queries(todoFb, slotFb, nullptr, hybrid ? wideBudget : 0, nullptr, 0));
    } else
        TRY(run_queries(todo, slot, nullptr, hybrid ? wideBudget : 0, nullptr, 0));
    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: budgeted pass done\n", tms(tStart, tnow()));
    if (hybrid) {
        std::vector<int32_t> wide;
        for (int i = 0; i < n; i++)
            if (ho[i].status == -5) { wide.push_back(i); if (ho[i].bestNode == -2) heavyQueries = true; }
        if (!preIdx.empty()) {
            // replay what was scored on the side, together with as many unannounced ones as the spare rows take
            const int mZ = (int)preIdx.size();
            std::vector<int32_t> qn, sl, rowsNow, rest, ql2;
            std::vector<uint8_t> qt2;
            std::vector<double> qb2;
            for (int32_t i : wide) {
                const int node = nodes[i];
                // (a tree with local references: what the tier did not finish over these rows goes frame by frame, below -- the
                // one-wavefront-per-search replay wants the removed list in every frame)
                if (matPre) { rest.push_back(i); continue; }
                if (preRowOf[i] >= 0) { qn.push_back(node); sl.push_back(i); rowsNow.push_back(preRowOf[i]); }
                else if ((int)ql2.size() < preSpare) {
                    qn.push_back(node); sl.push_back(i); rowsNow.push_back(mZ + (int)ql2.size());
                    ql2.push_back(c->h_tree_lower[node]); qt2.push_back(c->h_tree_tip[node]); qb2.push_back(c->h_tree_dist[node]);
                } else rest.push_back(i);
            }
            if (!ql2.empty()) {
                const int m2 = (int)ql2.size();
                TRY(h2d(c, c->s_i32[6], ql2.data(), (size_t)m2));
                TRY(h2d(c, c->s_u8[3], qt2.data(), (size_t)m2));
                TRY(h2d(c, c->s_f64[3], qb2.data(), (size_t)m2));
                double qBytes = 0.0;
                for (int k = 0; k < m2; k++) qBytes += 8.0 * c->h_n_ent[ql2[k]] + 8.0 * c->h_n_aux[ql2[k]];
                TRY(launch_append_queries(c, c->stream, m2, c->s_i32[6].p, c->n_scored, c->t_i32[8].p, 0, 0.0,
                                          c->s_cache.p + (size_t)mZ * nTpre, nTpre, c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p,
                                          MAPLE_K_SPR_SCORE, (double)m2 * c->scored_bytes_total + qBytes, nullptr, nullptr, nullptr, 0, 1,
                                          useFin ? c->s_fin_mask.p + (size_t)mZ * finWords : nullptr));
                if (useFin) TRY(fin_prefix(c->stream, (size_t)mZ, (size_t)m2));
            }
            if (afterLaunch) { std::function<int()> f; f.swap(afterLaunch); TRY(f()); }   // (no lane launch took it)
            HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
            if (!qn.empty()) {
                rowOverride = &rowsNow;
                if (useFin) finRows = FiniteRows{c->s_fin_mask.p, c->s_fin_prefix.p, finWords};
                const int rc = run_queries(qn, sl, c->s_cache.p, 0, nullptr, 0);
                finRows = FiniteRows{nullptr, nullptr, 0};
                rowOverride = nullptr;
                if (rc != MAPLE_OK) return rc;
            }
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %zu pre-scored and %zu other wide searches replayed, %zu left\n", tms(tStart, tnow()), qn.size() - ql2.size(), ql2.size(), rest.size());
            wide.swap(rest);
        }
        const PlaceMeta &F = *c->place;
        const int nT = c->dtree.n, nF = c->tree_has_mut ? F.nF : 1;
        const size_t rowBytes = (size_t)nT * sizeof(double);
        size_t cacheBudget = (size_t)4ull << 30;                      // (query x node) score table: 4 GiB, more on big trees
        {   // a row of a 1 000 000-tip tree is 16 MB and every launch over the table wants thousands of searches (one per
            // wavefront): up to half of what is free, 96 GiB at most
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                const size_t half = (freeB + c->s_cache.cap * sizeof(double)) / 2;
                cacheBudget = std::max(cacheBudget, std::min(half, (size_t)96ull << 30));
            }
        }
        size_t chunk = cacheBudget / rowBytes;
        if (chunk < 1) chunk = 1;
        // Trees with local references: the rows of the searches that ran over their budget are made in the ROOT's frame as well
        // (appendProbNode does not depend on the frame, see above) and replayed inside the frontier tier; only what the tier hands
        // back goes the frame-by-frame way.
        const bool rootRowsOK = nF > 1 && useFrontier && c->scan_valid && c->place && !c->tuning.noCladeScan && !c->tuning.wideOutsideFrontier
                                && wide.size() >= 64;
        std::vector<int32_t> frameWay;                                  // (searches of a tree with references left to the old path)
        for (int wpass = 0; wpass < 2; wpass++) {
        if (wpass == 1) { if (frameWay.empty()) break; wide.swap(frameWay); frameWay.clear(); }
        const bool rootRows = rootRowsOK && wpass == 0;
        size_t w0 = 0;
        while (w0 < wide.size()) {
            int m = (int)std::min(chunk, wide.size() - w0);
            if (nF > 1 && !rootRows) {
                // the removed list goes into EVERY reference frame: bound the batch by what the arena can take
                const int64_t freeEnt = (c->cap_ent - c->used_ent) / 3, freeAux = (c->cap_aux - c->used_aux) / 3;
                int64_t needEnt = 0, needAux = 0;
                int k = 0;
                for (; k < m; k++) {
                    const int32_t l = c->h_tree_lower[nodes[wide[w0 + k]]];
                    needEnt += (int64_t)nF * (c->h_n_ent[l] + 24);
                    needAux += (int64_t)nF * (c->h_n_aux[l] + 8);
                    if (needEnt > freeEnt || needAux > freeAux || (int64_t)(k + 1) * nF > (8ll << 20)) break;
                }
                if (k < m && k < 2048) {
                    // Too many frames for this arena (a 100 000-tip tree has ~2 000): batches this small would turn the
                    // replay into a chain of one-lane launches.  The remaining wide searches run lane-only instead.
                    std::vector<int32_t> rest, restSlot;
                    for (size_t w = w0; w < wide.size(); w++) { rest.push_back(nodes[wide[w]]); restSlot.push_back(wide[w]); }
                    TRY(run_queries(rest, restSlot, nullptr, 0, nullptr, 0));
                    break;
                }
                m = k;
            }
            std::vector<int32_t> qn(m), ql(m), sl(m);
            std::vector<uint8_t> qt(m);
            std::vector<double> qb(m);
            for (int k = 0; k < m; k++) {
                const int node = nodes[wide[w0 + k]];
                qn[k] = node; sl[k] = wide[w0 + k];
                ql[k] = c->h_tree_lower[node];                         // the removed subtree's lower list (M:6838)
                qt[k] = c->h_tree_tip[node];                           // isRemovedTip (M:6846)
                qb[k] = c->h_tree_dist[node];                          // removedBLen = dist[node] (M:9644)
            }
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: wide chunk of %d searches starts\n", tms(tStart, tnow()), m);
            {
                size_t need = (size_t)m * nT;
                if (need > c->s_cache.cap) need = std::min(std::max(need, 2 * c->s_cache.cap), std::max(chunk * (size_t)nT, need));
                HIPCK(c, c->s_cache.reserve_exact(need));
            }
            if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: score table reserved\n", tms(tStart, tnow())); }
            TRY(h2d(c, c->s_u8[3], qt.data(), (size_t)m));
            TRY(h2d(c, c->s_f64[3], qb.data(), (size_t)m));
            if (nF == 1 || rootRows) {
                int64_t chunkMark = -1;
                if (rootRows) {                                        // candidates (once per tree) and this chunk's removed lists in the root's frame
                    TRY(ensure_cand_root(c));
                    TRY(maple_arena_mark(c, &chunkMark));
                    std::vector<int32_t> qf(m);
                    for (int k = 0; k < m; k++) qf[k] = F.frameOf[qn[k]];
                    TRY(lists_to_root_frame(c, ql, qf));
                    useFin = true;
                }
                TRY(h2d(c, c->s_i32[6], ql.data(), (size_t)m));
                double qBytes = 0.0;                                   // each query list once per launch
                for (int k = 0; k < m; k++) qBytes += 8.0 * c->h_n_ent[ql[k]] + 8.0 * c->h_n_aux[ql[k]];
                TRY(fin_reserve((size_t)m));
                TRY(launch_append_queries(c, c->stream, m, c->s_i32[6].p, c->n_scored, rootRows ? c->s_cand_root.p : c->t_i32[8].p, 0, 0.0, c->s_cache.p, nT,
                                          rootRows ? c->t_cand_rank.p : c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p, MAPLE_K_SPR_SCORE,
                                          (double)m * c->scored_bytes_total + qBytes, nullptr, nullptr, nullptr, 0, 1, useFin ? c->s_fin_mask.p : nullptr));
                if (useFin) TRY(fin_prefix(c->stream, 0, (size_t)m));
                if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: scored\n", tms(tStart, tnow())); }
                if (useFin) finRows = FiniteRows{c->s_fin_mask.p, c->s_fin_prefix.p, finWords};
                if (useFrontier && useFin && !c->tuning.wideOutsideFrontier && m >= 64) {
                    // the searches that ran over their budget: back through the frontier tier as whole-tree searches -- their
                    // updating steps batched, their clades scanned over the rows just made (k_fr_replay_wide); what the tier
                    // hands back is replayed one wavefront per search as before, on its own row
                    std::vector<int32_t> rowId(m);
                    for (int k = 0; k < m; k++) rowId[k] = k;
                    FrontierWide fw2{rowId.data(), c->s_cache.p, finRows, nullptr, 1};
                    if (rootRows) {
                        TRY(h2d(c, c->s_frame_parent, F.frameParent.data(), F.frameParent.size()));
                        TRY(h2d(c, c->s_frame_node, F.frameNode.data(), F.frameNode.size()));
                        fw2.frameParent = c->s_frame_parent.p; fw2.frameNode = c->s_frame_node.p; fw2.nFrames = (int)F.frameParent.size();
                    }
                    std::vector<SearchOut> part(m);
                    FrontierStats fs2;
                    const int rcF = frontier_search(c, P, m, qn.data(), 1 << 30, 0, part.data(), poolW, poolA, poolUsed, poolCapW, poolCapA, &fs2, 0, &fw2);
                    if (rcF != MAPLE_OK) { finRows = FiniteRows{nullptr, nullptr, 0}; return rcF; }
                    std::vector<int32_t> qn2, sl2, rows2;
                    for (int k = 0; k < m; k++) {
                        if (part[k].status == FR_STATUS_FALLBACK || part[k].status == -5) { qn2.push_back(qn[k]); sl2.push_back(sl[k]); rows2.push_back(k); }
                        else ho[sl[k]] = part[k];
                    }
                    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %d searches over budget replayed inside the frontier tier (%d levels), %zu handed back\n",
                                      tms(tStart, tnow()), m, fs2.levels, qn2.size());
                    qn.swap(qn2); sl.swap(sl2);
                    if (rootRows) {                                     // (the one-wavefront-per-search replay wants the list in every frame)
                        for (int32_t i : sl) frameWay.push_back(i);
                        qn.clear();
                    }
                    if (!qn.empty()) rowOverride = &rows2;
                    const int rcW = qn.empty() ? MAPLE_OK : run_queries(qn, sl, c->s_cache.p, 0, nullptr, 0);
                    rowOverride = nullptr;
                    finRows = FiniteRows{nullptr, nullptr, 0};
                    TRY(rcW);
                } else if (rootRows) {                                  // (a tail chunk too small for the tier: the frame-by-frame way)
                    for (int32_t i : sl) frameWay.push_back(i);
                    finRows = FiniteRows{nullptr, nullptr, 0};
                } else {
                const int rcW = run_queries(qn, sl, c->s_cache.p, 0, nullptr, 0);
                finRows = FiniteRows{nullptr, nullptr, 0};
                TRY(rcW);
                }
                if (chunkMark >= 0) TRY(maple_arena_release(c, chunkMark));
                if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: replayed\n", tms(tStart, tnow()));
            } else {
                // the removed list in every MAT reference frame, along the paths the traversal itself takes
                // (passGenomeListThroughBranch up the chain of enclosing frames, M:6844-6847 / 7392, then down into every
                // other frame from the nearest frame already known, M:7119 / 7342)
                int64_t mark = 0;
                TRY(maple_arena_mark(c, &mark));
                std::vector<int32_t> R((size_t)m * nF, -1), cur(m), src, ml, out;
                std::vector<uint8_t> dir;
                for (int k = 0; k < m; k++) { cur[k] = F.frameOf[qn[k]]; R[(size_t)k * nF + cur[k]] = ql[k]; }
                for (;;) {                                             // up, one enclosing frame per round
                    src.clear(); ml.clear();
                    std::vector<int> who;
                    for (int k = 0; k < m; k++)
                        if (cur[k] != 0) { who.push_back(k); src.push_back(R[(size_t)k * nF + cur[k]]); ml.push_back(c->h_tree_mut[F.frameNode[cur[k]]]); }
                    if (who.empty()) break;
                    dir.assign(who.size(), 1);
                    out.resize(who.size());
                    TRY(maple_pass_branch_batch(c, (int32_t)who.size(), src.data(), ml.data(), dir.data(), out.data()));
                    for (size_t i = 0; i < who.size(); i++) {
                        const int k = who[i];
                        cur[k] = F.frameParent[cur[k]];
                        R[(size_t)k * nF + cur[k]] = out[i];
                    }
                }
                // down, one nesting level per round: on the device (k_fan_*), all (query, frame) items of the level at once
                double qBytes = 0.0;                                   // every frame's copy of the query that is read
                for (size_t k = 0; k < R.size(); k++) if (R[k] >= 0) qBytes += 8.0 * c->h_n_ent[R[k]] + 8.0 * c->h_n_aux[R[k]];
                TRY(h2d(c, c->s_i32[6], R.data(), R.size()));
                {
                    std::vector<int32_t> fm((size_t)nF, 0);
                    for (int f = 1; f < nF; f++) fm[f] = c->h_tree_mut[F.frameNode[f]];
                    TRY(h2d(c, c->s_i32[4], F.frameParent.data(), (size_t)nF));
                    TRY(h2d(c, c->s_i32[5], fm.data(), (size_t)nF));
                    HIPCK(c, hipStreamSynchronize(c->stream));
                }
                int a = 1;
                for (size_t l = 0; l < F.levelStart.size(); l++) {
                    const int b = F.levelStart[l];
                    TRY(fan_out_level(c, m, nF, a, b, c->s_i32[6].p, c->s_i32[4].p, c->s_i32[5].p, c->s_fan, c->s_fan_tmp, &qBytes));
                    a = b;
                }
                if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: removed lists in all %d frames\n", tms(tStart, tnow()), nF); }
                if (c->n_frame_chunks > 0 && m >= 32)
                    TRY(launch_append_queries(c, c->stream, m, c->s_i32[6].p, c->n_scored, c->t_i32[8].p, 0, 0.0, c->s_cache.p, nT,
                                              c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p, MAPLE_K_SPR_SCORE,
                                              (double)m * c->scored_bytes_total + qBytes, nullptr, nullptr, c->t_frame_chunks.p,
                                              c->n_frame_chunks, nF));
                else
                TRY(launch_place_score(c, m, nF, c->s_i32[6].p, c->n_scored, c->t_i32[8].p, c->t_scored_frame.p, 0, 0.0,
                                       c->s_cache.p, nT, c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p, MAPLE_K_SPR_SCORE,
                                       (double)m * c->scored_bytes_total + qBytes));
                if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: scored\n", tms(tStart, tnow())); }
                TRY(run_queries(qn, sl, c->s_cache.p, 0, c->s_i32[6].p, nF));
                if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: replayed\n", tms(tStart, tnow()));
                TRY(maple_arena_release(c, mark));
            }
            w0 += (size_t)m;
        }
        }   // (wpass)
    }
#ifdef MAPLE_SPR_PROFILE
    {
        for (int pass = 0; pass < 2; pass++) {
        long long ts = 0, tr = 0, tf = 0, ns = 0, nl = 0, mxs = 0, mxr = 0, mxf = 0, mxl = 0, cnt = 0, visits = 0;
        for (int i = 0; i < n; i++) {
            if (pass == 1 && !(ho[i].status == 0 && ho[i].nAppend <= wideBudget)) continue;   // pass 1: finished by the lane tier
            cnt++; visits += ho[i].nAppend;
            ts += ho[i].tStep; tr += ho[i].tReplay; tf += ho[i].tRefine; ns += ho[i].nSteps; nl += ho[i].nShortList;
            mxs = std::max<long long>(mxs, ho[i].tStep); mxr = std::max<long long>(mxr, ho[i].tReplay);
            mxf = std::max<long long>(mxf, ho[i].tRefine); mxl = std::max<long long>(mxl, ho[i].nShortList);
        }
        {
            std::vector<double> tot;
            for (int i = 0; i < n; i++) {
                const bool laneTier = ho[i].status == 0 && ho[i].nAppend <= wideBudget;
                if (ho[i].status != 0 || (pass == 1) != laneTier) continue;
                tot.push_back((ho[i].tStep + ho[i].tReplay + ho[i].tRefine) * 1e-5);
            }
            if (c->tuning.verbose > 1) {
                std::vector<int> idx;
                for (int i = 0; i < n; i++) if (ho[i].status == 0 && (ho[i].nAppend <= wideBudget) == (pass == 1)) idx.push_back(i);
                std::sort(idx.begin(), idx.end(), [&](int a, int b) {
                    return ho[a].tStep + ho[a].tReplay + ho[a].tRefine > ho[b].tStep + ho[b].tReplay + ho[b].tRefine; });
                for (size_t k = 0; k < std::min<size_t>(12, idx.size()); k++) {
                    const int i = idx[k];
                    const int32_t l = c->h_tree_lower[nodes[i]];
                    fprintf(stderr, "[maple]   slow search: node %d depth %d, %d placements, %d updating steps %.1f ms, visits %.1f ms, %d refinements %.1f ms, removed list %d entries\n",
                            nodes[i], c->h_depth[nodes[i]], ho[i].nAppend, ho[i].nSteps, ho[i].tStep * 1e-5, ho[i].tReplay * 1e-5,
                            ho[i].nShortList, ho[i].tRefine * 1e-5, l >= 0 ? c->h_n_ent[l] : -1);
                }
            }
            std::sort(tot.begin(), tot.end());
            if (!tot.empty())
                fprintf(stderr, "[maple] per-search time, %s (%zu): median %.2f ms, p90 %.2f, p99 %.2f, p99.9 %.2f, max %.2f\n",
                        pass ? "lane tier" : "dense tier", tot.size(), tot[tot.size() / 2], tot[tot.size() * 9 / 10],
                        tot[tot.size() * 99 / 100], tot[tot.size() * 999 / 1000], tot.back());
        }
        {
            long long tw = 0, tl = 0;
            for (int i = 0; i < n; i++) {
                if (pass == 1 && !(ho[i].status == 0 && ho[i].nAppend <= wideBudget)) continue;
                if (ho[i].rprWoff < 0) { tw += ho[i].rprN; tl += ho[i].rprNA; }
            }
            fprintf(stderr, "[maple]   of which inside append_walk %.1f ms, list lookup before it %.1f ms\n", tw * 1e-3, tl * 1e-3);
        }
        fprintf(stderr, "[maple] profile over %lld searches (%s; last launch each; %lld placements): updating steps %.1f ms total (max %.2f), "
                        "other visits %.1f (max %.2f), refine %.1f (max %.2f); %lld updating steps, %lld short-listed branches (max %lld)\n",
                cnt, pass ? "lane tier only" : "all", visits, ts * 1e-5, mxs * 1e-5, tr * 1e-5, mxr * 1e-5, tf * 1e-5, mxf * 1e-5, ns, nl, mxl);
        }
    }
#endif
    if (preMark >= 0) TRY(maple_arena_release(c, preMark));             // (the root-frame copies of this call)
    for (int i = 0; i < n; i++) {
        bestNode[i] = ho[i].bestNode; bestScore[i] = ho[i].bestScore;
        blen3[3 * i] = ho[i].blen[0]; blen3[3 * i + 1] = ho[i].blen[1]; blen3[3 * i + 2] = ho[i].blen[2];
        placement[i] = ho[i].placement; improvement[i] = ho[i].improvement; currentLK[i] = ho[i].currentLK;
        nAppend[i] = ho[i].nAppend; status[i] = ho[i].status;
    }
    if (outRprList) {                                                  // bestRemovedPartials become arena lists
        std::vector<int64_t> woff(n, 0), aoff(n, 0);
        std::vector<int32_t> ne(n, -1), na(n, 0);
        for (int i = 0; i < n; i++)
            if (ho[i].status == 0 && ho[i].rprWoff >= 0) { woff[i] = ho[i].rprWoff; aoff[i] = ho[i].rprAoff; ne[i] = ho[i].rprN; na[i] = ho[i].rprNA; }
        HIPCK(c, c->s_i32[2].reserve(n));
        HIPCK(c, c->s_i32[3].reserve(n));
        HIPCK(c, hipMemcpyAsync(c->s_i32[2].p, ne.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i32[3].p, na.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        TRY(stage_begin(c, (size_t)n * 48 + 256));
        STAGE(dwo, c, woff.data(), n); STAGE(dao, c, aoff.data(), n);
        TRY(stage_flush(c));
        TRY(commit_lists(c, n, dwo, dao, c->s_i32[2].p, c->s_i32[3].p, outRprList, poolW, poolA));
    }
    return MAPLE_OK;
}

extern "C" int maple_spr_search_visited(maple_ctx *c, int64_t cap, int32_t *query, int32_t *node, int64_t *n)
{
    if (!c || cap < 0 || !n || (cap && (!query || !node))) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    if (!c->last_search_frontier_only)
        return fail(c, MAPLE_ERR_STATE, "the last maple_spr_search_batch did not run wholly in the frontier tier (use wideSearchBudget < 0)");
    long long nn = 0;
    const int rc = frontier_export(c, cap, query, node, &nn);
    *n = nn;
    return rc;
}

extern "C" int maple_debug_frontier_levels(maple_ctx *c, int32_t cap, int64_t *itemsUpdating, int64_t *itemsCached, float *msUpdating,
                                           float *msCached, int32_t *n, int64_t *waveItemsSmall, int64_t *waveItemsBig)
{
    if (!c || cap < 0 || !itemsUpdating || !itemsCached || !msUpdating || !msCached || !n) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int nn = 0;
    const int rc = frontier_level_profile(c, cap, (long long *)itemsUpdating, (long long *)itemsCached, msUpdating, msCached, &nn,
                                          (long long *)waveItemsSmall, (long long *)waveItemsBig);
    *n = nn;
    return rc;
}

// Calibration of the FETCH_SIZE counter for THIS library's access pattern (MI355X_MICROARCH.md, HBM section: the
// counter is only calibrated for 16 B/lane coalesced streams).  Every lane walks its own contiguous 512-byte "list"
// with dependent 8-byte loads, exactly like a genome-list walk, over a buffer far larger than the 256 MiB Infinity
// Cache; the byte count is known, so FETCH_SIZE / bytes is the correction factor for k_append*.
__global__ __launch_bounds__(MAPLE_BLOCK) void k_calib_walk(const unsigned long long *buf, long long nLists, unsigned long long *sink)
{
    unsigned long long acc = 0;
    for (long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x; l < nLists; l += (long long)gridDim.x * blockDim.x) {
        const unsigned long long *p = buf + l * 64;
        unsigned idx = 0;
        for (int k = 0; k < 64; k++) {
            unsigned long long w = p[idx];
            acc += w;
            idx = (idx + 1 + (unsigned)(w & 0)) & 63;                    // data-dependent next index, like a cursor
        }
    }
    if (acc == 0x123456789abcdefull) *sink = acc;
}

// WRITE_SIZE calibration: mode 1 writes `bytes` as a coalesced 8-byte-per-lane stream, mode 2 writes ONE 8-byte value into
// every 64-byte line of the buffer (the score-matrix pattern of k_append_queries: a lane's score lands in a line of its own)
__global__ __launch_bounds__(MAPLE_BLOCK) void k_calib_write(unsigned long long *buf, long long nWords, int strideWords)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i * strideWords < nWords; i += (long long)gridDim.x * blockDim.x)
        buf[i * strideWords] = (unsigned long long)i;
}

extern "C" int maple_debug_calib_write(maple_ctx *c, uint64_t bytes, int32_t mode, int32_t repeats, float *ms)
{
    if (!c || bytes < 512 || repeats <= 0 || mode < 1 || mode > 2) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    unsigned long long *buf = nullptr;
    HIPCK(c, hipMalloc((void **)&buf, bytes));
    HIPCK(c, hipMemset(buf, 0, bytes));
    HIPCK(c, hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCK(c, hipEventCreate(&e0));
    HIPCK(c, hipEventCreate(&e1));
    HIPCK(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < repeats; r++)
        hipLaunchKernelGGL(k_calib_write, dim3(4096), dim3(MAPLE_BLOCK), 0, c->stream, buf, (long long)(bytes / 8), mode == 1 ? 1 : 8);
    HIPCK(c, hipEventRecord(e1, c->stream));
    HIPCK(c, hipEventSynchronize(e1));
    if (ms) HIPCK(c, hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return MAPLE_OK;
}

extern "C" int maple_debug_calib_walk(maple_ctx *c, uint64_t bytes, int32_t repeats, float *ms)
{
    if (!c || bytes < 512 || repeats <= 0) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    unsigned long long *buf = nullptr, *sink = nullptr;
    HIPCK(c, hipMalloc((void **)&buf, bytes));
    HIPCK(c, hipMalloc((void **)&sink, 8));
    HIPCK(c, hipMemset(buf, 1, bytes));
    HIPCK(c, hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCK(c, hipEventCreate(&e0));
    HIPCK(c, hipEventCreate(&e1));
    HIPCK(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < repeats; r++)
        hipLaunchKernelGGL(k_calib_walk, dim3(2048), dim3(MAPLE_BLOCK), 0, c->stream, buf, (long long)(bytes / 512), sink);
    HIPCK(c, hipEventRecord(e1, c->stream));
    HIPCK(c, hipEventSynchronize(e1));
    if (ms) HIPCK(c, hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(sink);
    return MAPLE_OK;
}

// Parity hooks for the two innermost device functions, which no batched operator exposes on their own: getPartialVec
// (M:4073-4141) with the caller's matrix (the reference passes mutMatrices[pos] = Q * siteRates[pos]) and simplify
// (M:3697-3717).  One lane per call.
struct MatCtx {                        // what gpv_vec / gpv_nuc need from a context: q(r, i, j) of THIS call's matrix
    const double *M;
    __device__ inline double q(double, int i, int j) const { return M[i * 4 + j]; }
};
struct ThrCtx { struct { double thresholdProb, thresholdProb4; } m; };

__global__ __launch_bounds__(MAPLE_BLOCK) void k_debug_gpv(int n, int usingErrorRate, const int32_t *i12, const double *totLen,
                                                           const double *M16, const double *errorRate, const double *vect,
                                                           const uint8_t *upNode, const uint8_t *flag, double *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        MatCtx c{M16 + 16 * (size_t)i};
        double o[4];
        if (i12[i] == 6) gpv_vec(c, 1.0, vect + 4 * (size_t)i, totLen[i], upNode[i] != 0, o);
        else if (usingErrorRate) gpv_nuc<MatCtx, true>(c, 1.0, i12[i], totLen[i], errorRate[i], upNode[i] != 0, flag[i] != 0, o);
        else gpv_nuc<MatCtx, false>(c, 1.0, i12[i], totLen[i], errorRate[i], upNode[i] != 0, false, o);
        for (int k = 0; k < 4; k++) out[4 * (size_t)i + k] = o[k];
    }
}

__global__ __launch_bounds__(MAPLE_BLOCK) void k_debug_simplify(int n, double thresholdProb, double thresholdProb4, const double *vec,
                                                                const int32_t *refA, int32_t *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        ThrCtx c;
        c.m.thresholdProb = thresholdProb; c.m.thresholdProb4 = thresholdProb4;
        out[i] = simplify(c, vec + 4 * (size_t)i, refA[i]);
    }
}

extern "C" int maple_debug_gpv_batch(maple_ctx *c, int32_t n, const int32_t *i12, const double *totLen, const double *M16,
                                     const double *errorRate, const double *vect4, const uint8_t *upNode, const uint8_t *flag,
                                     double *out4)
{
    if (!c || n < 0 || !i12 || !totLen || !M16 || !errorRate || !vect4 || !upNode || !flag || !out4) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(h2d(c, c->s_i32[0], i12, (size_t)n));
    TRY(h2d(c, c->s_f64[0], totLen, (size_t)n));
    TRY(h2d(c, c->s_f64[1], M16, (size_t)16 * n));
    TRY(h2d(c, c->s_f64[2], errorRate, (size_t)n));
    TRY(h2d(c, c->s_f64[3], vect4, (size_t)4 * n));
    TRY(h2d(c, c->s_u8[0], upNode, (size_t)n));
    TRY(h2d(c, c->s_u8[1], flag, (size_t)n));
    HIPCK(c, c->s_aux.reserve((size_t)4 * n));
    hipLaunchKernelGGL(k_debug_gpv, dim3(grid_for(n)), dim3(MAPLE_BLOCK), 0, c->stream, n, c->dm.usingErrorRate, c->s_i32[0].p,
                       c->s_f64[0].p, c->s_f64[1].p, c->s_f64[2].p, c->s_f64[3].p, c->s_u8[0].p, c->s_u8[1].p, c->s_aux.p);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out4, c->s_aux.p, (size_t)4 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

extern "C" int maple_debug_simplify_batch(maple_ctx *c, int32_t n, const double *vec4, const int32_t *refA, int32_t *out)
{
    if (!c || n < 0 || !vec4 || !refA || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(h2d(c, c->s_f64[0], vec4, (size_t)4 * n));
    TRY(h2d(c, c->s_i32[0], refA, (size_t)n));
    HIPCK(c, c->s_i32[1].reserve(n));
    hipLaunchKernelGGL(k_debug_simplify, dim3(grid_for(n)), dim3(MAPLE_BLOCK), 0, c->stream, n, c->dm.thresholdProb,
                       c->dm.thresholdProb4, c->s_f64[0].p, c->s_i32[0].p, c->s_i32[1].p);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->s_i32[1].p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

// debugging aid: record the visit sequence (t1, direction, needsUpdating, failedPasses, lastLK, midProb) of one query
// appendProbNode by a whole wavefront per pair (wave_dev.h), for parity tests against the one-lane walk and for timing
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_wave_append(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *pl,
                                                    const int32_t *cl, const uint8_t *tip, const double *bl, double *out)
{
    __shared__ Lds lds;
    __shared__ WaveLds wl;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double v = wave_append(c, list_ref(av, pl[i]), av.n_ent[pl[i]], list_ref(av, cl[i]), av.n_ent[cl[i]], tip[i] != 0, bl[i], wl);
        if (threadIdx.x == 0) out[i] = v;
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int maple_debug_wave_append_batch(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl, const uint8_t *tip,
                                             const double *bl, double *out, float *ms)
{
    if (!c || n < 0 || !pl || !cl || !tip || !bl || !out) return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    TRY(check_ids(c, n, pl, false, "parentList"));
    TRY(check_ids(c, n, cl, false, "childList"));
    TRY(stage_begin(c, (size_t)n * 32 + 256));
    STAGE(dpl, c, pl, n); STAGE(dcl, c, cl, n); STAGE(dtip, c, tip, n); STAGE(dbl, c, bl, n);
    TRY(stage_flush(c));
    HIPCK(c, c->s_f64[1].reserve(n));
    hipEvent_t e0, e1;
    TRY(ev_pair(c, &e0, &e1, MAPLE_K_OTHER, (double)n, 0.0));
    HIPCK(c, hipEventRecord(e0, c->stream));
    DISPATCH3(c, k_wave_append, <<<std::min(n, 256 * 16), 64, 0, c->stream>>>(c->d_model, view(c), n, dpl, dcl, dtip, dbl, c->s_f64[1].p));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(e1, c->stream));
    HIPCK(c, hipMemcpyAsync(out, c->s_f64[1].p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (ms) HIPCK(c, hipEventElapsedTime(ms, e0, e1));
    return MAPLE_OK;
}

extern "C" int maple_debug_trace_query(maple_ctx *c, int32_t query)
{
    if (!c) return MAPLE_ERR_ARG;
    c->trace_query = query;
    if (query >= 0) {
        HIPCK(c, c->s_trace_i.reserve(4 * 4096 + 4));
        HIPCK(c, c->s_trace_d.reserve(2 * 4096));
        HIPCK(c, hipMemset(c->s_trace_i.p, 0, (4 * 4096 + 4) * sizeof(int32_t)));
    }
    return MAPLE_OK;
}

extern "C" int maple_debug_trace_read(maple_ctx *c, int32_t *n, int32_t *items4, double *vals2)
{
    if (!c || !n || !items4 || !vals2 || !c->s_trace_i.p) return MAPLE_ERR_ARG;
    HIPCK(c, hipMemcpy(n, c->s_trace_i.p + 4 * 4096, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(items4, c->s_trace_i.p, 4 * 4096 * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(vals2, c->s_trace_d.p, 2 * 4096 * sizeof(double), hipMemcpyDeviceToHost));
    return MAPLE_OK;
}

extern "C" int maple_timing_reset(maple_ctx *c)
{
    if (!c) return MAPLE_ERR_ARG;
    c->ev_used = 0;
    return MAPLE_OK;
}

extern "C" int maple_timing_read_each(maple_ctx *c, int32_t cap, float *ms, int32_t *n_launches)
{
    if (!c || cap < 0 || !ms || !n_launches) return MAPLE_ERR_ARG;
    int k = 0;
    for (size_t i = 0; i + 1 < c->ev_used && k < cap; i += 2, k++) {
        HIPCK(c, hipEventSynchronize(c->evs[i + 1]));
        HIPCK(c, hipEventElapsedTime(&ms[k], c->evs[i], c->evs[i + 1]));
    }
    *n_launches = k;
    return MAPLE_OK;
}

extern "C" int maple_timing_read(maple_ctx *c, int32_t *n_launches, double *total_ms)
{
    if (!c || !n_launches || !total_ms) return MAPLE_ERR_ARG;
    double tot = 0.0;
    for (size_t k = 0; k + 1 < c->ev_used; k += 2) {
        float ms = 0.f;
        HIPCK(c, hipEventSynchronize(c->evs[k + 1]));
        HIPCK(c, hipEventElapsedTime(&ms, c->evs[k], c->evs[k + 1]));
        tot += ms;
    }
    *n_launches = (int32_t)(c->ev_used / 2);
    *total_ms = tot;
    return MAPLE_OK;
}

extern "C" int maple_timing_read_kind(maple_ctx *c, int32_t kind, int32_t *n_launches, double *total_ms, double *units,
                                      double *alg_bytes)
{
    if (!c || !n_launches || !total_ms) return MAPLE_ERR_ARG;
    double tot = 0.0, u = 0.0, b = 0.0;
    int32_t n = 0;
    for (size_t k = 0; k + 1 < c->ev_used; k += 2) {
        if (c->ev_kind[k / 2] != kind) continue;
        float ms = 0.f;
        HIPCK(c, hipEventSynchronize(c->evs[k + 1]));
        HIPCK(c, hipEventElapsedTime(&ms, c->evs[k], c->evs[k + 1]));
        tot += ms; u += c->ev_units[k / 2]; b += c->ev_bytes[k / 2];
        n++;
    }
    *n_launches = n; *total_ms = tot;
    if (units) *units = u;
    if (alg_bytes) *alg_bytes = b;
    return MAPLE_OK;
}

extern "C" int maple_append_algorithmic_bytes(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl,
                                              int child_once, uint64_t *bytes)
{
    if (!c || n < 0 || !pl || !bytes) return MAPLE_ERR_ARG;
    TRY(check_ids(c, n, pl, false, "parentList"));
    uint64_t b = 0;
    // 8 B per entry word + 8 B per aux double (d0/d1 scalars, 4 per O vector) + 8 B result   
    for (int i = 0; i < n; i++) b += 8ull * c->h_n_ent[pl[i]] + 8ull * c->h_n_aux[pl[i]] + 8ull;
    if (cl) {
        if (child_once) b += 8ull * c->h_n_ent[cl[0]] + 8ull * c->h_n_aux[cl[0]];
        else {
            TRY(check_ids(c, n, cl, false, "childList"));
            for (int i = 0; i < n; i++) b += 8ull * c->h_n_ent[cl[i]] + 8ull * c->h_n_aux[cl[i]];
        }
    }
    *bytes = b;
    return MAPLE_OK;
}
