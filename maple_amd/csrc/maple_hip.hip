// maple_amd/csrc/maple_hip.hip -- libmaple_hip.so: context, model, arena, the batched list operators and their kernels, the
// placement search, updatePartials, the RCCL arg-max, measurement aids (C ABI: include/maple_hip.h).  The tree mirror and the SPR
// search batch live in spr_batch.hip, the frontier tier in frontier*.hip, the witness filter in witness.hip.
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
#include "batch_host.h"
#include "frontier.h"
#include "witness.h"

// =================================================================================================
// kernels
// =================================================================================================

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
#ifndef MAPLE_DENSE_PLAIN
                // (staged in the SKIPPING form of append_lds.h: the tail-less reference runs in front of single-site entries are left
                // out -- half the entries, and with them half the steps of every walk over the chunk)
                int kept = 0;
                for (int j0 = 0; j0 < nw; j0 += 64) {
                    const int j = j0 + lane;
                    unsigned long long w = 0;
                    bool keep = false;
                    if (j < nw) { w = sw[j]; keep = !skip_form_drops(w, j + 1 < nw ? sw[j + 1] : 0ull, j + 1 == nw); }
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {
                        const int d = w0 + kept + __popcll(bal & ((1ull << lane) - 1ull));
                        cW[d] = w;
                        if (RV) cR[d] = c.rate((int)(uint32_t)w - 1);
                    }
                    kept += __popcll(bal);
                }
#else
                for (int j = lane; j < nw; j += 64) {
                    const unsigned long long w = sw[j];
                    cW[w0 + j] = w;
                    if (RV) cR[w0 + j] = c.rate((int)(uint32_t)w - 1);
                }
#endif
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
#ifndef MAPLE_DENSE_PLAIN
                const unsigned long long *qsrc = (const unsigned long long *)qref.w;
                int kept = 0;
                for (int j0 = 0; j0 < nq; j0 += 64) {
                    const int j = j0 + lane;
                    unsigned long long w = 0;
                    bool keep = false;
                    if (j < nq) { w = qsrc[j]; keep = !skip_form_drops(w, j + 1 < nq ? qsrc[j + 1] : 0ull, j + 1 == nq); }
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {
                        const int d = kept + __popcll(bal & ((1ull << lane) - 1ull));
                        myq[d] = w;
                        if (RV) myqR[d] = c.rate((int)(uint32_t)w - 1);
                    }
                    kept += __popcll(bal);
                }
#else
                for (int i = lane; i < nq; i += 64) {
                    const unsigned long long w = ((const unsigned long long *)qref.w)[i];
                    myq[i] = w;
                    if (RV) myqR[i] = c.rate((int)(uint32_t)w - 1);
                }
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            bool finite = false;
#ifndef MAPLE_DENSE_PLAIN
            // (the walk in skipping form votes over the whole wavefront, append_lds.h: every lane calls, those without a candidate
            // with valid = false)
            const bool validL = cl >= 0;
            double lkAll;
            {
                const bool tipq = qTip ? qTip[q] != 0 : isTip != 0;
                const double blq = qBLen ? qBLen[q] : bLen;
                const MemLG qL{(lds_u64p)myq, qref.aux, (lds_f64p)myqR};
                const MemG qG{(const unsigned long long *)qref.w, qref.aux};
                if (stagedC) {                                              // (block-uniform; stagedQ is wave-uniform)
                    const MemL pL{(lds_u64p)(cW + myW), (lds_f64p)(cA + myA), (lds_f64p)(cR + myW)};
                    lkAll = stagedQ ? append_walk_c(c, pL, qL, tipq, blq, validL) : append_walk_c(c, pL, qG, tipq, blq, validL);
                } else {
                    const ListRef pr = validL ? list_ref(av, cl) : qref;
                    const MemG pG{(const unsigned long long *)pr.w, pr.aux};
                    lkAll = stagedQ ? append_walk_c(c, pG, qL, tipq, blq, validL) : append_walk_c(c, pG, qG, tipq, blq, validL);
                }
            }
#endif
            if (cl >= 0) {
#ifndef MAPLE_DENSE_PLAIN
                const double lk = lkAll;
#else
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
#endif
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
    if (t->structSize < sizeof(uint32_t) + sizeof(int32_t)) return fail(c, MAPLE_ERR_ARG, "maple_set_tuning: structSize is not set (sizeof(maple_tuning) of the caller's header)");
    const int32_t verbose = c->tuning.verbose;
    maple_tuning mine{};                                              // (what the caller's struct does not reach keeps the library's choice)
    memcpy(&mine, t, std::min<size_t>(t->structSize, sizeof(maple_tuning)));
    mine.structSize = (uint32_t)sizeof(maple_tuning);
    c->tuning = mine;
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
        M.d_leafList.release(); M.d_leafFrame.release(); M.d_pn.release();
        delete c->place;
    }
    if (c->ahead) { c->ahead->release(); delete c->ahead; }
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
    c->h_over_hint.clear();                                           // (which searches run over the budget is a property of the model too)
    if (c->ahead) { c->ahead->join(); c->ahead->spec.row = -1; c->ahead->active = false; }   // (rows scored under another model)
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
    // order) move with one copy per run instead of one per list; the copies of up to 32 M entries are queued together and
    // awaited once (a tree whose four kinds of list are interleaved in the arena is millions of one-list runs)
    std::vector<uint2> w;
    struct Run { int first; int64_t ne, wOff; };
    std::vector<Run> runs;
    int i = 0;
    while (i < n) {
        runs.clear();
        int64_t tot = 0;
        const int chunkStart = i;
        while (i < n && tot < ((int64_t)32 << 20) && runs.size() < (size_t)1 << 16) {
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
            runs.push_back(Run{i, ne, tot});
            tot += ne;
            const int id0 = ids[i];
            if (na) HIPCK(c, hipMemcpyAsync(aux + aux_off[i], c->d_aux + c->h_aux_off[id0], na * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            i = j;
        }
        (void)chunkStart;
        w.resize((size_t)tot);
        for (const Run &r : runs)
            if (r.ne) HIPCK(c, hipMemcpyAsync(w.data() + r.wOff, c->d_words + c->h_ent_off[ids[r.first]], r.ne * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        for (const Run &r : runs) {
            const int64_t o = ent_off[r.first];
            for (int64_t k = 0; k < r.ne; k++) { pos[o + k] = (int32_t)w[r.wOff + k].x; meta[o + k] = w[r.wOff + k].y; }
        }
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
    if (c->ahead && c->ahead->active)                                   // (rows of samples whose lists go with the release)
        for (int32_t id : c->ahead->q) if (id >= mark) { c->ahead->active = false; break; }
    if (mark < c->cand_root_end) c->cand_root_end = -1;                 // (the candidates' root-frame copies go with the release)
    if (mark < c->cand_root_top) c->cand_root_mark = c->cand_root_top = -1;
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
    if (c && c->ahead) { c->ahead->join(); c->ahead->spec.row = -1; c->ahead->active = false; }
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
    c->cand_root_mark = c->cand_root_top = -1;                          // (the lists were renumbered)
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
// Move freshly produced scratch lists into the arena and hand out ids (or -1 for None).  d_woff / d_aoff: the per-item
// scratch offsets, already on the device.  One synchronisation (the sizes come back), then one staged copy (destinations and
// list ids) and the copy kernel, which also writes the new rows of the device-side list table; nothing waits for it.
static int commit_known(maple_ctx *c, int32_t n, const int64_t *d_woff, const int64_t *d_aoff, const int32_t *d_n_ent,
                        const int32_t *d_n_aux, const std::vector<int32_t> &ne, const std::vector<int32_t> &na, int32_t *outList,
                        const uint2 *srcW, const double *srcA);
int commit_lists(maple_ctx *c, int32_t n, const int64_t *d_woff, const int64_t *d_aoff, int32_t *d_n_ent, int32_t *d_n_aux,
                 int32_t *outList, const uint2 *srcW, const double *srcA)
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
int launch_append_queries(maple_ctx *c, hipStream_t s, int nQ, const int32_t *qList, int nC, const int32_t *cand,
                          int isTip, double bLen, double *out, long long ldOut, const int32_t *outCol,
                          const uint8_t *qTip, const double *qBLen, int kind, double algBytes, TileBest *tileBest,
                          const int32_t *visitRank, const int4 *chunkTab, int nChunkTab, int nF,
                          unsigned long long *finMask, bool lanesOnly)
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
    // (lanesOnly: many queries against a HANDFUL of candidates -- the columns a placement changed, for every sample still waiting,
    // placement_host.h: the LDS kernel would be one workgroup with a few lanes of each wavefront at work)
    if (!lanesOnly && (chunkTab || nQ >= 32)) {
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

#include "placement_host.h"
#include "update_host.h"
#include "rebuild_host.h"

// appendProbNode by a whole wavefront per pair (wave_dev.h), small batches of maple_append_batch; the hook of the parity tests is below
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

#ifdef MAPLE_DEBUG_ABI                                                   // (libmaple_hip_debug.so: include/maple_hip_debug.h)
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

#endif  // MAPLE_DEBUG_ABI

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
