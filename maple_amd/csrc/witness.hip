// maple_amd/csrc/witness.hip -- which branches can a whole-tree SPR search score above -inf at all?
//
// A pruned node on a zero-length branch is searched with removedBLen = 0 (MAPLEv0.7.5.4.py:9644).  Without an error model
// appendProbNode then returns -inf as soon as the candidate's probVectTotUp holds a nucleotide X WITHOUT a stored length at
// a site where the removed list holds another nucleotide, also without one: the total length between the two observations
// is zero and they differ (M:6663 on the way down the lists, M:6742 for the reference nucleotide).  The dense tier walked
// every (search, branch) pair to find that out -- 5 x 10^9 walks per round of the 100 000-tip bench tree, of which 1 in 1000
// ends finite.  Here every candidate branch names ONE such entry of its list as its WITNESS -- the one the fewest other
// candidates share, so the one a random search is least likely to agree with -- and the candidates are bucketed by witness
// (site, nucleotide).  A search then collects the buckets its removed list is compatible with: its own nucleotide at a
// site, every nucleotide where it holds N, an O vector or an entry with a stored length (there the walk may well be finite),
// nothing at all where it holds the reference without a length.  Only those pairs are walked, with the same appendProbNode
// as everywhere else; every other pair is -inf BY THE WALK'S OWN RULE, so the rows that come out -- finite scores plus the
// bitmap of where they are (FiniteRows, search_dev.h) -- are the rows the dense kernel writes, bit for bit.
//
// Exclusion needs a proof, inclusion does not: a candidate is left out only if its witness entry (type 0-3 = a nucleotide
// that is not the reference's, no stored length, no second length) meets an entry of the removed list that is a nucleotide
// or the reference, again without lengths.  In the walk that pair takes the "two different nucleotides" branch with
// cl = bLen + 0 + 0 = 0 and no error model: dead (append_walk, genome_dev.h; append_walk_m, append_lds.h).
#include "ctx_host.h"
#include "witness.h"

#include <algorithm>

namespace {

#define WIT_BLOCK 256

struct WitnessScratch {
    DevBuf<int32_t> hist, witB, cnt, start, cursor, bcand, qCnt, pairQ, pairK;
    DevBuf<long long> qOff;
};

// a witness: a nucleotide other than the reference's (the entry keeps the reference's in bits 3-4), no lengths attached
__device__ __forceinline__ bool wit_eligible(uint32_t meta)
{
    const uint32_t t = meta & 7u;
    return t < 4u && !(meta & (3u << 5)) && t != ((meta >> 3) & 3u);
}
__device__ __forceinline__ int wit_key(uint2 e) { return (int)e.x * 4 + (int)(e.y & 7u); }

__global__ __launch_bounds__(WIT_BLOCK) void k_wit_hist(ArenaView av, int nC, const int32_t *cand, int32_t *hist)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nC; k += gridDim.x * blockDim.x) {
        const int id = cand[k];
        if (id < 0) continue;
        const uint2 *w = av.words + av.ent_off[id];
        const int n = av.n_ent[id];
        for (int i = 0; i < n; i++) {
            const uint2 e = w[i];
            if (wit_eligible(e.y)) atomicAdd(&hist[wit_key(e)], 1);
        }
    }
}

// bucket of a candidate: 1 + the key of its rarest eligible entry (the first of equally rare ones), 0 = it has none
__global__ __launch_bounds__(WIT_BLOCK) void k_wit_pick(ArenaView av, int nC, const int32_t *cand, const int32_t *hist, int32_t *witB,
                                                        int32_t *cnt)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nC; k += gridDim.x * blockDim.x) {
        const int id = cand[k];
        int best = -1, bestH = 0x7fffffff;
        if (id >= 0) {
            const uint2 *w = av.words + av.ent_off[id];
            const int n = av.n_ent[id];
            for (int i = 0; i < n; i++) {
                const uint2 e = w[i];
                if (!wit_eligible(e.y)) continue;
                const int key = wit_key(e), h = hist[key];
                if (h < bestH) { bestH = h; best = key; }
            }
        }
        witB[k] = best + 1;
        atomicAdd(&cnt[best + 1], 1);
    }
}

// exclusive prefix sums of n counts by ONE workgroup (n ~ 10^5): out[0 .. n], out[n] = the total
template <class TI, class TO>
__global__ __launch_bounds__(1024) void k_wit_scan(int n, const TI *in, TO *out)
{
    __shared__ TO part[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    TO s = 0;
    for (int i = lo; i < hi; i++) s += (TO)in[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        TO run = 0;
        for (int i = 0; i < 1024; i++) { const TO v = part[i]; part[i] = run; run += v; }
        out[n] = run;
    }
    __syncthreads();
    TO run = part[t];
    for (int i = lo; i < hi; i++) { out[i] = run; run += (TO)in[i]; }
}

__global__ __launch_bounds__(WIT_BLOCK) void k_wit_fill(int nC, const int32_t *witB, const int32_t *start, int32_t *cursor, int32_t *bcand)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nC; k += gridDim.x * blockDim.x) {
        const int b = witB[k];
        bcand[start[b] + atomicAdd(&cursor[b], 1)] = k;
    }
}

// The buckets a removed list is compatible with, entry by entry (an entry covers the sites after the one before it up to its
// own position): f(first bucket, last bucket) for every run of buckets, the bucket of the candidates without a witness first.
template <class F> __device__ __forceinline__ void wit_ranges(const uint2 *w, int n, F f)
{
    f(0, 0);
    int prev = 0;
    for (int i = 0; i < n; i++) {
        const uint2 e = w[i];
        const int s0 = prev + 1, s1 = (int)e.x;
        prev = s1;
        const uint32_t t = e.y & 7u;
        if (t >= 5u || (e.y & (3u << 5))) f(s0 * 4 + 1, s1 * 4 + 3 + 1);     // N, O, or lengths attached: every witness at these sites
        else if (t < 4u) f(s1 * 4 + (int)t + 1, s1 * 4 + (int)t + 1);       // a nucleotide: the candidates whose witness is that nucleotide
        // (the reference without a length: no witness at these sites is compatible)
    }
}

__global__ __launch_bounds__(WIT_BLOCK) void k_wit_count(ArenaView av, int nQ, const int32_t *qList, const int32_t *start, int32_t *qCnt)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nQ; q += gridDim.x * blockDim.x) {
        const int id = qList[q];
        int tot = 0;
        wit_ranges(av.words + av.ent_off[id], av.n_ent[id], [&](int b0, int b1) { tot += start[b1 + 1] - start[b0]; });
        qCnt[q] = tot;
    }
}

__global__ __launch_bounds__(WIT_BLOCK) void k_wit_pairs(ArenaView av, int nQ, const int32_t *qList, const int32_t *start, const int32_t *bcand,
                                                         const long long *qOff, int32_t *pairQ, int32_t *pairK)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nQ; q += gridDim.x * blockDim.x) {
        const int id = qList[q];
        long long o = qOff[q];
        wit_ranges(av.words + av.ent_off[id], av.n_ent[id], [&](int b0, int b1) {
            for (int j = start[b0]; j < start[b1 + 1]; j++) { pairQ[o] = q; pairK[o] = bcand[j]; o++; }
        });
    }
}

// appendProbNode of the pairs that are left; a finite score goes to its place in the search's row, its bit into the bitmap
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(WIT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_wit_score(const DevModel *__restrict__ mp, ArenaView av, long long nPairs, const int32_t *pairQ, const int32_t *pairK,
                 const int32_t *cand, const int32_t *qList, const uint8_t *qTip, const double *qBLen, double *out, long long ldOut,
                 const int32_t *outCol, unsigned long long *finMask, int nWords)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nPairs; i += (long long)gridDim.x * blockDim.x) {
        const int q = pairQ[i], k = pairK[i];
        const int cl = cand[k];
        if (cl < 0) continue;
        const double lk = append_walk(c, list_ref(av, cl), list_ref(av, qList[q]), qTip[q] != 0, qBLen[q]);
        if (lk > -INFINITY) {
            out[(long long)q * ldOut + (outCol ? outCol[k] : k)] = lk;
            atomicOr(&finMask[(long long)q * nWords + (k >> 6)], 1ull << (k & 63));
        }
    }
}

#define WIT_DISPATCH3(c, KERNEL, ...)                                                                      \
    do {                                                                                                  \
        const bool rv_ = (c)->dm.useRateVariation, u_ = (c)->dm.usingErrorRate, ss_ = (c)->dm.errorRateSiteSpecific; \
        if (!rv_ && !u_) KERNEL<false, false, false> __VA_ARGS__;                                          \
        else if (rv_ && !u_) KERNEL<true, false, false> __VA_ARGS__;                                       \
        else if (!rv_ && u_ && !ss_) KERNEL<false, true, false> __VA_ARGS__;                               \
        else if (!rv_ && u_ && ss_) KERNEL<false, true, true> __VA_ARGS__;                                 \
        else if (rv_ && u_ && !ss_) KERNEL<true, true, false> __VA_ARGS__;                                 \
        else KERNEL<true, true, true> __VA_ARGS__;                                                         \
    } while (0)

}  // namespace

void witness_scratch_free(maple_ctx *c)
{
    WitnessScratch *W = (WitnessScratch *)c->witness;
    if (!W) return;
    W->hist.release(); W->witB.release(); W->cnt.release(); W->start.release(); W->cursor.release(); W->bcand.release();
    W->qCnt.release(); W->pairQ.release(); W->pairK.release(); W->qOff.release();
    delete W;
    c->witness = nullptr;
}

// Rows of the score table for nQ whole-tree searches (removed lists qList, ALL searched with removedBLen = 0 -- the caller checks
// its host copy of the lengths -- no error model; on a tree with local references both sides re-expressed in the root's frame) against the nC candidate lists `cand`: out[q * ldOut + outCol[k]] for the finite scores,
// finMask[q * nWords + k / 64] bit k % 64 set exactly for those.  Blocks on `st` once (the number of pairs).
int witness_score(maple_ctx *c, hipStream_t st, int nQ, const int32_t *qList, const uint8_t *qTip, const double *qBLen, int nC,
                  const int32_t *cand, const int32_t *outCol, double *out, long long ldOut, unsigned long long *finMask, int nWords,
                  double meanCandBytes, double queryBytes, long long *pairsOut)
{
    if (c->dm.usingErrorRate) return fail(c, MAPLE_ERR_STATE, "witness_score: not valid with an error model");
    if (!c->witness) c->witness = new WitnessScratch();
    WitnessScratch &W = *(WitnessScratch *)c->witness;
    const int nB = 4 * (c->lRef + 1) + 1;                                   // bucket 0 = no witness, 1 + site * 4 + nucleotide
    HIPCK(c, W.hist.reserve((size_t)nB)); HIPCK(c, W.cnt.reserve((size_t)nB)); HIPCK(c, W.cursor.reserve((size_t)nB));
    HIPCK(c, W.start.reserve((size_t)nB + 1));
    HIPCK(c, W.witB.reserve((size_t)nC)); HIPCK(c, W.bcand.reserve((size_t)nC));
    HIPCK(c, W.qCnt.reserve((size_t)nQ)); HIPCK(c, W.qOff.reserve((size_t)nQ + 1));
    hipEvent_t e0, e1;
    TRY(maple_internal_ev_pair(c, &e0, &e1, MAPLE_K_SPR_SCORE, 0.0, 0.0));
    const size_t slot = c->ev_used / 2 - 1;
    HIPCK(c, hipEventRecord(e0, st));
    HIPCK(c, hipMemsetAsync(W.hist.p, 0, (size_t)nB * sizeof(int32_t), st));
    HIPCK(c, hipMemsetAsync(W.cnt.p, 0, (size_t)nB * sizeof(int32_t), st));
    HIPCK(c, hipMemsetAsync(W.cursor.p, 0, (size_t)nB * sizeof(int32_t), st));
    HIPCK(c, hipMemsetAsync(finMask, 0, (size_t)nQ * nWords * sizeof(unsigned long long), st));
    const ArenaView av = view(c);
    const int gC = std::max(1, std::min(4096, (nC + WIT_BLOCK - 1) / WIT_BLOCK)), gQ = std::max(1, std::min(4096, (nQ + WIT_BLOCK - 1) / WIT_BLOCK));
    k_wit_hist<<<gC, WIT_BLOCK, 0, st>>>(av, nC, cand, W.hist.p);
    k_wit_pick<<<gC, WIT_BLOCK, 0, st>>>(av, nC, cand, W.hist.p, W.witB.p, W.cnt.p);
    k_wit_scan<int32_t, int32_t><<<1, 1024, 0, st>>>(nB, W.cnt.p, W.start.p);
    k_wit_fill<<<gC, WIT_BLOCK, 0, st>>>(nC, W.witB.p, W.start.p, W.cursor.p, W.bcand.p);
    k_wit_count<<<gQ, WIT_BLOCK, 0, st>>>(av, nQ, qList, W.start.p, W.qCnt.p);
    k_wit_scan<int32_t, long long><<<1, 1024, 0, st>>>(nQ, W.qCnt.p, W.qOff.p);
    HIPCK(c, hipGetLastError());
    long long nPairs = 0;
    HIPCK(c, hipMemcpyAsync(&nPairs, W.qOff.p + nQ, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCK(c, hipStreamSynchronize(st));
    if (nPairs > 0) {
        HIPCK(c, W.pairQ.reserve_exact(std::max(W.pairQ.cap, (size_t)nPairs)));
        HIPCK(c, W.pairK.reserve_exact(std::max(W.pairK.cap, (size_t)nPairs)));
        k_wit_pairs<<<gQ, WIT_BLOCK, 0, st>>>(av, nQ, qList, W.start.p, W.bcand.p, W.qOff.p, W.pairQ.p, W.pairK.p);
        const int gP = (int)std::max<long long>(1, std::min<long long>(8192, (nPairs + WIT_BLOCK - 1) / WIT_BLOCK));
        WIT_DISPATCH3(c, k_wit_score, <<<gP, WIT_BLOCK, 0, st>>>(c->d_model, av, nPairs, W.pairQ.p, W.pairK.p, cand, qList, qTip, qBLen, out,
                                                                 ldOut, outCol, finMask, nWords));
        HIPCK(c, hipGetLastError());
    }
    HIPCK(c, hipEventRecord(e1, st));
    c->ev_units[slot] = (double)nPairs;
    c->ev_bytes[slot] = (double)nPairs * meanCandBytes + queryBytes;
    if (pairsOut) *pairsOut = nPairs;
    return MAPLE_OK;
}
