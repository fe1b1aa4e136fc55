// maple_amd/csrc/wave_update.h -- one item of a level of updatePartials (M:5479-5815) by a whole wavefront.
//
// mergeVectors (M:4446-4859), shorten (M:3721-3745) and areVectorsDifferent (M:5419-5472)
// are walks over two lists in which a step needs nothing from the steps before it but the position reached -- and that is
// known from the lists' end positions alone.  So, as in wave_dev.h, the walk is cut along its merge path: lane d finds the
// two entries of step d with a binary search over the end positions (both lists staged in LDS) and runs merge_step /
// differ_step (below: the step of the one-lane walks, merge_step_body.inc) on them; the entries of the result are then compacted in
// walk order with a prefix sum over the wavefront.  Same entries, same arithmetic per entry: the lists are the one-lane
// walk's, bit for bit.  One lane takes ~0.1 ms per list operation (a chain of dependent loads), the wavefront a few us:
// what a single-change updatePartials -- a chain of ~8 levels, each waiting for the one before -- is made of.
// Every lane of the wavefront calls these with the same arguments and gets the same result.
#pragma once
#include "genome_dev.h"
#include "wave_dev.h"

namespace maple {

#ifndef MAPLE_WU_IN
#define MAPLE_WU_IN 256                // entries per input list the cooperative walks stage (longer: one lane's walk)
#endif
#define MAPLE_WU_CAP (2 * MAPLE_WU_IN)

struct WaveUpdLds {                    // per wavefront
    unsigned long long in[MAPLE_WU_CAP];       // words of the two input lists; later those of the shortened list
    unsigned long long m[MAPLE_WU_CAP];        // merged list
    unsigned long long old[MAPLE_WU_CAP];      // the list the new one is compared with
    double maux[5 * MAPLE_WU_CAP], baux[5 * MAPLE_WU_CAP];
    int mark[MAPLE_WU_CAP];                    // shorten: 1 = absorbed by the entry before, 2 = to be decided in order
    unsigned long long seq[MAPLE_WU_CAP / 64];
};

__device__ inline void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Step d of the walk over lists A and B (end positions in the low words): i, k = entries of A, B the step looks at.
// False if d is no step of its own (the second half of a tie: both lists ended at the same position, one step).
__device__ inline bool merge_path(const unsigned long long *A, int nA, const unsigned long long *B, int nB, int d, int &i, int &k)
{
    int lo = max(0, d - nB), hi = min(d, nA);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((uint32_t)A[mid] <= (uint32_t)B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    i = lo; k = d - lo;
    if (i < nA && (k >= nB || (uint32_t)A[i] <= (uint32_t)B[k])) return k < nB;
    return i < nA && k < nB && !(i > 0 && (uint32_t)A[i - 1] == (uint32_t)B[k]);
}

__device__ inline int wave_excl_sum(int v, int lane, int &total)
{
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        incl += (lane >= off) ? t : 0;
    }
    total = __shfl(incl, 63);
    return incl - v;
}

__device__ inline int aux_count(uint32_t meta)
{
    return (int)((meta >> 5) & 1u) + (int)((meta >> 6) & 1u) + (((meta & 7u) == 6u) ? 4 : 0);
}

// One step of mergeVectors on its own: the body the one-lane walk runs (merge_step_body.inc), without the likelihood.
template <bool RV, bool U, bool SS>
__device__ __forceinline__ int merge_step(const Ctx<RV, U, SS> &c, const Ent &e1, const Ent &e2, const int pos, const double bLen1,
                                          const bool tip1, const double bLen2, const bool tip2, const bool upDown, Writer &o)
{
    typedef Ctx<RV, U, SS> CT;
    const double *rf = c.rf;
    const double *cr = c.m.cumulativeRate, *cer = c.m.cumulativeErrorRate;
    constexpr bool wantLK = false;
    double lk = 0.0, totalFactor = 1.0;
    int newPos;
#include "merge_step_body.inc"
    (void)lk; (void)totalFactor; (void)cr; (void)cer; (void)newPos;
    return 0;
}

// One step of areVectorsDifferent (M:5419-5472): do the two entries that cover a stretch of positions differ?  The
// conditions of differ_walk (genome_dev.h), which returns at the first step that says yes.
template <class C> __device__ __forceinline__ bool differ_step(const C &c, const Ent &e1, const Ent &e2)
{
    const double thr = c.m.thresholdProb;
    if (e1.type != e2.type) return true;
    if (e1.hasD0 != e2.hasD0 || e1.hasD1 != e2.hasD1) return true;         // tuple lengths
    if (e1.type < 5) {
        if (e1.hasD0) {
            if (fabs(e1.d0 - e2.d0) > thr) return true;
            if (e1.hasD1 && fabs(e1.d1 - e2.d1) > thr) return true;
            if (e1.flag != e2.flag) return true;                            // |True-False| = 1 > thr
        }
    } else if (e1.type == 6) {
        if (e1.hasD0 && fabs(e1.d0 - e2.d0) > thr) return true;
        for (int i = 0; i < 4; i++) {
            double x = e1.vec[i], y = e2.vec[i];
            double d = fabs(x - y);
            if (d != 0.0) {
                if (x == 0.0 || y == 0.0) return true;
                if (d > c.m.thresholdDiffForUpdate
                    || (d > thr && ((d / x > c.m.thresholdFoldChangeUpdate) || (d / y > c.m.thresholdFoldChangeUpdate))))
                    return true;
            }
        }
    }
    return false;
}

// mergeVectors without the likelihood: L1, L2 -> L.m / L.maux.  Returns the number of entries (nAux: aux doubles), -1 for
// the reference's None, -2 for a fatal state -- whatever the FIRST failing step of the walk returns.
template <bool RV, bool U, bool SS>
__device__ inline int wave_merge(const Ctx<RV, U, SS> &c, ListRef L1, int n1, double bLen1, bool tip1, ListRef L2, int n2,
                                 double bLen2, bool tip2, bool upDown, WaveUpdLds &L, int &nAuxOut)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long *w1 = (const unsigned long long *)L1.w, *w2 = (const unsigned long long *)L2.w;
    unsigned long long *A = L.in, *B = L.in + MAPLE_WU_IN;
    for (int i = lane; i < n1; i += 64) A[i] = w1[i];
    for (int i = lane; i < n2; i += 64) B[i] = w2[i];
    wave_sync();
    const int nSteps = n1 + n2;
    int nOut = 0, nAux = 0;
    for (int base = 0; base < nSteps; base += 64) {
        const int d = base + lane;
        int i = 0, k = 0;
        const bool seg = d < nSteps && merge_path(A, n1, B, n2, d, i, k);
        int st = 0;
        uint2 lw[1];
        double la[6];
        Writer o;
        o.init(lw, la);
        if (seg) {
            const int pa = i > 0 ? (int)(uint32_t)A[i - 1] : 0, pb = k > 0 ? (int)(uint32_t)B[k - 1] : 0;
            Ent e1, e2;
            decode_word(A[i], L1.aux, e1);
            decode_word(B[k], L2.aux, e2);
            st = merge_step(c, e1, e2, max(pa, pb), bLen1, tip1, bLen2, tip2, upDown, o);
        }
        const unsigned long long bad = __ballot(st != 0);
        if (bad) return __shfl(st, __ffsll((long long)bad) - 1);
        const unsigned long long segs = __ballot(seg);
        const int na = seg ? o.na : 0;
        int total;
        const int ao = nAux + wave_excl_sum(na, lane, total);
        if (seg) {
            const int idx = nOut + __popcll(segs & ((1ull << lane) - 1ull));
            L.m[idx] = (unsigned long long)lw[0].x | ((unsigned long long)((lw[0].y & 0xFFu) | ((uint32_t)ao << 8)) << 32);
            for (int j = 0; j < na; j++) L.maux[ao + j] = la[j];
        }
        nOut += __popcll(segs);
        nAux += total;
    }
    wave_sync();
    nAuxOut = nAux;
    return nOut;
}

// shorten: the n entries of L.m -> the list at (gw, gaux), its words also in L.in and its aux in L.baux.  A run of R
// entries of one kind collapses to its last entry; whether an entry with distances joins the run is decided against the
// run's FIRST entry with a tolerance, so those (few) are decided in order.
template <class C> __device__ inline int wave_shorten(const C &c, WaveUpdLds &L, int n, uint2 *gw, double *gaux, int &nAuxOut)
{
    const int lane = threadIdx.x & 63;
    const double thr = c.m.thresholdProb;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        int mark = 0;
        if (k < n && k > 0) {
            const uint32_t m1 = (uint32_t)(L.m[k] >> 32), m0 = (uint32_t)(L.m[k - 1] >> 32);
            if ((m1 & 7u) == 4u && (m0 & 7u) == 4u && ((m1 ^ m0) & 0x60u) == 0u) mark = (m1 & 0x20u) ? 2 : 1;
        }
        if (k < n) L.mark[k] = mark;
        const unsigned long long sm = __ballot(mark == 2);
        if (lane == 0) L.seq[base >> 6] = sm;
    }
    wave_sync();
    {
        int head = -1, prevK = -2;
        bool prevAbs = false;
        for (int ch = 0; ch * 64 < n; ch++) {
            unsigned long long sm = L.seq[ch];
            while (sm) {
                const int k = ch * 64 + __ffsll((long long)sm) - 1;
                sm &= sm - 1;
                if (!(prevK == k - 1 && prevAbs)) head = k - 1;            // the run's first entry (the reference's entryOld)
                Ent nw, hd;
                decode_word(L.m[k], L.maux, nw);
                decode_word(L.m[head], L.maux, hd);
                bool absorb;
                if (fabs(nw.d0 - hd.d0) > thr) absorb = false;
                else if (nw.hasD1 && fabs(nw.d1 - hd.d1) > thr) absorb = false;
                else absorb = (nw.flag == hd.flag);
                if (lane == 0) L.mark[k] = absorb ? 1 : 0;
                prevK = k; prevAbs = absorb;
            }
        }
    }
    wave_sync();
    int nOut = 0, nAux = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const bool keep = j < n && (j == n - 1 || L.mark[j + 1] != 1);
        const unsigned long long w = keep ? L.m[j] : 0ull;
        const uint32_t meta = (uint32_t)(w >> 32);
        const int na = keep ? aux_count(meta) : 0;
        const unsigned long long kept = __ballot(keep);
        int total;
        const int ao = nAux + wave_excl_sum(na, lane, total);
        if (keep) {
            const int idx = nOut + __popcll(kept & ((1ull << lane) - 1ull));
            const uint32_t nm = (meta & 0xFFu) | ((uint32_t)ao << 8);
            L.in[idx] = (w & 0xFFFFFFFFull) | ((unsigned long long)nm << 32);
            gw[idx] = make_uint2((uint32_t)w, nm);
            const int src = (int)(meta >> 8);
            for (int t = 0; t < na; t++) { const double v = L.maux[src + t]; L.baux[ao + t] = v; gaux[ao + t] = v; }
        }
        nOut += __popcll(kept);
        nAux += total;
    }
    wave_sync();
    nAuxOut = nAux;
    return nOut;
}

// areVectorsDifferent(A, B): both lists' words in LDS
template <class C>
__device__ inline bool wave_differ(const C &c, const unsigned long long *A, const double *auxA, int nA, const unsigned long long *B,
                                   const double *auxB, int nB)
{
    const int lane = threadIdx.x & 63;
    const int nSteps = nA + nB;
    for (int base = 0; base < nSteps; base += 64) {
        const int d = base + lane;
        int i = 0, k = 0;
        const bool seg = d < nSteps && merge_path(A, nA, B, nB, d, i, k);
        bool diff = false;
        if (seg) {
            Ent e1, e2;
            decode_word(A[i], auxA, e1);
            decode_word(B[k], auxB, e2);
            diff = differ_step(c, e1, e2);
        }
        if (__ballot(diff)) return true;
    }
    return false;
}

// estimateBranchLengthWithDerivative (M:5040-5358) by the wavefront: lane d evaluates step d of the walk over the parent-side
// list P and the child-side list Cl (blen_step_body.inc, the text the one-lane walk runs); what the steps add to the constant
// c1 is added in walk order (the sum is rounded term by term), the 1/(a_i+t) terms are compacted in walk order into L.f, and
// the bracketing and bisection (blen_solve_body.inc) then run on them as they do for one lane.  `terms`: 128 doubles of LDS.
template <bool RV, bool U, bool SS>
__device__ inline double wave_blen(const Ctx<RV, U, SS> &c, ListRef P, int nP, ListRef Cl, int nC, bool fromTipC, WaveLds &L,
                                   double *terms, bool *isFalse)
{
    const int lane = threadIdx.x & 63;
    const double *rf = c.rf;
    const double *cr = c.m.cumulativeRate;
    const unsigned long long *pw = (const unsigned long long *)P.w, *cw = (const unsigned long long *)Cl.w;
    for (int i = lane; i < nP; i += 64) L.a[i] = pw[i];
    for (int i = lane; i < nC; i += 64) L.b[i] = cw[i];
    wave_sync();
    const int nSteps = nP + nC;
    int nA = 0, nZeros = 0;
    double c1 = c.m.globalTotRate;
    *isFalse = false;
    for (int base = 0; base < nSteps; base += 64) {
        const int d = base + lane;
        int i = 0, k = 0;
        const bool seg = d < nSteps && merge_path(L.a, nP, L.b, nC, d, i, k);
        int nTerm = 0;
        double t0 = 0.0, t1 = 0.0, aiv = 0.0;
        bool hasAi = false, isZero = false;
        if (seg) {
            int pos = max(i > 0 ? (int)(uint32_t)L.a[i - 1] : 0, k > 0 ? (int)(uint32_t)L.b[k - 1] : 0);
            Ent e1, e2;
            decode_word(L.a[i], P.aux, e1);
            decode_word(L.b[k], Cl.aux, e2);
#define BLEN_C1_ADD(x) do { const double v_ = (x); if (nTerm == 0) t0 = v_; else t1 = v_; ++nTerm; } while (0)
#define BLEN_C1_SUB(x) do { const double v_ = -(x); if (nTerm == 0) t0 = v_; else t1 = v_; ++nTerm; } while (0)
#define BLEN_AIS(v) do { aiv = (v); hasAi = true; } while (0)
#define BLEN_ZERO() isZero = true
#include "blen_step_body.inc"
#undef BLEN_C1_ADD
#undef BLEN_C1_SUB
#undef BLEN_AIS
#undef BLEN_ZERO
            (void)pos;
        }
        int total;
        const int to = wave_excl_sum(nTerm, lane, total);
        if (nTerm > 0) terms[to] = t0;
        if (nTerm > 1) terms[to + 1] = t1;
        const unsigned long long am = __ballot(hasAi);
        if (hasAi) L.f[nA + __popcll(am & ((1ull << lane) - 1ull))] = aiv;
        nA += __popcll(am);
        nZeros += __popcll(__ballot(isZero));
        wave_sync();
        for (int j = 0; j < total; j++) c1 += terms[j];                    // (C - x is C + (-x), bit for bit)
        wave_sync();
    }
    const double *ais = L.f;
    constexpr int stride = 1;
#include "blen_solve_body.inc"
}

}  // namespace maple
