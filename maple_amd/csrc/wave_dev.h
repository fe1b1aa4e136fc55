// maple_amd/csrc/wave_dev.h -- appendProbNode (MAPLEv0.7.5.4.py:6505-6785) by a whole wavefront.
//
// One lane walking two lists is a chain of ~100 dependent steps (PairWalk, genome_dev.h): 70-200 us for ONE placement score,
// whatever else the GPU is doing.  The steps only depend on each other through the running product.  So here the two-list
// walk is cut along its merge path: step d of the walk -- "the list whose entry ends first advances" -- is found by lane d
// with a binary search over the two lists' end positions (both staged in LDS), every lane evaluates the site factor of
// ITS step with the code of PairWalk::step, the factors of the steps that need work are written to LDS in walk order, and
// the running product -- with the reference's rescaling rule (M:6772-6783), a few instructions per factor -- is then taken
// over them in that same order.  Same factors, same order of multiplications and logarithms: the result is the walk's, bit
// for bit.  Every lane of the wavefront must call it with the same arguments; every lane gets the result.
#pragma once
#include "genome_dev.h"

namespace maple {

#ifndef MAPLE_WAVE_CAPW
#define MAPLE_WAVE_CAPW 256            // entries per list the cooperative walk stages (longer lists: one lane's walk)
#endif

struct WaveLds {                       // per wavefront
    unsigned long long a[MAPLE_WAVE_CAPW], b[MAPLE_WAVE_CAPW];
    double f[2 * MAPLE_WAVE_CAPW];
};

template <bool RV, bool U, bool SS>
__device__ inline double wave_append(const Ctx<RV, U, SS> &c, ListRef P, int nP, ListRef Cl, int nC, bool isTipC, double bLen,
                                     WaveLds &L)
{
    const int lane = threadIdx.x & 63;
    if (nP > MAPLE_WAVE_CAPW || nC > MAPLE_WAVE_CAPW) return append_walk(c, P, Cl, isTipC, bLen);   // (every lane, same result)
    const unsigned long long *pw = (const unsigned long long *)P.w, *cw = (const unsigned long long *)Cl.w;
    for (int i = lane; i < nP; i += 64) L.a[i] = pw[i];
    for (int i = lane; i < nC; i += 64) L.b[i] = cw[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int lRef = c.m.lRef;
    const int nSteps = nP + nC;
    constexpr unsigned long long WORK = work_table();
    int nWork = 0;
    bool anyDead = false;
    for (int base = 0; base < nSteps; base += 64) {
        const int d = base + lane;
        bool work = false, dead = false;
        double f = 1.0;
        if (d < nSteps) {
            // merge path: i = entries of P consumed before step d (P first on ties), k = those of the child list
            int lo = max(0, d - nC), hi = min(d, nP);
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((uint32_t)L.a[mid] <= (uint32_t)L.b[d - 1 - mid]) lo = mid + 1; else hi = mid;
            }
            const int i = lo, k = d - lo;
            bool seg;
            if (i < nP && (k >= nC || (uint32_t)L.a[i] <= (uint32_t)L.b[k])) seg = k < nC;          // this step ends P's entry i
            else seg = i < nP && !(i > 0 && (uint32_t)L.a[i - 1] == (uint32_t)L.b[k]);              // ... the child's entry k (not a tie's second half)
            if (seg) {
                const unsigned long long wa = L.a[i], wb = L.b[k];
                const int pa = (int)(uint32_t)wa, pb = (int)(uint32_t)wb;
                const uint32_t m1 = (uint32_t)(wa >> 32), m2 = (uint32_t)(wb >> 32);
                const int t1 = m1 & 7u, t2 = m2 & 7u;
                const int pos = min(pa, pb);
                if ((WORK >> (t1 * 8 + t2)) & 1ull) {                     // PairWalk::step, the part of one segment
                    work = true;
                    const int site = pos - 1;
                    if (t1 == 6 || t2 == 6 || (m1 & (1u << 6))) {
                        Ent e1, e2;
                        decode_word(wa, P.aux, e1);
                        decode_word(wb, Cl.aux, e2);
                        if (site_factor(c, e1, e2, site, isTipC, bLen, &f) == 1) dead = true;
                    } else {
                        double cl = bLen;                                // M:6640-6668, 6713-6742
                        if (m1 & (1u << 5)) cl += P.aux[m1 >> 8];
                        if ((m2 & (1u << 5)) && !(m2 & (1u << 6))) cl += Cl.aux[m2 >> 8];
                        const int i1 = (t1 == 4) ? (int)((m2 >> 3) & 3u) : t1;
                        const int i2 = (t2 == 4) ? (int)((m1 >> 3) & 3u) : t2;
                        const double qv = c.q(c.rate(site), i1, i2);
                        f = fmin_py(0.25, qv * cl);
                        if (U) {
                            const bool flag1 = (t1 != 4) && (m1 & (1u << 5)) && (m1 & (1u << 7));
                            const bool flag2 = isTipC || ((m2 & (1u << 5)) && (m2 & (1u << 7)));
                            if (t1 == 4) { if (flag2) f += c.err(site) * 0.33333; else if (cl == 0.0) dead = true; }
                            else if (flag1 || flag2) f += (double)((int)flag1 + (int)flag2) * 0.33333 * c.err(site);
                            else if (cl == 0.0) dead = true;
                        } else if (cl == 0.0) dead = true;               // zero-length mismatch: -inf (M:6663, 6742)
                    }
                }
                (void)lRef;
            }
        }
        const unsigned long long wm = __ballot(work);
        if (__ballot(dead)) anyDead = true;
        if (work) L.f[nWork + __popcll(wm & ((1ull << lane) - 1ull))] = f;
        nWork += __popcll(wm);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (anyDead) return -INFINITY;                                       // the walk would have stopped at that step (or earlier: -inf too)
    // the running product in walk order (every lane the same)
    const double carry = c.m.minimumCarryOver;
    double tf = 1.0, Lk = bLen * c.m.globalTotRate;                      // M:6541
    if (U && isTipC) Lk += c.m.totError;                                 // M:6542-6543
    double carry1 = 1.0, carry2 = 1.0;
    int nCarry = 0;
    for (int j = 0; j < nWork; j++) {
        tf *= L.f[j];
        if (tf <= carry) {                                               // M:6772-6783
            if (tf < 2.2250738585072014e-308) return -INFINITY;
            if (nCarry == 2) { Lk += log(carry1); carry1 = carry2; nCarry = 1; }
            if (nCarry == 0) carry1 = tf; else carry2 = tf;
            ++nCarry;
            tf = 1.0;
        }
    }
    double lk = Lk;
    if (nCarry >= 1) lk += log(carry1);
    if (nCarry >= 2) lk += log(carry2);
    return (tf > 0.0) ? lk + log(tf) : -INFINITY;
}

}  // namespace maple
