// maple_amd/csrc/append_lds.h -- appendProbNode (MAPLEv0.7.5.4.py:6505-6785) with the genome lists behind memory accessors.
//
// The batch-scoring kernel is bound by the vector-memory address path, not by HBM or the VALU: one lane walks one
// (candidate list, query list) pair, so every load of a list word or of a stored branch length touches a cache line of its
// own (rocprofv3: 285 vector-memory instructions per 64 pairs, waves parked on s_waitcnt 62 % of their life,
// profiles/r02_100k_ratevar.md).  A tile's 64 candidate lists are shared by EVERY query, so k_append_queries_lds stages
// them in LDS once per workgroup with coalesced loads and lets 8 wavefronts walk them from there for a block of queries.
// For that the walk reads its lists through small accessor types, so that LDS-resident lists are read with ds_read
// (address space 3) instead of flat loads.  The arithmetic is the PairWalk of genome_dev.h, operation for operation: the
// result is bit-identical.
#pragma once
#include "genome_dev.h"

namespace maple {

typedef __attribute__((address_space(3))) const unsigned long long *lds_u64p;
typedef __attribute__((address_space(3))) const double *lds_f64p;

// rate(i, site, c): the per-site rate multiplier at the site entry i ends on (a site that needs work is always the last
// -- the only -- site of a single-site entry of one of the two lists).  Staged lists carry it next to their words, gathered
// once at staging time: in the walk the gather from the 240 KB siteRates table was the longest wait of a step.
struct MemG {                          // words and aux doubles in global memory (or anywhere, through generic pointers)
    const unsigned long long *w;
    const double *a;
    __device__ __forceinline__ unsigned long long word(int i) const { return w[i]; }
    __device__ __forceinline__ double aux(uint32_t off) const { return a[off]; }
    template <class C> __device__ __forceinline__ double rate(int, int site, const C &c) const { return c.rate(site); }
};
struct MemL {                          // words, aux doubles and per-entry site rates in LDS
    lds_u64p w;
    lds_f64p a;
    lds_f64p r;
    __device__ __forceinline__ unsigned long long word(int i) const { return w[i]; }
    __device__ __forceinline__ double aux(uint32_t off) const { return a[off]; }
    template <class C> __device__ __forceinline__ double rate(int i, int, const C &) const { return r[i]; }
};
struct MemLG {                         // words and site rates in LDS, aux doubles in global memory (the staged query list)
    lds_u64p w;
    const double *a;
    lds_f64p r;
    __device__ __forceinline__ unsigned long long word(int i) const { return w[i]; }
    __device__ __forceinline__ double aux(uint32_t off) const { return a[off]; }
    template <class C> __device__ __forceinline__ double rate(int i, int, const C &) const { return r[i]; }
};

struct EntV {                          // a decoded entry with its payload in registers
    int pos, type, ref;
    bool hasD0, hasD1, flag;
    double d0, d1;
    double v[4];                       // type 6
};

template <class M> __device__ __forceinline__ void decode_v(unsigned long long w, const M &mem, EntV &e)
{
    const uint32_t meta = (uint32_t)(w >> 32);
    e.pos = (int)(uint32_t)w;
    e.type = meta & 7u;
    e.ref = (meta >> 3) & 3u;
    e.hasD0 = meta & (1u << 5);
    e.hasD1 = meta & (1u << 6);
    e.flag = meta & (1u << 7);
    uint32_t off = meta >> 8;
    e.d0 = 0.0; e.d1 = 0.0;
    if (e.hasD0) { e.d0 = mem.aux(off); off++; }
    if (e.hasD1) { e.d1 = mem.aux(off); off++; }
    if (e.type == 6) { e.v[0] = mem.aux(off); e.v[1] = mem.aux(off + 1); e.v[2] = mem.aux(off + 2); e.v[3] = mem.aux(off + 3); }
}

// site_factor of genome_dev.h on register-resident entries (M:6611-6761)
template <bool RV, bool U, bool SS>
__device__ __forceinline__ double site_factor_v(const Ctx<RV, U, SS> &c, const EntV &e1, const EntV &e2, int site, double r,
                                                bool isTipC, double bLen)
{
    typedef Ctx<RV, U, SS> CT;
    const double *rf = c.rf;
    double cl = bLen;                                                   // M:6586-6599
    if (e1.type < 5) { if (e1.hasD1) cl += e1.d1; else if (e1.hasD0) cl += e1.d0; }
    else if (e1.hasD0) cl += e1.d0;
    if (e2.type < 5) { if (e2.hasD0 && !e2.hasD1) cl += e2.d0; }
    else if (e2.hasD0) cl += e2.d0;
    const bool flag1 = U && e1.type < 5 && e1.hasD0 && e1.flag;
    const bool flag2 = U && e2.type < 5 && (isTipC || (e2.hasD0 && e2.flag));
    double f;
    if (e1.type == 6 && e2.type == 6) {                                 // M:6677-6686
        double t3[4];
        gpv_vec(c, r, e2.v, cl, false, t3);
        double tot = 0.0;
        for (int j = 0; j < 4; j++) tot += e1.v[j] * t3[j];
        f = tot;
    } else if (e1.type == 6) {                                          // O over nucleotide/R, M:6687-6703
        int i2 = (e2.type == 4) ? e1.ref : e2.type;
        double p = sel4(e1.v, i2);
        if (p > 0.02) f = p;
        else {
            double t3[4];
            gpv_nuc<CT, U>(c, r, i2, cl, flag2 ? c.err(site) : 0.0, false, flag2, t3);
            double tot = 0.0;
            for (int j = 0; j < 4; j++) tot += e1.v[j] * t3[j];
            f = tot;
        }
    } else if (e2.type == 6) {                                          // nucleotide/R over O, M:6611-6633, 6744-6761
        int i1 = (e1.type == 4) ? e2.ref : e1.type;
        double p = sel4(e2.v, i1);
        if (p > 0.02) f = p;
        else if (e1.hasD1) {
            double t2[4], t3[4];
            gpv_vec(c, r, e2.v, cl, false, t3);
            gpv_nuc<CT, U>(c, r, i1, e1.d0, c.err(site), false, flag1, t2);
            double tot = 0.0;
            if (e1.type == 4) { for (int i = 0; i < 4; i++) tot += t3[i] * t2[i] * rf[i]; }
            else { for (int i = 0; i < 4; i++) tot += t2[i] * t3[i] * rf[i]; }
            f = tot / rf[i1];
        } else if (cl != 0.0) {
            double t3[4];
            gpv_vec(c, r, e2.v, cl, false, t3);
            f = sel4(t3, i1);
        } else f = p;
    } else {                                                            // two nucleotides, observation beyond the root on the
        int i1 = (e1.type == 4) ? e2.ref : e1.type;                     // parent side (M:6644-6654, 6726-6733)
        int i2 = (e2.type == 4) ? e1.ref : e2.type;
        double t2[4], t3[4];
        double er = c.err(site);
        gpv_nuc<CT, U>(c, r, i2, cl, er, false, flag2, t3);
        gpv_nuc<CT, U>(c, r, i1, e1.d0, er, false, flag1, t2);
        double tot = 0.0;
        if (e1.type == 4) { for (int i = 0; i < 4; i++) tot += t3[i] * t2[i] * rf[i]; }
        else { for (int j = 0; j < 4; j++) tot += rf[j] * t3[j] * t2[j]; }
        f = tot / rf[i1];
    }
    return f;
}

// PairWalk of genome_dev.h over accessors: P = parent (candidate) list, C = child (query) list
template <bool RV, bool U, bool SS, class PM, class CM>
__device__ __forceinline__ double append_walk_m(const Ctx<RV, U, SS> &c, const PM &P, const CM &C, bool isTipC, double bLen)
{
    const int lRef = c.m.lRef;
    const double carry = c.m.minimumCarryOver;
    unsigned long long wa = P.word(0), wb = C.word(0);
    int ia = 0, ib = 0;
    double tf = 1.0;
    double Lk = bLen * c.m.globalTotRate;                               // M:6541
    if (U && isTipC) Lk += c.m.totError;                                // M:6542-6543
    double carry1 = 1.0, carry2 = 1.0;
    int nCarry = 0;
    constexpr unsigned long long WORK = work_table();
    for (;;) {
#ifdef MAPLE_WALK_GATHER                                                     // (measured on the 1 000 000-tip leg: 8 % slower)
        // (a lane first runs ahead over the steps that need no work and stops at its next site: the per-site arithmetic below is
        // then executed once for all lanes of the wavefront that have a site, not at every step for the few that do -- see
        // PairWalk::run, genome_dev.h.  Same steps in the same order per lane.)
        bool atEnd = false;
        for (;;) {
            const int pa0 = (int)(uint32_t)wa, pb0 = (int)(uint32_t)wb;
            if ((WORK >> ((int)((wa >> 32) & 7ull) * 8 + (int)((wb >> 32) & 7ull))) & 1ull) break;
            const int pos0 = min(pa0, pb0);
            if (pos0 == lRef) { atEnd = true; break; }
            if (pa0 == pos0) { ++ia; wa = P.word(ia); }
            if (pb0 == pos0) { ++ib; wb = C.word(ib); }
        }
        if (atEnd) break;
#endif
        const int pa = (int)(uint32_t)wa, pb = (int)(uint32_t)wb;
        const uint32_t m1 = (uint32_t)(wa >> 32), m2 = (uint32_t)(wb >> 32);
        const int t1 = m1 & 7u, t2 = m2 & 7u;
        const int pos = min(pa, pb);
#ifndef MAPLE_WALK_OLD
        if ((t1 != 5) & (t2 != 5) & ((t1 != t2) | (t1 == 6))) {         // (work_table() in 32-bit compares)
#else
        if ((WORK >> (t1 * 8 + t2)) & 1ull) {
#endif
            const int site = pos - 1;
            bool dead = false;
            const double r = RV ? ((pa == pos) ? P.rate(ia, site, c) : C.rate(ib, site, c)) : 1.0;
            if (t1 == 6 || t2 == 6 || (m1 & (1u << 6))) {               // O vector or observation beyond the root
#ifndef MAPLE_WALK_OLD
                // (one O vector against a nucleotide / R: the vector's entry for that nucleotide first -- above 0.02 it IS the factor,
                // M:6615 / 6692 / 6746, five of six such sites on the bench trees: one read instead of two decoded entries)
                double f = 0.0;
                bool general = true;
                if ((t1 == 6) != (t2 == 6)) {
                    const bool o1 = t1 == 6;
                    const uint32_t mo = o1 ? m1 : m2;
                    const int tn = o1 ? t2 : t1;
                    const uint32_t off = (mo >> 8) + ((mo >> 5) & 1u) + ((mo >> 6) & 1u) + (uint32_t)((tn == 4) ? (int)((mo >> 3) & 3u) : tn);
                    f = o1 ? P.aux(off) : C.aux(off);
                    general = !(f > 0.02);
                }
                if (general) {
                    EntV e1, e2;
                    decode_v(wa, P, e1);
                    decode_v(wb, C, e2);
                    f = site_factor_v(c, e1, e2, site, r, isTipC, bLen);
                }
                tf *= f;
#else
                EntV e1, e2;
                decode_v(wa, P, e1);
                decode_v(wb, C, e2);
                tf *= site_factor_v(c, e1, e2, site, r, isTipC, bLen);
#endif
            } else {                                                     // two different nucleotides (R = the reference one)
                double cl = bLen;                                        // M:6640-6668, 6713-6742
                if (m1 & (1u << 5)) cl += P.aux(m1 >> 8);
                if ((m2 & (1u << 5)) && !(m2 & (1u << 6))) cl += C.aux(m2 >> 8);
                const int i1 = (t1 == 4) ? (int)((m2 >> 3) & 3u) : t1;
                const int i2 = (t2 == 4) ? (int)((m1 >> 3) & 3u) : t2;
                const double qv = c.q(r, i1, i2);
                double f = fmin_py(0.25, qv * cl);
                if (U) {
                    const bool flag1 = (t1 != 4) && (m1 & (1u << 5)) && (m1 & (1u << 7));
                    const bool flag2 = isTipC || ((m2 & (1u << 5)) && (m2 & (1u << 7)));
                    if (t1 == 4) { if (flag2) f += c.err(site) * 0.33333; else if (cl == 0.0) dead = true; }
                    else if (flag1 || flag2) f += (double)((int)flag1 + (int)flag2) * 0.33333 * c.err(site);
                    else if (cl == 0.0) dead = true;
                } else if (cl == 0.0) dead = true;                       // zero-length mismatch: -inf (M:6663, 6742)
                tf *= f;
            }
            if (dead) return -INFINITY;
            if (tf <= carry) {                                           // M:6772-6783
                if (tf < 2.2250738585072014e-308) return -INFINITY;
                if (nCarry == 2) { Lk += log(carry1); carry1 = carry2; nCarry = 1; }
                if (nCarry == 0) carry1 = tf; else carry2 = tf;
                ++nCarry;
                tf = 1.0;
            }
        }
        if (pos == lRef) break;
#ifdef MAPLE_WALK_BRANCHFREE
        // both cursors reloaded every step (a list always ends at lRef, so the index never runs past its last entry): two
        // reads that are cheap when the lists sit in LDS, instead of two exec-mask branches on the scalar unit
        ia += (pa == pos) ? 1 : 0;
        ib += (pb == pos) ? 1 : 0;
        wa = P.word(ia);
        wb = C.word(ib);
#else
        if (pa == pos) { ++ia; wa = P.word(ia); }
        if (pb == pos) { ++ib; wb = C.word(ib); }
#endif
    }
    double lk = Lk;
    if (nCarry >= 1) lk += log(carry1);
    if (nCarry >= 2) lk += log(carry2);
    return (tf > 0.0) ? lk + log(tf) : -INFINITY;
}

// ---- the same walk over lists whose plain reference runs may be left out ------------------------------------------------------
// Half the entries of a genome list are reference runs without a tail, and two of three steps of the walk above end on the
// boundary of one and do nothing.  A list in the SKIPPING form leaves out every tail-less R entry that is followed by a
// single-site entry (types 0-3, 6): what lies between the entry before and that single-site entry is then reference by
// omission.  Kept are all single-site entries, N runs, R runs with a tail, the list's last entry (so that the walk still ends on
// lRef), and a tail-less R that is followed by a run -- so a run's first position is always the position after the entry
// before it.  (A list with nothing left out is a list in skipping form too.)  The walk goes from event to event: where both
// lists are reference by omission nothing is looked at.  The sites that need work, their entries and the order of the factors
// are those of append_walk_m: bit-identical.
// keep(w, next): may entry w be left out?  No if it is the last entry (next == 0).
__device__ __forceinline__ bool skip_form_drops(unsigned long long w, unsigned long long next, bool last)
{
    const uint32_t m = (uint32_t)(w >> 32), t = m & 7u, tn = (uint32_t)(next >> 32) & 7u;
    return !last && t == 4u && !(m & 0x60u) && (tn < 4u || tn == 6u);
}
// (Tried on top of it and not kept: the lanes of a wavefront taking the general site_factor TOGETHER -- a lane that reaches such
// a site waits, the path runs when a dozen lanes wait or nobody else can go on.  Bit-identical, and 1.7 x SLOWER on the
// 100 000-tip tree, 512 queries x every branch: 56.7 ms against 32.8 -- the waiting lanes lengthen every wavefront's walk by
// more than the shared path saves, and the state kept across the vote costs another 100 spilled registers.)
template <bool RV, bool U, bool SS, class PM, class CM>
__device__ __forceinline__ double append_walk_c(const Ctx<RV, U, SS> &c, const PM &P, const CM &C, bool isTipC, double bLen, bool valid = true)
{
    if (!valid) return -INFINITY;
    const int lRef = c.m.lRef;
    const double carry = c.m.minimumCarryOver;
    unsigned long long wa = P.word(0), wb = C.word(0);
    int ia = 0, ib = 0, cur = 0;
    double tf = 1.0;
    double Lk = bLen * c.m.globalTotRate;                               // M:6541
    if (U && isTipC) Lk += c.m.totError;                                // M:6542-6543
    double carry1 = 1.0, carry2 = 1.0;
    int nCarry = 0;
    for (;;) {
        const int pa = (int)(uint32_t)wa, pb = (int)(uint32_t)wb;
        const uint32_t ma = (uint32_t)(wa >> 32), mb = (uint32_t)(wb >> 32);
        const int ta = ma & 7u, tb = mb & 7u;
        const bool sa = ta < 4 || ta == 6, sb = tb < 4 || tb == 6;       // single-site entries
        // a run reaches back to the position after the entry before it: it covers cur + 1; a single-site entry covers cur + 1 only
        // if it IS there -- else reference by omission up to the site before it
        bool realA = !sa || pa == cur + 1, realB = !sb || pb == cur + 1;
        if (!realA && !realB) {                                          // both reference by omission: on to the next entry of either
            cur = min(pa, pb) - 1;
            realA = pa == cur + 1; realB = pb == cur + 1;
        }
        const int endA = realA ? pa : pa - 1, endB = realB ? pb : pb - 1;
        const int pos = min(endA, endB);
        const uint32_t m1 = realA ? ma : 4u, m2 = realB ? mb : 4u;
        const int t1 = m1 & 7u, t2 = m2 & 7u;
        if ((t1 != 5) & (t2 != 5) & ((t1 != t2) | (t1 == 6))) {
            const int site = pos - 1;
            bool dead = false;
            const double r = RV ? ((realA && pa == pos) ? P.rate(ia, site, c) : C.rate(ib, site, c)) : 1.0;
            if (t1 == 6 || t2 == 6 || (m1 & (1u << 6))) {               // O vector or observation beyond the root
                double f = 0.0;
                bool general = true;
                if ((t1 == 6) != (t2 == 6)) {                            // (the reference's shortcut first, see append_walk_m)
                    const bool o1 = t1 == 6;
                    const uint32_t mo = o1 ? m1 : m2;
                    const int tn = o1 ? t2 : t1;
                    const uint32_t off = (mo >> 8) + ((mo >> 5) & 1u) + ((mo >> 6) & 1u) + (uint32_t)((tn == 4) ? (int)((mo >> 3) & 3u) : tn);
                    f = o1 ? P.aux(off) : C.aux(off);
                    general = !(f > 0.02);
                }
                if (general) {
                    EntV e1, e2;
                    decode_v(realA ? wa : (((unsigned long long)4u << 32) | (uint32_t)pos), P, e1);
                    decode_v(realB ? wb : (((unsigned long long)4u << 32) | (uint32_t)pos), C, e2);
                    f = site_factor_v(c, e1, e2, site, r, isTipC, bLen);
                }
                tf *= f;
            } else {                                                     // two different nucleotides (R = the reference one)
                double cl = bLen;                                        // M:6640-6668, 6713-6742
                if (m1 & (1u << 5)) cl += P.aux(m1 >> 8);
                if ((m2 & (1u << 5)) && !(m2 & (1u << 6))) cl += C.aux(m2 >> 8);
                const int i1 = (t1 == 4) ? (int)((m2 >> 3) & 3u) : t1;
                const int i2 = (t2 == 4) ? (int)((m1 >> 3) & 3u) : t2;
                const double qv = c.q(r, i1, i2);
                double f = fmin_py(0.25, qv * cl);
                if (U) {
                    const bool flag1 = (t1 != 4) && (m1 & (1u << 5)) && (m1 & (1u << 7));
                    const bool flag2 = isTipC || ((m2 & (1u << 5)) && (m2 & (1u << 7)));
                    if (t1 == 4) { if (flag2) f += c.err(site) * 0.33333; else if (cl == 0.0) dead = true; }
                    else if (flag1 || flag2) f += (double)((int)flag1 + (int)flag2) * 0.33333 * c.err(site);
                    else if (cl == 0.0) dead = true;
                } else if (cl == 0.0) dead = true;                       // zero-length mismatch: -inf (M:6663, 6742)
                tf *= f;
            }
            if (dead) return -INFINITY;
            if (tf <= carry) {                                           // M:6772-6783
                if (tf < 2.2250738585072014e-308) return -INFINITY;
                if (nCarry == 2) { Lk += log(carry1); carry1 = carry2; nCarry = 1; }
                if (nCarry == 0) carry1 = tf; else carry2 = tf;
                ++nCarry;
                tf = 1.0;
            }
        }
        if (pos == lRef) break;
        cur = pos;
        if (realA && pa == pos) { ++ia; wa = P.word(ia); }
        if (realB && pb == pos) { ++ib; wb = C.word(ib); }
    }
    double lk = Lk;
    if (nCarry >= 1) lk += log(carry1);
    if (nCarry >= 2) lk += log(carry2);
    return (tf > 0.0) ? lk + log(tf) : -INFINITY;
}

}  // namespace maple
