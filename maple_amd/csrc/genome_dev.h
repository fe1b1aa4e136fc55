// maple_amd/csrc/genome_dev.h -- gfx950 device functions for MAPLE genome lists.
//
// Packed CSR genome lists (see include/maple_hip.h) are walked by ONE LANE per
// (parent, child) pair; 64 independent walks per wavefront.  The arithmetic
// follows MAPLE v0.7.5.4 (reference MAPLEv0.7.5.4.py, cited as M:<line>) in
// operand order so that fp64 results are reproducible; compile with
// -ffp-contract=off.  The 4x4 rate matrix and root frequencies live in LDS
// (dynamic per-lane indexing), per-site rate / error vectors are gathered
// through L2 only at non-reference sites.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace maple {

// ---- model ---------------------------------------------------------------------
struct DevModel {                      // kernel argument (uniform -> SGPRs)
    int32_t lRef;
    int32_t useRateVariation, usingErrorRate, errorRateSiteSpecific;
    const double *siteRates;           // [lRef] or null
    const double *errorRates;          // [lRef] or null
    const double *cumulativeRate;      // [lRef+1]
    const double *cumulativeErrorRate; // [lRef+1] or null
    double Q[16];
    double rootFreqs[4];
    double errorRate, totError, globalTotRate, minimumCarryOver;
    double thresholdProb, thresholdProb4, minBLenSensitivity, thresholdDiffForUpdate, thresholdFoldChangeUpdate;
    double defaultBLen;
    // tables of findProbRoot (M:4865-4912)
    double rootFreqsLog[4];
    const int32_t *cumulativeBases;    // [(lRef+1)*4], M:3669-3674
    const double *rootFreqsLogErrorCumulative;   // [lRef+1] when usingErrorRate, M:6373-6390
};

struct Lds {                           // per-workgroup staging of the tiny tables
    double Q[16];
    double rf[4];
};

__device__ inline void stage_model(const DevModel &m, Lds &s)
{
    if (threadIdx.x < 16) s.Q[threadIdx.x] = m.Q[threadIdx.x];
    if (threadIdx.x < 4) s.rf[threadIdx.x] = m.rootFreqs[threadIdx.x];
    __syncthreads();
}

typedef __attribute__((address_space(1))) const double *glb_f64p;
template <bool RV, bool U, bool SS> struct Ctx {
    const DevModel &m;
    const double *Q;                   // LDS
    const double *rf;                  // LDS
    // The per-site tables are gathered at every site that needs work.  Their base pointers are read from the model ONCE,
    // here: through `m` each gather was a scalar load of the pointer, a wait, and then a flat load of the value -- two
    // dependent memory latencies in the middle of a lane's walk.  (Address space 1: a global_load, not a flat one.)
    glb_f64p sr, er;
    double errorRate;
    __device__ Ctx(const DevModel &m_, const Lds &s)
        : m(m_), Q(s.Q), rf(s.rf), sr((glb_f64p)m_.siteRates), er((glb_f64p)m_.errorRates), errorRate(m_.errorRate) {}
    // site rate multiplier of mutMatrices[pos] = Q * siteRates[pos]  (M:6361-6366)
    __device__ inline double rate(int pos) const { return RV ? sr[pos] : 1.0; }
    __device__ inline double q(double r, int i, int j) const { return RV ? Q[i * 4 + j] * r : Q[i * 4 + j]; }
    // errorRate resolved like "if usingErrorRate and errorRateSiteSpecific: errorRate=errorRates[pos]"
    __device__ inline double err(int pos) const { return (U && SS) ? er[pos] : errorRate; }
};

// ---- packed lists -----------------------------------------------------------------
struct ListRef {
    const uint2 *w;                    // entry words {pos, meta}
    const double *aux;                 // this list's aux block
};

struct Ent {                           // decoded view of one entry
    int pos;                           // last covered position (1-based)
    int type;                          // 0-3 nuc, 4 R, 5 N, 6 O
    int ref;                           // local reference nucleotide (single-site entries)
    bool hasD0, hasD1, flag;
    double d0, d1;
    const double *vec;                 // type 6
    __device__ inline bool single() const { return type < 4 || type == 6; }
};

struct Cursor {
    ListRef l;
    int idx;
    Ent e;
    __device__ inline void load()
    {
        uint2 w = l.w[idx];
        e.pos = (int)w.x;
        uint32_t meta = w.y;
        e.type = meta & 7u;
        e.ref = (meta >> 3) & 3u;
        e.hasD0 = meta & (1u << 5);
        e.hasD1 = meta & (1u << 6);
        e.flag = meta & (1u << 7);
        const double *a = l.aux + (meta >> 8);
        e.d0 = 0.0; e.d1 = 0.0;
        if (e.hasD0) { e.d0 = *a++; }
        if (e.hasD1) { e.d1 = *a++; }
        e.vec = a;
    }
    __device__ inline void init(ListRef r) { l = r; idx = 0; load(); }
    __device__ inline void next() { ++idx; load(); }
    // common tail of every two-list walk (e.g. M:4841-4854): step past single-site entries
    // and past runs that end at `pos`
    __device__ inline void step(int pos) { if (e.single() || e.pos == pos) next(); }
};

struct Writer {                        // builds one packed list
    uint2 *w;
    double *aux;
    int n, na;
    __device__ inline void init(uint2 *w_, double *a_) { w = w_; aux = a_; n = 0; na = 0; }
    __device__ inline void put(int type, int pos, int ref, bool hasD0, double d0, bool hasD1, double d1, bool flag,
                               const double *vec)
    {
        uint32_t meta = (uint32_t)type | ((uint32_t)ref << 3) | (hasD0 ? 1u << 5 : 0u) | (hasD1 ? 1u << 6 : 0u)
                        | (flag ? 1u << 7 : 0u) | ((uint32_t)na << 8);
        w[n++] = make_uint2((uint32_t)pos, meta);
        if (hasD0) aux[na++] = d0;
        if (hasD1) aux[na++] = d1;
        if (type == 6) { aux[na] = vec[0]; aux[na + 1] = vec[1]; aux[na + 2] = vec[2]; aux[na + 3] = vec[3]; na += 4; }
    }
    __device__ inline void bare(int type, int pos, int ref) { put(type, pos, ref, false, 0, false, 0, false, nullptr); }
    __device__ inline void copy(const Ent &e, int type, int pos, int ref)
    {
        put(type, pos, ref, e.hasD0, e.d0, e.hasD1, e.d1, e.flag, e.vec);
    }
};

// ---- getPartialVec (M:4073-4141) -----------------------------------------------------
// O vector moved along a branch: v + t*(Q v) going down, v + t*(Q^T v) going up
template <class C> __device__ inline void gpv_vec(const C &c, double r, const double *v, double t, bool up, double *out)
{
    if (t == 0.0) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3]; return; }
    double nv[4];
    bool neg = false;
    for (int i = 0; i < 4; i++) {
        double tot = 0.0;
        for (int j = 0; j < 4; j++) tot += (up ? c.q(r, j, i) : c.q(r, i, j)) * v[j];
        tot *= t;
        tot += v[i];
        if (tot < 0) neg = true;
        nv[i] = tot;
    }
    for (int i = 0; i < 4; i++) out[i] = neg ? 0.25 : nv[i];
}
// one-hot nucleotide (optionally error-smeared, M:4109-4125) moved along a branch
template <class C, bool U>
__device__ inline void gpv_nuc(const C &c, double r, int nuc, double t, double errorRate, bool up, bool flag, double *out)
{
    if (U && flag) {
        double nv[4];
        double off = errorRate * 0.33333;
        for (int i = 0; i < 4; i++) nv[i] = (i == nuc) ? 1.0 - errorRate : off;
        if (t == 0.0) { for (int i = 0; i < 4; i++) out[i] = nv[i]; return; }
        double mv[4];
        bool neg = false;
        for (int j = 0; j < 4; j++) {
            double tot = 0.0;
            for (int i = 0; i < 4; i++) tot += c.q(r, j, i) * nv[i];
            tot *= t;
            tot += nv[j];
            if (tot < 0) neg = true;
            mv[j] = tot;
        }
        for (int i = 0; i < 4; i++) out[i] = neg ? 0.25 : mv[i];
    } else {
        if (t == 0.0) { for (int i = 0; i < 4; i++) out[i] = (i == nuc) ? 1.0 : 0.0; return; }
        double nv[4];
        for (int i = 0; i < 4; i++) nv[i] = (up ? c.q(r, nuc, i) : c.q(r, i, nuc)) * t;
        double d = 0.0;
        for (int i = 0; i < 4; i++) if (i == nuc) { nv[i] += 1.0; d = nv[i]; }
        bool neg = d < 0;
        for (int i = 0; i < 4; i++) out[i] = neg ? 0.25 : nv[i];
    }
}

// NOTE on the early "return [0.25]*4" of the reference: it leaves the loop at the FIRST negative
// component; every later component is then irrelevant, so testing after the loop is equivalent.

// ---- simplify (M:3697-3717): 4 = R, 0-3 = nucleotide, 6 = keep O, -1 = fatal ------------
template <class C> __device__ inline int simplify(const C &c, const double *v, int refNuc)
{
    double maxP = 0.0; int maxI = 0, numA = 0;
    for (int i = 0; i < 4; i++) {
        if (v[i] > maxP) { maxP = v[i]; maxI = i; }
        if (v[i] > c.m.thresholdProb) numA++;
    }
    if (maxP < c.m.thresholdProb4) return -1;
    if (numA == 1) return maxI == refNuc ? 4 : maxI;
    return 6;
}

__device__ inline double fmin_py(double a, double b) { return (b < a) ? b : a; }   // Python min(a, b)
// v[i] for a register-resident 4-vector without forcing it into scratch memory
__device__ inline double sel4(const double *v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }

// ---- appendProbNode (M:6505-6785) -----------------------------------------------------
// decode one packed word (+ its aux doubles) into the Ent view
__device__ __forceinline__ void decode_word(unsigned long long w, const double *aux, Ent &e)
{
    const uint32_t meta = (uint32_t)(w >> 32);
    e.pos = (int)(uint32_t)w;
    e.type = meta & 7u;
    e.ref = (meta >> 3) & 3u;
    e.hasD0 = meta & (1u << 5);
    e.hasD1 = meta & (1u << 6);
    e.flag = meta & (1u << 7);
    const double *a = aux + (meta >> 8);
    e.d0 = 0.0; e.d1 = 0.0;
    if (e.hasD0) { e.d0 = *a++; }
    if (e.hasD1) { e.d1 = *a++; }
    e.vec = a;
}

// Contribution of ONE site whose entries involve an O vector or an observation beyond the root (e1 = parent side,
// e2 = child side; neither is N, they are not both R).  Returns 0 and the factor in *f, or 1 when the reference
// returns -inf (zero-length mismatch).  Always inlined: a call would force the entry views through scratch memory.
template <bool RV, bool U, bool SS>
__device__ __forceinline__ int site_factor(const Ctx<RV, U, SS> &c, const Ent &e1, const Ent &e2, int site, bool isTipC,
                                           double bLen, double *fOut)
{
    typedef Ctx<RV, U, SS> CT;
    const double *rf = c.rf;
    double cl = bLen;                                                   // M:6586-6599
    if (e1.type < 5) { if (e1.hasD1) cl += e1.d1; else if (e1.hasD0) cl += e1.d0; }
    else if (e1.hasD0) cl += e1.d0;
    if (e2.type < 5) { if (e2.hasD0 && !e2.hasD1) cl += e2.d0; }
    else if (e2.hasD0) cl += e2.d0;
    const double r = c.rate(site);
    const bool flag1 = U && e1.type < 5 && e1.hasD0 && e1.flag;
    const bool flag2 = U && e2.type < 5 && (isTipC || (e2.hasD0 && e2.flag));
    double f;
    if (e1.type == 6 && e2.type == 6) {                                 // M:6677-6686
        double t3[4];
        gpv_vec(c, r, e2.vec, cl, false, t3);
        double tot = 0.0;
        for (int j = 0; j < 4; j++) tot += e1.vec[j] * t3[j];
        f = tot;
    } else if (e1.type == 6) {                                          // O over nucleotide/R, M:6687-6703
        int i2 = (e2.type == 4) ? e1.ref : e2.type;
        double p = e1.vec[i2];
        if (p > 0.02) f = p;
        else {
            double t3[4];
            gpv_nuc<CT, U>(c, r, i2, cl, flag2 ? c.err(site) : 0.0, false, flag2, t3);
            double tot = 0.0;
            for (int j = 0; j < 4; j++) tot += e1.vec[j] * t3[j];
            f = tot;
        }
    } else if (e2.type == 6) {                                          // nucleotide/R over O, M:6611-6633, 6744-6761
        int i1 = (e1.type == 4) ? e2.ref : e1.type;
        double p = e2.vec[i1];
        if (p > 0.02) f = p;
        else if (e1.hasD1) {
            double t2[4], t3[4];
            gpv_vec(c, r, e2.vec, cl, false, t3);
            gpv_nuc<CT, U>(c, r, i1, e1.d0, c.err(site), false, flag1, t2);
            double tot = 0.0;
            if (e1.type == 4) { for (int i = 0; i < 4; i++) tot += t3[i] * t2[i] * rf[i]; }
            else { for (int i = 0; i < 4; i++) tot += t2[i] * t3[i] * rf[i]; }
            f = tot / rf[i1];
        } else if (cl != 0.0) {
            double t3[4];
            gpv_vec(c, r, e2.vec, cl, false, t3);
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
    *fOut = f;
    return 0;
}

// The walk is a small state machine (start / step / finish): one lane, factors multiplied in genome order exactly
// like the reference, so the result is bit-for-bit that of the CPU restatement (apart from log()).
// Every entry carries its last position, so the end of the current segment is min(end1, end2) for every type pair,
// and "advance the cursor whose entry ends here" is the reference's stepping rule (M:6545-6770) in one line.
// bit (t1 * 8 + t2) set: an entry pair of these types needs per-site work (types 0-3 nucleotides, 4 R, 5 N, 6 O)
__host__ __device__ constexpr unsigned long long work_table()
{
    unsigned long long w = 0;
    for (int a = 0; a < 7; a++)
        for (int b = 0; b < 7; b++) {
            const bool work = (a != 5) && (b != 5) && !(a == 4 && b == 4) && !(a == b && a < 4);
            if (work) w |= 1ull << (a * 8 + b);
        }
    return w;
}

template <bool RV, bool U, bool SS> struct PairWalk {
    typedef Ctx<RV, U, SS> CT;
    const CT &c;
    ListRef Cl;                        // child list
    const unsigned long long *cw;
    bool isTipC; double bLen;
    ListRef P;                         // parent list
    const unsigned long long *pw;
    unsigned long long wa, wb;         // current words: low half = pos, high half = meta
    int ia, ib;
    double tf, Lk;
    bool dead;
    int lRef; double carry;            // hot scalars of the model kept in registers
    // log(totalFactor) is owed every time the running product falls below minimumCarryOver (M:6772-6783), about once
    // per pair -- i.e. in every third iteration SOME lane of a wavefront would run the ~90-instruction log() alone.
    // The products are parked here instead and their logs are added, in the same order, by finish().
    double carry1, carry2;
    int nCarry;

    // cwStaged: the child list's words staged somewhere faster than Cl.w (the batch kernels keep the query in LDS)
    __device__ PairWalk(const CT &c_, ListRef Cl_, bool isTipC_, double bLen_, const unsigned long long *cwStaged = nullptr)
        : c(c_), Cl(Cl_), cw(cwStaged ? cwStaged : (const unsigned long long *)Cl_.w), isTipC(isTipC_), bLen(bLen_),
          lRef(c_.m.lRef), carry(c_.m.minimumCarryOver) {}

    __device__ inline void start(ListRef P_)
    {
        P = P_;
        pw = (const unsigned long long *)P.w;
        wa = pw[0]; wb = cw[0];
        ia = ib = 0;
        tf = 1.0;
        Lk = bLen * c.m.globalTotRate;                                   // M:6541
        if (U && isTipC) Lk += c.m.totError;                             // M:6542-6543
        dead = false;
        carry1 = carry2 = 1.0;
        nCarry = 0;
    }

    // one segment of the two-list walk; returns true when the end of the genome (or a dead end) is reached
#ifndef MAPLE_WALK_OLD
    // The lanes of a wavefront walk unrelated pairs, so a step costs the wavefront the UNION of the cases its lanes are in.
    // Per pair of neighbouring lists on the bench trees (94 steps): 67 steps need no work, 13 are two different nucleotides,
    // 11.5 an O vector against a nucleotide / R whose entry of the vector is > 0.02 (the reference's shortcut, M:6615 / 6692 /
    // 6746: the factor IS that entry), 2 the same with a smaller entry, 0.3 anything else.  So: the test for work in 32-bit
    // compares (no 64-bit shift), the two common cases without decoding the entries (one load for the shortcut; the branch
    // lengths of the two-nucleotide case and the site's rate requested together, one wait), the rest through the general
    // site_factor as before.  Same operations on the same operands in the same order: bit-identical.
    __device__ inline bool step()
    {
        const int pa = (int)(uint32_t)wa, pb = (int)(uint32_t)wb;
        const uint32_t m1 = (uint32_t)(wa >> 32), m2 = (uint32_t)(wb >> 32);
        const int t1 = m1 & 7u, t2 = m2 & 7u;
        const int pos = min(pa, pb);
        // work unless N is involved, both sides are reference runs, or both show the same nucleotide (work_table())
        if ((t1 != 5) & (t2 != 5) & ((t1 != t2) | (t1 == 6))) {
            const int site = pos - 1;
            const bool o1 = t1 == 6, o2 = t2 == 6;
            double f = 0.0;
            bool general = false;
            if (!(o1 | o2)) {
                if (m1 & (1u << 6)) general = true;                      // observation beyond the root on the parent side
                else {                                                   // two different nucleotides (R = the reference one)
                    double d1v = 0.0, d2v = 0.0;                         // M:6640-6668, 6713-6742
                    if (m1 & (1u << 5)) d1v = P.aux[m1 >> 8];
                    if ((m2 & (1u << 5)) && !(m2 & (1u << 6))) d2v = Cl.aux[m2 >> 8];
                    const double r = c.rate(site);
                    const double er = U ? c.err(site) : 0.0;
                    const double cl = (bLen + d1v) + d2v;                // (x + 0.0 == x: the sums the reference makes)
                    const int i1 = (t1 == 4) ? (int)((m2 >> 3) & 3u) : t1;
                    const int i2 = (t2 == 4) ? (int)((m1 >> 3) & 3u) : t2;
                    const double qv = c.q(r, i1, i2);
                    f = fmin_py(0.25, qv * cl);
                    if (U) {
                        const bool flag1 = (t1 != 4) && (m1 & (1u << 5)) && (m1 & (1u << 7));
                        const bool flag2 = isTipC || ((m2 & (1u << 5)) && (m2 & (1u << 7)));
                        if (t1 == 4) { if (flag2) f += er * 0.33333; else if (cl == 0.0) dead = true; }
                        else if (flag1 || flag2) f += (double)((int)flag1 + (int)flag2) * 0.33333 * er;
                        else if (cl == 0.0) dead = true;
                    } else if (cl == 0.0) dead = true;                   // zero-length mismatch: -inf (M:6663, 6742)
                }
            } else if (o1 & o2) general = true;
            else {
                // one O vector: its entry for the other side's nucleotide (R: the O entry's own reference nucleotide)
                const uint32_t mo = o1 ? m1 : m2;
                const double *ao = o1 ? P.aux : Cl.aux;
                const int tn = o1 ? t2 : t1;
                const int i = (tn == 4) ? (int)((mo >> 3) & 3u) : tn;
                f = ao[(mo >> 8) + ((mo >> 5) & 1u) + ((mo >> 6) & 1u) + (uint32_t)i];
                general = !(f > 0.02);                                   // M:6615, 6692, 6746
            }
            if (general) {
                Ent e1, e2;
                decode_word(wa, P.aux, e1);
                decode_word(wb, Cl.aux, e2);
                if (site_factor(c, e1, e2, site, isTipC, bLen, &f) == 1) dead = true;
            }
            if (dead) return true;
            tf *= f;
            if (tf <= carry) {                                           // M:6772-6783
                if (tf < 2.2250738585072014e-308) { dead = true; return true; }
                if (nCarry == 2) { Lk += log(carry1); carry1 = carry2; nCarry = 1; }     // a third one: settle the oldest
                if (nCarry == 0) carry1 = tf; else carry2 = tf;
                ++nCarry;
                tf = 1.0;
            }
        }
        if (pos == lRef) return true;
        if (pa == pos) { ++ia; wa = pw[ia]; }
        if (pb == pos) { ++ib; wb = cw[ib]; }
        return false;
    }
#else
    __device__ inline bool step()
    {
        const int pa = (int)(uint32_t)wa, pb = (int)(uint32_t)wb;
        const uint32_t m1 = (uint32_t)(wa >> 32), m2 = (uint32_t)(wb >> 32);
        const int t1 = m1 & 7u, t2 = m2 & 7u;
        const int pos = min(pa, pb);
        // a site needs work unless N is involved, both sides are reference runs, or both show the same nucleotide:
        // one bit per (t1, t2) of a 64-bit table instead of six compares
        constexpr unsigned long long WORK = work_table();
        if ((WORK >> (t1 * 8 + t2)) & 1ull) {
            const int site = pos - 1;
            if (t1 == 6 || t2 == 6 || (m1 & (1u << 6))) {               // O vector or observation beyond the root
                Ent e1, e2;
                decode_word(wa, P.aux, e1);
                decode_word(wb, Cl.aux, e2);
                double f;
                if (site_factor(c, e1, e2, site, isTipC, bLen, &f) == 1) dead = true;
                else tf *= f;
            } else {                                                     // two different nucleotides (R = the reference one)
                double cl = bLen;                                        // M:6640-6668, 6713-6742
                if (m1 & (1u << 5)) cl += P.aux[m1 >> 8];
                if ((m2 & (1u << 5)) && !(m2 & (1u << 6))) cl += Cl.aux[m2 >> 8];
                const int i1 = (t1 == 4) ? (int)((m2 >> 3) & 3u) : t1;
                const int i2 = (t2 == 4) ? (int)((m1 >> 3) & 3u) : t2;
                const double qv = c.q(c.rate(site), i1, i2);
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
            // the running product and the dead flag only change here, so only here can they end or rescale the walk
            if (dead) return true;
            if (tf <= carry) {                                           // M:6772-6783
                if (tf < 2.2250738585072014e-308) { dead = true; return true; }
                if (nCarry == 2) { Lk += log(carry1); carry1 = carry2; nCarry = 1; }     // a third one: settle the oldest
                if (nCarry == 0) carry1 = tf; else carry2 = tf;
                ++nCarry;
                tf = 1.0;
            }
        }
        if (pos == lRef) return true;
        if (pa == pos) { ++ia; wa = pw[ia]; }
        if (pb == pos) { ++ib; wb = cw[ib]; }
        return false;
    }
#endif

    // The whole walk with the lanes of a wavefront brought together at the sites that need work.  Most steps of a walk need
    // none (both sides reference runs, or missing data); a wavefront that calls step() in a loop executes the ~100
    // instructions of the per-site arithmetic at EVERY step, because at every step some lane of the 64 has a site to work on.
    // Here a lane first runs ahead over its work-free steps (a dozen instructions each) and stops at its next site; the
    // arithmetic is then executed once for all lanes that have one.  Same steps in the same order per lane: same result.
    __device__ inline void run()
    {
        constexpr unsigned long long WORK = work_table();
        for (;;) {
            bool end = false;
            for (;;) {
                const int pa = (int)(uint32_t)wa, pb = (int)(uint32_t)wb;
                const int t1 = (int)((wa >> 32) & 7ull), t2 = (int)((wb >> 32) & 7ull);
                if ((WORK >> (t1 * 8 + t2)) & 1ull) break;
                const int pos = min(pa, pb);
                if (pos == lRef) { end = true; break; }
                if (pa == pos) { ++ia; wa = pw[ia]; }
                if (pb == pos) { ++ib; wb = cw[ib]; }
            }
            if (end) return;
            if (step()) return;
        }
    }

    __device__ inline double finish() const
    {
        if (dead) return -INFINITY;
        double lk = Lk;
        if (nCarry >= 1) lk += log(carry1);
        if (nCarry >= 2) lk += log(carry2);
        return (tf > 0.0) ? lk + log(tf) : -INFINITY;
    }
};

template <bool RV, bool U, bool SS>
__device__ inline double append_walk(const Ctx<RV, U, SS> &c, ListRef P, ListRef Cl, bool isTipC, double bLen)
{
    PairWalk<RV, U, SS> w(c, Cl, isTipC, bLen);
    w.start(P);
    while (!w.step()) {}
    return w.finish();
}
// ... for kernels whose lanes walk unrelated pairs side by side (PairWalk::run)
template <bool RV, bool U, bool SS>
__device__ inline double append_walk_gathered(const Ctx<RV, U, SS> &c, ListRef P, ListRef Cl, bool isTipC, double bLen)
{
    PairWalk<RV, U, SS> w(c, Cl, isTipC, bLen);
    w.start(P);
    w.run();
    return w.finish();
}

// ---- mergeVectors (M:4446-4859) -----------------------------------------------------------
// returns number of entries written, -1 for the reference's None, -2 for a fatal state
template <bool RV, bool U, bool SS>
__device__ int merge_walk(const Ctx<RV, U, SS> &c, ListRef L1, double bLen1, bool tip1, ListRef L2, double bLen2,
                          bool tip2, bool upDown, bool wantLK, int numMinor1, int numMinor2, Writer &o, double *outLK)
{
    typedef Ctx<RV, U, SS> CT;
    const int lRef = c.m.lRef;
    const double *rf = c.rf;
    const double *cr = c.m.cumulativeRate, *cer = c.m.cumulativeErrorRate;
    Cursor a, b;
    a.init(L1); b.init(L2);
    int pos = 0;
    double totalFactor = 1.0, lk = 0.0;
    if (wantLK) {                                                       // M:4486-4494
        lk = (bLen1 + bLen2) * c.m.globalTotRate;
        if (U) {
            if (tip1 || numMinor1) lk += c.m.totError * (1 + numMinor1);
            if (tip2 || numMinor2) lk += c.m.totError * (1 + numMinor2);
        }
    }
    for (;;) {
        const Ent &e1 = a.e, &e2 = b.e;
        int newPos;
#include "merge_step_body.inc"
        pos = newPos;
        if (wantLK && totalFactor <= c.m.minimumCarryOver) {            // M:4830-4839
            if (totalFactor < 2.2250738585072014e-308) return -2;
            lk += log(totalFactor);
            totalFactor = 1.0;
        }
        if (pos == lRef) break;
        a.step(pos);
        b.step(pos);
    }
    if (wantLK && outLK) *outLK = lk + log(totalFactor);
    return o.n;
}

// ---- estimateBranchLengthWithDerivative (M:5040-5358) ---------------------------------------
// `ais` = per-lane scratch of at least (entries of P + entries of C) doubles, strided by `stride`
template <bool RV, bool U, bool SS>
__device__ double blen_walk(const Ctx<RV, U, SS> &c, ListRef P, ListRef Cl, bool fromTipC, double *ais, int stride,
                            bool *isFalse)
{
    const int lRef = c.m.lRef;
    const double *rf = c.rf;
    const double *cr = c.m.cumulativeRate;
    Cursor a, b;
    a.init(P); b.init(Cl);
    int pos = 0, nA = 0, nZeros = 0;
    double c1 = c.m.globalTotRate;
    *isFalse = false;
    for (;;) {
        const Ent &e1 = a.e, &e2 = b.e;
#define BLEN_C1_ADD(x) c1 += (x)
#define BLEN_C1_SUB(x) c1 -= (x)
#define BLEN_AIS(v) do { ais[(size_t)nA * stride] = (v); nA++; } while (0)
#define BLEN_ZERO() nZeros++
#include "blen_step_body.inc"
#undef BLEN_C1_ADD
#undef BLEN_C1_SUB
#undef BLEN_AIS
#undef BLEN_ZERO
        if (pos == lRef) break;
        a.step(pos);
        b.step(pos);
    }
    // bracket and bisect  sum 1/(a_i+t) + nZeros/t = c1   (M:5298-5358)
#include "blen_solve_body.inc"
}

// ---- areVectorsDifferent (M:5419-5472) -------------------------------------------------------
template <class C> __device__ bool differ_walk(const C &c, ListRef L1, ListRef L2)
{
    const int lRef = c.m.lRef;
    const double thr = c.m.thresholdProb;
    Cursor a, b;
    a.init(L1); b.init(L2);
    int pos = 0;
    for (;;) {
        const Ent &e1 = a.e, &e2 = b.e;
        if (e1.type != e2.type) return true;
        if (e1.hasD0 != e2.hasD0 || e1.hasD1 != e2.hasD1) return true;     // tuple lengths
        if (e1.type < 5) {
            if (e1.hasD0) {
                if (fabs(e1.d0 - e2.d0) > thr) return true;
                if (e1.hasD1 && fabs(e1.d1 - e2.d1) > thr) return true;
                if (e1.flag != e2.flag) return true;                        // |True-False| = 1 > thr
            }
            pos = (e1.type < 4) ? pos + 1 : min(e1.pos, e2.pos);
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
            pos += 1;
        } else pos = min(e1.pos, e2.pos);
        if (pos == lRef) break;
        a.step(pos);
        b.step(pos);
    }
    return false;
}

// ---- findProbRoot (M:4865-4912): log-likelihood of a lower list (already in the root frame) at the root ----
template <bool RV, bool U, bool SS> __device__ double root_prob_walk(const Ctx<RV, U, SS> &c, ListRef L)
{
    const int lRef = c.m.lRef;
    const double *rf = c.rf;
    const int32_t *cb = c.m.cumulativeBases;
    Cursor a;
    a.init(L);
    double logLK = 0.0, logFactor = 1.0;
    int pos = 0;
    for (;;) {
        const Ent &e = a.e;
        if (U && e.type < 5 && e.hasD0 && e.flag) {
            if (e.type == 4) { logLK += c.m.rootFreqsLogErrorCumulative[e.pos] - c.m.rootFreqsLogErrorCumulative[pos]; pos = e.pos; }
            else {
                double er = c.err(pos);
                logFactor *= (sel4(rf, e.type) * (1.0 - 1.33333 * er) + 0.33333 * er);
                pos += 1;
            }
        } else if (e.type == 4) {
            for (int i = 0; i < 4; i++) logLK += c.m.rootFreqsLog[i] * (cb[e.pos * 4 + i] - cb[pos * 4 + i]);
            pos = e.pos;
        } else if (e.type < 4) { logLK += c.m.rootFreqsLog[e.type]; pos += 1; }
        else if (e.type == 6) {
            double tot = 0.0;
            for (int i = 0; i < 4; i++) tot += rf[i] * e.vec[i];
            logFactor *= tot;
            pos += 1;
        } else pos = e.pos;
        if (logFactor <= c.m.minimumCarryOver) {
            if (logFactor < 2.2250738585072014e-308) return -INFINITY;
            logLK += log(logFactor);
            logFactor = 1.0;
        }
        if (pos == lRef) break;
        a.next();
    }
    logLK += log(logFactor);
    return logLK;
}

// ---- isMinorSequence (M:5919-6004): 1 = list2 is at most as informative as list1, 2 = the opposite, 0 = neither ----
__device__ inline int minor_walk(int lRef, ListRef L1, ListRef L2, bool onlyFindIdentical)
{
    Cursor a, b;
    a.init(L1); b.init(L2);
    int pos = 0;
    bool big1 = false, big2 = false;
    for (;;) {
        const Ent &e1 = a.e, &e2 = b.e;
        if (e1.type != e2.type) {
            if (onlyFindIdentical) return 0;
            if (e1.type == 5) { pos = (e2.type == 4) ? min(e1.pos, e2.pos) : pos + 1; big2 = true; }
            else if (e2.type == 5) { pos = (e1.type == 4) ? min(e1.pos, e2.pos) : pos + 1; big1 = true; }
            else if (e1.type == 6) {
                int i2 = (e2.type == 4) ? e1.ref : e2.type;
                if (e1.vec[i2] > 0.1) big2 = true; else return 0;
                pos += 1;
            } else if (e2.type == 6) {
                int i1 = (e1.type == 4) ? e2.ref : e1.type;
                if (e2.vec[i1] > 0.1) big1 = true; else return 0;
                pos += 1;
            } else return 0;
        } else if (e1.type == 6) {
            for (int j = 0; j < 4; j++) {
                if (onlyFindIdentical) { if (e2.vec[j] != e1.vec[j]) return 0; }
                else if (e2.vec[j] > 0.1 && e1.vec[j] < 0.1) big1 = true;
                else if (e1.vec[j] > 0.1 && e2.vec[j] < 0.1) big2 = true;
            }
            pos += 1;
        } else pos = (e1.type < 4) ? pos + 1 : min(e1.pos, e2.pos);
        if (big1 && big2) return 0;
        if (pos == lRef) break;
        a.step(pos);
        b.step(pos);
    }
    if (big1) return 1;
    return big2 ? 2 : 1;
}

// ---- passGenomeListThroughBranch (M:3749-3877) -----------------------------------------------
// mutations: int32 triples (pos, from, to) sorted by pos
__device__ inline int pass_walk(int lRef, ListRef L, const int32_t *mut, int nMut, bool dirIsUp, Writer &o)
{
    Cursor a;
    a.init(L);
    int iM = 0, lastPos = 0;
    for (;;) {
        const Ent &e = a.e;
        if (e.type == 5) {
            o.bare(5, e.pos, 0);
            lastPos = e.pos;
            while (iM < nMut && mut[iM * 3] <= lastPos) iM++;
        } else if (e.type == 4) {
            // split the reference run around mutated sites; each mutated site becomes an explicit nucleotide
            while (iM < nMut && mut[iM * 3] <= e.pos) {
                int mp = mut[iM * 3];
                if (mp > lastPos + 1) { lastPos = mp - 1; o.copy(e, 4, lastPos, 0); }
                lastPos += 1;
                int from = mut[iM * 3 + 1], to = mut[iM * 3 + 2];
                o.copy(e, dirIsUp ? to : from, lastPos, dirIsUp ? from : to);
                iM++;
            }
            if (lastPos < e.pos) { lastPos = e.pos; o.copy(e, 4, lastPos, 0); }
        } else {
            lastPos += 1;
            if (iM < nMut && mut[iM * 3] <= lastPos) {
                int newRef = dirIsUp ? mut[iM * 3 + 1] : mut[iM * 3 + 2];
                iM++;
                if (e.type == 6) o.copy(e, 6, lastPos, newRef);
                else if (e.type == newRef) o.copy(e, 4, lastPos, 0);        // equals the new reference -> R
                else o.copy(e, e.type, lastPos, newRef);
            } else o.copy(e, e.type, lastPos, e.ref);
        }
        if (lastPos == lRef) break;
        a.next();
    }
    return o.n;
}

// ---- shorten (M:3721-3745) -------------------------------------------------------------------
// Adjacent R entries with matching tails collapse into the later one.  The reference compares
// every candidate with the FIRST entry of the current group (it never refreshes `entryOld` after
// a pop), which is reproduced by keeping `head`.
template <class C> __device__ int shorten_walk(const C &c, ListRef L, int nEnt, Writer &o)
{
    const double thr = c.m.thresholdProb;
    Cursor a;
    a.init(L);
    Ent head = a.e;                    // entryOld
    Ent last = a.e;                    // entry currently at vec[index]
    for (int k = 1; k < nEnt; k++) {
        a.next();
        const Ent &nw = a.e;
        bool absorb = false;
        if (nw.type == 4 && head.type == 4 && nw.hasD0 == head.hasD0 && nw.hasD1 == head.hasD1) {
            if (!nw.hasD0) absorb = true;
            else if (fabs(nw.d0 - head.d0) > thr) absorb = false;
            else if (nw.hasD1 && fabs(nw.d1 - head.d1) > thr) absorb = false;
            else absorb = (nw.flag == head.flag);
        }
        if (!absorb) {
            o.copy(last, last.type, last.pos, last.ref);
            head = nw;
        }
        last = nw;
    }
    o.copy(last, last.type, last.pos, last.ref);
    return o.n;
}

// ---- rootVector body (M:4941-4986): lower list (already in the root frame) -> upper list ------
template <bool RV, bool U, bool SS>
__device__ int root_walk(const Ctx<RV, U, SS> &c, ListRef L, double bLen, bool isFromTip, Writer &o)
{
    const int lRef = c.m.lRef;
    const double *rf = c.rf;
    Cursor a;
    a.init(L);
    int prev = 0;                      // positions consumed so far
    for (;;) {
        const Ent &e = a.e;
        if (e.type == 5) o.bare(5, e.pos, 0);
        else if (e.type == 6) {
            double nv[4];
            double tb = bLen;
            if (e.hasD0) tb += e.d0;
            if (tb != 0.0) { gpv_vec(c, c.rate(prev), e.vec, tb, false, nv); for (int i = 0; i < 4; i++) nv[i] *= rf[i]; }
            else for (int i = 0; i < 4; i++) nv[i] = e.vec[i] * rf[i];
            double s = 0.0;
            for (int i = 0; i < 4; i++) s += nv[i];
            for (int i = 0; i < 4; i++) nv[i] /= s;
            o.put(6, e.pos, e.ref, false, 0, false, 0, false, nv);
        } else {
            bool fl = U && ((e.hasD0 && e.flag) || isFromTip);
            if (e.hasD0) o.put(e.type, e.pos, e.ref, true, e.d0 + bLen, true, 0.0, fl, nullptr);
            else if (bLen != 0.0 || fl) o.put(e.type, e.pos, e.ref, true, bLen, true, 0.0, fl, nullptr);
            else o.bare(e.type, e.pos, e.ref);
        }
        prev = e.pos;
        if (prev == lRef) break;
        a.next();
    }
    return o.n;
}

} // namespace maple
