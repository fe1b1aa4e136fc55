/* maple_amd/csrc/synth_gen.c -- the synthetic-input generator of SURVEY.md section 8d ("Synthetic inputs") in plain C.
 *
 * Host code, not on the hot path: it makes the INPUTS of bench.py and of the big-tree tests (a reference genome, a random
 * bifurcating tree, substitutions dropped on its branches, and the samples' MAPLE-format difference lists, the format
 * read by readConciseAlignment, MAPLEv0.7.5.4.py:3498-3553).  maple_amd/synth.py's make_dataset does the same in Python
 * with numpy's generator and stays the generator of the 10 000 / 100 000-sample workloads of rounds 1-3; at 1 000 000
 * samples it takes minutes, this one seconds (one depth-first pass over the tree carrying the current state as a small
 * sorted array).  Same model, its own seeded stream (xoshiro256**): "synth v2".
 *
 * Built by __graft_entry__.build() with gcc into maple_amd/libmaple_synth.so; bound by maple_amd/synth.py (ctypes).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t n_samples, n_nodes, l_ref, n_diffs;
    int8_t *ref;          /* [l_ref] 0..3 */
    int64_t *parent;      /* [n_nodes], -1 root; parents precede children */
    double *blen;         /* [n_nodes] */
    int64_t *tip_node;    /* [n_samples] node of sample i */
    int64_t *diff_off;    /* [n_samples + 1] */
    uint8_t *diff_code;   /* [n_diffs] the MAPLE character: 'a' 'c' 'g' 't', 'n', or a two-state IUPAC code */
    int32_t *diff_pos;    /* [n_diffs] 1-based */
    int32_t *diff_len;    /* [n_diffs] 1, or the length of an 'n' run */
    double mean_depth, per_branch;
} maple_synth;

/* ---- xoshiro256** seeded by splitmix64 ---- */
typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix(uint64_t *x) { uint64_t z = (*x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static void rng_seed(rng_t *r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_u64(rng_t *r)
{
    uint64_t *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static inline double rng_f(rng_t *r) { return (double)(rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline int64_t rng_int(rng_t *r, int64_t n) { return (int64_t)(rng_f(r) * (double)n); }   /* [0, n) */
static int rng_poisson(rng_t *r, double lam)
{
    /* Knuth; lam is ~1 here */
    const double L = exp(-lam);
    int k = 0;
    double p = 1.0;
    do { k++; p *= rng_f(r); } while (p > L && k < 1000);
    return k - 1;
}
static int pick4(const double *cdf, double u) { return u < cdf[0] ? 0 : (u < cdf[1] ? 1 : (u < cdf[2] ? 2 : 3)); }

void maple_synth_free(maple_synth *d)
{
    if (!d) return;
    free(d->ref); free(d->parent); free(d->blen); free(d->tip_node); free(d->diff_off); free(d->diff_code); free(d->diff_pos); free(d->diff_len);
    free(d);
}

typedef struct { int32_t pos; int8_t nuc; } st_t;     /* one site that differs from the reference (0-based pos) */

/* the two-state IUPAC code of an unordered pair of states */
static uint8_t ambig2(int a, int b)
{
    if (a > b) { int t = a; a = b; b = t; }
    if (a == 0 && b == 2) return 'r';
    if (a == 1 && b == 3) return 'y';
    if (a == 1 && b == 2) return 's';
    if (a == 0 && b == 3) return 'w';
    if (a == 2 && b == 3) return 'k';
    return 'm';
}

/* site_cdf: cumulative distribution over the l_ref sites (null: uniform); freqs_cdf: of the reference composition;
 * exit_cdf[4][4]: per from-state cumulative exit probabilities (diagonal 0). */
maple_synth *maple_synth_generate(int64_t n_samples, int64_t l_ref, uint64_t seed, double mean_diffs, const double *site_cdf,
                                  const double *freqs_cdf, const double *exit_cdf, double frac_with_n, int32_t nrun_lo,
                                  int32_t nrun_hi, double frac_ambig, double zero_branch_frac)
{
    if (n_samples < 2 || l_ref < 16) return NULL;
    maple_synth *d = (maple_synth *)calloc(1, sizeof *d);
    if (!d) return NULL;
    const int64_t n = n_samples, nn = 2 * n - 1;
    d->n_samples = n; d->n_nodes = nn; d->l_ref = l_ref;
    rng_t R;
    rng_seed(&R, seed);
    d->ref = (int8_t *)malloc((size_t)l_ref);
    d->parent = (int64_t *)malloc((size_t)nn * sizeof(int64_t));
    d->blen = (double *)malloc((size_t)nn * sizeof(double));
    d->tip_node = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    d->diff_off = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    int64_t *tips = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int64_t *c0 = (int64_t *)malloc((size_t)nn * sizeof(int64_t)), *c1 = (int64_t *)malloc((size_t)nn * sizeof(int64_t));
    int32_t *depth = (int32_t *)malloc((size_t)nn * sizeof(int32_t)), *nmut = (int32_t *)malloc((size_t)nn * sizeof(int32_t));
    int64_t *moff = (int64_t *)malloc((size_t)(nn + 1) * sizeof(int64_t));
    if (!d->ref || !d->parent || !d->blen || !d->tip_node || !d->diff_off || !tips || !c0 || !c1 || !depth || !nmut || !moff) goto oom;
    for (int64_t i = 0; i < l_ref; i++) d->ref[i] = (int8_t)pick4(freqs_cdf, rng_f(&R));
    /* random bifurcating topology by sequential random attachment: a random tip becomes an internal node with two tips */
    for (int64_t v = 0; v < nn; v++) { d->parent[v] = -1; c0[v] = c1[v] = -1; }
    d->parent[1] = d->parent[2] = 0; c0[0] = 1; c1[0] = 2;
    tips[0] = 1; tips[1] = 2;
    {
        int64_t nt = 2, nxt = 3;
        for (int64_t s = 0; s < n - 2; s++) {
            const int64_t k = rng_int(&R, nt), t = tips[k], a = nxt, b = nxt + 1;
            nxt += 2;
            d->parent[a] = d->parent[b] = t; c0[t] = a; c1[t] = b;
            tips[k] = a; tips[nt++] = b;
        }
    }
    depth[0] = 0;
    for (int64_t v = 1; v < nn; v++) depth[v] = depth[d->parent[v]] + 1;
    {
        double sd = 0.0;
        for (int64_t v = 0; v < nn; v++) if (c0[v] < 0) sd += depth[v];
        d->mean_depth = sd / (double)n;
        if (d->mean_depth < 1.0) d->mean_depth = 1.0;
    }
    d->per_branch = mean_diffs / d->mean_depth / fmax(1e-9, 1.0 - zero_branch_frac);
    moff[0] = 0;
    for (int64_t v = 0; v < nn; v++) {
        int k = rng_poisson(&R, d->per_branch);
        if (rng_f(&R) < zero_branch_frac) k = 0;
        if (v == 0) k = 0;
        nmut[v] = k;
        d->blen[v] = (double)k / (double)l_ref;
        moff[v + 1] = moff[v] + k;
    }
    /* the sites the substitutions of every branch fall on (the new state depends on the state it meets: drawn in the pass) */
    int32_t *msite = (int32_t *)malloc((size_t)(moff[nn] + 1) * sizeof(int32_t));
    double *mu = (double *)malloc((size_t)(moff[nn] + 1) * sizeof(double));
    if (!msite || !mu) { free(msite); free(mu); goto oom; }
    for (int64_t j = 0; j < moff[nn]; j++) {
        const double u = rng_f(&R);
        int64_t p;
        if (site_cdf) {
            int64_t lo = 0, hi = l_ref - 1;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (site_cdf[mid] > u) hi = mid; else lo = mid + 1; }
            p = lo;
        } else p = (int64_t)(u * (double)l_ref);
        msite[j] = (int32_t)p;
        mu[j] = rng_f(&R);
    }
    /* depth-first pass: the state (sites that differ from the reference, sorted) is edited on the way down, restored on the
     * way up; every tip writes its difference list */
    {
        const int64_t maxd = 4096;
        size_t capS = 4096, nS = 0;
        st_t *S = (st_t *)malloc(capS * sizeof(st_t));
        /* undo log: (site, old nuc or -1 = was absent) per applied substitution */
        size_t capU = 1 << 16, nU = 0;
        st_t *Ulog = (st_t *)malloc(capU * sizeof(st_t));
        int64_t *stack = (int64_t *)malloc((size_t)(2 * maxd + 8) * sizeof(int64_t));
        size_t capD = (size_t)((double)n * (mean_diffs + 8.0)) + 1024, nD = 0;
        uint8_t *dc = (uint8_t *)malloc(capD);
        int32_t *dp = (int32_t *)malloc(capD * sizeof(int32_t)), *dl = (int32_t *)malloc(capD * sizeof(int32_t));
        int64_t *tipOff = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t)), *tipNode = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        int64_t nTip = 0;
        int ok = S && Ulog && stack && dc && dp && dl && tipOff && tipNode;
        int64_t sp = 0;
        if (ok) { stack[sp++] = 0; tipOff[0] = 0; }
        while (ok && sp > 0) {
            int64_t v = stack[--sp];
            if (v < 0) {                                            /* leaving node ~v: undo its substitutions, last first */
                v = ~v;
                for (int k = 0; k < nmut[v]; k++) {
                    const st_t u = Ulog[--nU];
                    /* find the site */
                    size_t lo = 0, hi = nS;
                    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (S[mid].pos < u.pos) lo = mid + 1; else hi = mid; }
                    const int present = lo < nS && S[lo].pos == u.pos;
                    if (u.nuc < 0) { if (present) { memmove(S + lo, S + lo + 1, (nS - lo - 1) * sizeof(st_t)); nS--; } }
                    else if (present) S[lo].nuc = u.nuc;
                    else { memmove(S + lo + 1, S + lo, (nS - lo) * sizeof(st_t)); S[lo] = u; nS++; }
                }
                continue;
            }
            for (int64_t j = moff[v]; j < moff[v + 1]; j++) {       /* entering: this branch's substitutions */
                const int32_t p = msite[j];
                size_t lo = 0, hi = nS;
                while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (S[mid].pos < p) lo = mid + 1; else hi = mid; }
                const int present = lo < nS && S[lo].pos == p;
                const int cur = present ? S[lo].nuc : d->ref[p];
                const int nw = pick4(exit_cdf + 4 * cur, mu[j]);
                if (nU + 1 > capU) { capU *= 2; Ulog = (st_t *)realloc(Ulog, capU * sizeof(st_t)); if (!Ulog) { ok = 0; break; } }
                Ulog[nU].pos = p; Ulog[nU].nuc = (int8_t)(present ? cur : -1); nU++;
                if (nw == d->ref[p]) { if (present) { memmove(S + lo, S + lo + 1, (nS - lo - 1) * sizeof(st_t)); nS--; } }
                else if (present) S[lo].nuc = (int8_t)nw;
                else {
                    if (nS + 1 > capS) { capS *= 2; S = (st_t *)realloc(S, capS * sizeof(st_t)); if (!S) { ok = 0; break; } }
                    memmove(S + lo + 1, S + lo, (nS - lo) * sizeof(st_t));
                    S[lo].pos = p; S[lo].nuc = (int8_t)nw; nS++;
                }
            }
            if (!ok) break;
            if (sp + 3 > 2 * maxd) { ok = 0; break; }               /* (a random tree of this kind is ~2 log2 n deep) */
            stack[sp++] = ~v;
            if (c0[v] >= 0) { stack[sp++] = c1[v]; stack[sp++] = c0[v]; continue; }
            /* a tip: its MAPLE entries */
            if (nD + nS + 16 > capD) {
                capD = capD * 2 + nS;
                dc = (uint8_t *)realloc(dc, capD); dp = (int32_t *)realloc(dp, capD * sizeof(int32_t)); dl = (int32_t *)realloc(dl, capD * sizeof(int32_t));
                if (!dc || !dp || !dl) { ok = 0; break; }
            }
            const size_t base = nD;
            for (size_t k = 0; k < nS; k++) { dc[nD] = (uint8_t)"acgt"[S[k].nuc]; dp[nD] = S[k].pos + 1; dl[nD] = 1; nD++; }
            if (rng_f(&R) < frac_ambig) {                           /* 1-2 two-state ambiguities that include the true state */
                const int cnt = 1 + (int)rng_int(&R, 2);
                for (int a = 0; a < cnt; a++) {
                    const int32_t p = (int32_t)rng_int(&R, l_ref);
                    size_t k = base;
                    while (k < nD && dp[k] < p + 1) k++;
                    const int present = k < nD && dp[k] == p + 1;
                    int cur = d->ref[p];
                    if (present) { const uint8_t ch = dc[k]; if (ch == 'a') cur = 0; else if (ch == 'c') cur = 1; else if (ch == 'g') cur = 2; else if (ch == 't') cur = 3; else continue; }
                    const int other = (cur + 1 + (int)rng_int(&R, 3)) & 3;
                    if (!present) { memmove(dc + k + 1, dc + k, nD - k); memmove(dp + k + 1, dp + k, (nD - k) * sizeof(int32_t)); memmove(dl + k + 1, dl + k, (nD - k) * sizeof(int32_t)); nD++; }
                    dc[k] = ambig2(cur, other); dp[k] = p + 1; dl[k] = 1;
                }
            }
            if (rng_f(&R) < frac_with_n) {                          /* 1-3 runs of missing data; what they cover is dropped */
                const int cnt = 1 + (int)rng_int(&R, 3);
                for (int a = 0; a < cnt; a++) {
                    int32_t ln = nrun_lo + (int32_t)rng_int(&R, (int64_t)(nrun_hi - nrun_lo + 1));
                    int32_t s = 1 + (int32_t)rng_int(&R, (l_ref - ln > 1 ? l_ref - ln : 1));
                    if (s + ln - 1 > l_ref) ln = (int32_t)(l_ref - s + 1);
                    int32_t e = s + ln;                             /* [s, e) */
                    /* absorb the runs it touches (until none is left that does), then drop the entries it covers */
                    size_t w;
                    for (int changed = 1; changed;) {
                        changed = 0;
                        w = base;
                        for (size_t k = base; k < nD; k++) {
                            const int32_t ks = dp[k], ke = dp[k] + dl[k];
                            if (dc[k] == 'n' && ks <= e && s <= ke) { if (ks < s) s = ks; if (ke > e) e = ke; changed = 1; continue; }
                            dc[w] = dc[k]; dp[w] = dp[k]; dl[w] = dl[k]; w++;
                        }
                        nD = w;
                    }
                    w = base;
                    for (size_t k = base; k < nD; k++) { if (dp[k] >= s && dp[k] < e) continue; dc[w] = dc[k]; dp[w] = dp[k]; dl[w] = dl[k]; w++; }
                    nD = w;
                    size_t k = base;
                    while (k < nD && dp[k] < s) k++;
                    memmove(dc + k + 1, dc + k, nD - k); memmove(dp + k + 1, dp + k, (nD - k) * sizeof(int32_t)); memmove(dl + k + 1, dl + k, (nD - k) * sizeof(int32_t)); nD++;
                    dc[k] = 'n'; dp[k] = s; dl[k] = e - s;
                }
            }
            tipNode[nTip] = v; tipOff[++nTip] = (int64_t)nD;
        }
        free(S); free(Ulog); free(stack);
        if (!ok || nTip != n) { free(dc); free(dp); free(dl); free(tipOff); free(tipNode); free(msite); free(mu); goto oom; }
        /* the samples in a random order (Fisher-Yates over the tips as the pass met them) */
        int64_t *perm = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        d->diff_code = (uint8_t *)malloc(nD + 1);
        d->diff_pos = (int32_t *)malloc((nD + 1) * sizeof(int32_t));
        d->diff_len = (int32_t *)malloc((nD + 1) * sizeof(int32_t));
        if (!perm || !d->diff_code || !d->diff_pos || !d->diff_len) { free(perm); free(dc); free(dp); free(dl); free(tipOff); free(tipNode); free(msite); free(mu); goto oom; }
        for (int64_t i = 0; i < n; i++) perm[i] = i;
        for (int64_t i = n - 1; i > 0; i--) { const int64_t j = rng_int(&R, i + 1), t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        int64_t at = 0;
        d->diff_off[0] = 0;
        for (int64_t i = 0; i < n; i++) {
            const int64_t t = perm[i], lo = tipOff[t], hi = tipOff[t + 1];
            memcpy(d->diff_code + at, dc + lo, (size_t)(hi - lo));
            memcpy(d->diff_pos + at, dp + lo, (size_t)(hi - lo) * sizeof(int32_t));
            memcpy(d->diff_len + at, dl + lo, (size_t)(hi - lo) * sizeof(int32_t));
            at += hi - lo;
            d->diff_off[i + 1] = at;
            d->tip_node[i] = tipNode[t];
        }
        d->n_diffs = at;
        free(perm); free(dc); free(dp); free(dl); free(tipOff); free(tipNode);
    }
    free(msite); free(mu);
    free(tips); free(c0); free(c1); free(depth); free(nmut); free(moff);
    return d;
oom:
    free(tips); free(c0); free(c1); free(depth); free(nmut); free(moff);
    maple_synth_free(d);
    return NULL;
}
