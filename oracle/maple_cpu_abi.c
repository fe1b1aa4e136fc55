/*
 * oracle/maple_cpu_abi.c -- TEST INFRASTRUCTURE ONLY: libmaple_cpu.so, the CPU twin of libmaple_hip.so.
 *
 * "A CPU build of the same .so exports the identical ABI and is the on-box baseline."  The entry
 * points of include/maple_hip.h that make up the operator boundary of section 8b -- context, model tables, the genome-list
 * arena, MAT mutation lists, the batched list operators, the tree mirror and the SPR search batch -- implemented over the
 * plain-C restatement of the reference (maple_oracle.c, maple_oracle_search.c), with the same signatures, status codes and
 * return conventions (list id -1 = None, IEEE -inf, isFalse flags).  A caller -- tests/test_cpu_twin.py drives it through
 * the very binding class the product uses, maple_amd.runtime.Device -- can swap the two libraries and diff results.
 * The entry points outside that set (placement search, updatePartials, tree patch, RCCL arg-max, timing, debug) are NOT
 * exported: a binding that needs them fails at load time instead of computing something else.
 *
 * Nothing under maple_amd/ loads this library unless a test hands it in explicitly; the product path is libmaple_hip.so
 * and fails loudly without it.  Lists are kept as OEntry tuples (maple_oracle.h), converted from / to the packed form of
 * the ABI at upload / download.
 */
#include "../include/maple_hip.h"
#include "maple_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct maple_ctx {
    OModel m;
    maple_params params;
    unsigned char *refIdx;
    double *siteRates, *errorRates, *cumRate, *cumErr, *rflec;
    int *cumBases;
    int model_set;
    /* genome lists */
    OEntry *ent;
    long long capEnt, usedEnt;
    long long *off;
    int *len;
    long long nLists, capLists;
    /* mutation lists */
    int *mut3;
    long long capMut, usedMut;
    long long *moff;
    int *mcnt;
    long long nMut, capMutLists;
    /* tree */
    int tree_set, tn, troot;
    int *tup, *tc0, *tc1, *tminor;
    double *tdist;
    unsigned char *ttip;
    int *tcol[4], *tmut;
    int tolerate;
    char err[512];
};

static int fail(maple_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

int maple_abi_version(void) { return MAPLE_ABI_VERSION; }
const char *maple_last_error(maple_ctx *c) { return c ? c->err : "null context"; }

int maple_create(maple_ctx **out, int device, int32_t lRef, const uint8_t *refIdx, const double *rootFreqs4,
                 const maple_params *params, uint64_t arena_bytes)
{
    (void)device; (void)arena_bytes;
    if (!out || lRef <= 0 || !refIdx || !rootFreqs4 || !params) return MAPLE_ERR_ARG;
    maple_ctx *c = (maple_ctx *)calloc(1, sizeof *c);
    if (!c) return MAPLE_ERR_NOMEM;
    c->params = *params;
    c->refIdx = (unsigned char *)malloc((size_t)lRef);
    memcpy(c->refIdx, refIdx, (size_t)lRef);
    OModel *m = &c->m;
    m->lRef = lRef;
    m->refIdx = c->refIdx;
    for (int i = 0; i < 4; i++) m->rootFreqs[i] = rootFreqs4[i];
    m->globalTotRate = -(double)lRef;                                   /* M:3607 */
    m->minimumCarryOver = 2.2250738585072014e-308 * 1e50;               /* M:3623 */
    m->thresholdProb = params->thresholdProb;
    m->minBLenSensitivity = params->minBLenSensitivity;
    m->thresholdDiffForUpdate = params->thresholdDiffForUpdate;
    m->thresholdFoldChangeUpdate = params->thresholdFoldChangeUpdate;
    c->cumBases = (int *)calloc((size_t)(lRef + 1) * 4, sizeof(int));   /* M:3669-3674 */
    for (int i = 0; i < lRef; i++) {
        for (int k = 0; k < 4; k++) c->cumBases[(i + 1) * 4 + k] = c->cumBases[i * 4 + k];
        c->cumBases[(i + 1) * 4 + refIdx[i]]++;
    }
    *out = c;
    return MAPLE_OK;
}

int maple_destroy(maple_ctx *c)
{
    if (!c) return MAPLE_OK;
    free(c->refIdx); free(c->siteRates); free(c->errorRates); free(c->cumRate); free(c->cumErr); free(c->rflec); free(c->cumBases);
    free(c->ent); free(c->off); free(c->len); free(c->mut3); free(c->moff); free(c->mcnt);
    free(c->tup); free(c->tc0); free(c->tc1); free(c->tminor); free(c->tdist); free(c->ttip); free(c->tmut);
    for (int k = 0; k < 4; k++) free(c->tcol[k]);
    free(c);
    return MAPLE_OK;
}

int maple_set_fatal_policy(maple_ctx *c, int tolerate) { if (!c) return MAPLE_ERR_ARG; c->tolerate = tolerate != 0; return MAPLE_OK; }

/* updateMutMatrices (M:6350-6370) and updateErrorRates (M:6373-6390) */
int maple_set_model(maple_ctx *c, const double *Q16, const double *siteRates, int usingErrorRate, double errorRateGlobal,
                    const double *errorRates)
{
    if (!c || !Q16) return MAPLE_ERR_ARG;
    OModel *m = &c->m;
    const int L = m->lRef;
    for (int i = 0; i < 16; i++) m->Q[i] = Q16[i];
    free(c->siteRates); free(c->errorRates); free(c->cumRate); free(c->cumErr); free(c->rflec);
    c->siteRates = c->errorRates = c->cumErr = c->rflec = NULL;
    m->useRateVariation = siteRates != NULL;
    if (siteRates) { c->siteRates = (double *)malloc((size_t)L * sizeof(double)); memcpy(c->siteRates, siteRates, (size_t)L * sizeof(double)); }
    m->siteRates = c->siteRates;
    c->cumRate = (double *)calloc((size_t)L + 1, sizeof(double));
    for (int i = 0; i < L; i++) {
        const int r = c->refIdx[i];
        const double d = siteRates ? Q16[r * 4 + r] * siteRates[i] : Q16[r * 4 + r];
        c->cumRate[i + 1] = c->cumRate[i] + d;
    }
    m->cumulativeRate = c->cumRate;
    m->usingErrorRate = usingErrorRate != 0;
    m->errorRateSiteSpecific = usingErrorRate && errorRates != NULL;
    m->errorRate = errorRateGlobal;
    m->totError = 0.0;
    m->errorRates = NULL; m->cumulativeErrorRate = NULL;
    if (usingErrorRate) {
        c->cumErr = (double *)calloc((size_t)L + 1, sizeof(double));
        c->rflec = (double *)calloc((size_t)L + 1, sizeof(double));
        if (errorRates) { c->errorRates = (double *)malloc((size_t)L * sizeof(double)); memcpy(c->errorRates, errorRates, (size_t)L * sizeof(double)); }
        m->errorRates = c->errorRates;
        double tot = 0.0;
        for (int i = 0; i < L; i++) {
            const double e = errorRates ? errorRates[i] : errorRateGlobal;
            c->cumErr[i + 1] = c->cumErr[i] + e;
            tot -= e;
            c->rflec[i + 1] = c->rflec[i] + log(m->rootFreqs[c->refIdx[i]] * (1.0 - 1.33333 * e) + 0.333333 * e);   /* M:6384 (0.333333 there) */
        }
        (void)tot;
        m->totError = errorRates ? -c->cumErr[L] : -errorRateGlobal * L;         /* M:6385 / 6390 */
        m->cumulativeErrorRate = errorRates ? c->cumErr : NULL;
    }
    c->model_set = 1;
    return MAPLE_OK;
}

int maple_get_model(maple_ctx *c, double *cumulativeRate, double *cumulativeErrorRate, double *totError)
{
    if (!c) return MAPLE_ERR_ARG;
    if (!c->model_set) return fail(c, MAPLE_ERR_STATE, "maple_set_model has not been called");
    const size_t n = (size_t)c->m.lRef + 1;
    if (cumulativeRate) memcpy(cumulativeRate, c->cumRate, n * sizeof(double));
    if (cumulativeErrorRate) { if (c->cumErr) memcpy(cumulativeErrorRate, c->cumErr, n * sizeof(double)); else memset(cumulativeErrorRate, 0, n * sizeof(double)); }
    if (totError) *totError = c->m.totError;
    return MAPLE_OK;
}

/* ---- lists ------------------------------------------------------------------------------------------------------ */
static int room(maple_ctx *c, long long nEnt, long long nLists)
{
    if (c->usedEnt + nEnt > c->capEnt) {
        long long cap = c->capEnt ? c->capEnt : 1 << 16;
        while (cap < c->usedEnt + nEnt) cap *= 2;
        OEntry *p = (OEntry *)realloc(c->ent, (size_t)cap * sizeof(OEntry));
        if (!p) return fail(c, MAPLE_ERR_NOMEM, "out of memory for %lld entries", cap);
        c->ent = p; c->capEnt = cap;
    }
    if (c->nLists + nLists > c->capLists) {
        long long cap = c->capLists ? c->capLists : 1 << 12;
        while (cap < c->nLists + nLists) cap *= 2;
        long long *o = (long long *)realloc(c->off, (size_t)cap * sizeof(long long));
        int *l = (int *)realloc(c->len, (size_t)cap * sizeof(int));
        if (!o || !l) return fail(c, MAPLE_ERR_NOMEM, "out of memory for %lld lists", cap);
        c->off = o; c->len = l; c->capLists = cap;
    }
    return MAPLE_OK;
}
/* a finished list at c->ent + c->usedEnt becomes list id c->nLists */
static int commit(maple_ctx *c, int n) { c->off[c->nLists] = c->usedEnt; c->len[c->nLists] = n; c->usedEnt += n; return (int)c->nLists++; }
static int ok_id(const maple_ctx *c, int id) { return id >= 0 && id < c->nLists; }
#define LIST(c, id) ((c)->ent + (c)->off[id])

int maple_lists_upload(maple_ctx *c, int32_t n, const int64_t *ent_off, const int32_t *pos, const uint32_t *meta,
                       const int64_t *aux_off, const double *aux, int32_t *first_id)
{
    if (!c || n < 0 || !ent_off || !aux_off || !first_id) return MAPLE_ERR_ARG;
    int rc = room(c, ent_off[n], n);
    if (rc) return rc;
    *first_id = (int32_t)c->nLists;
    for (int i = 0; i < n; i++) {
        const long long one[2] = {0, ent_off[i + 1] - ent_off[i]}, oneA[2] = {0, aux_off[i + 1] - aux_off[i]};
        omo_entries_from_packed(1, one, pos + ent_off[i], meta + ent_off[i], oneA, aux + aux_off[i], c->m.usingErrorRate,
                                c->ent + c->usedEnt, 1);
        commit(c, (int)one[1]);
    }
    return MAPLE_OK;
}

int maple_lists_update(maple_ctx *c, int32_t n, const int32_t *ids, const int64_t *ent_off, const int32_t *pos,
                       const uint32_t *meta, const int64_t *aux_off, const double *aux)
{
    if (!c || n < 0 || !ids || !ent_off || !aux_off) return MAPLE_ERR_ARG;
    for (int i = 0; i < n; i++) if (!ok_id(c, ids[i])) return fail(c, MAPLE_ERR_ARG, "list %d is not a list id", ids[i]);
    int rc = room(c, ent_off[n], 0);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        const long long one[2] = {0, ent_off[i + 1] - ent_off[i]}, oneA[2] = {0, aux_off[i + 1] - aux_off[i]};
        omo_entries_from_packed(1, one, pos + ent_off[i], meta + ent_off[i], oneA, aux + aux_off[i], c->m.usingErrorRate,
                                c->ent + c->usedEnt, 1);
        c->off[ids[i]] = c->usedEnt; c->len[ids[i]] = (int)one[1];
        c->usedEnt += one[1];
    }
    return MAPLE_OK;
}

static int n_aux_of(const OEntry *e, int u)
{
    if (e->type == 5) return 0;
    if (e->type == 6) return 4 + (e->len == 4 ? 1 : 0);
    return e->len - 2 - ((u && e->len > 2) ? 1 : 0);
}

int maple_lists_sizes(maple_ctx *c, int32_t n, const int32_t *ids, int32_t *n_ent, int32_t *n_aux)
{
    if (!c || n < 0 || !ids || !n_ent || !n_aux) return MAPLE_ERR_ARG;
    for (int i = 0; i < n; i++) {
        if (!ok_id(c, ids[i])) return fail(c, MAPLE_ERR_ARG, "list %d is not a list id", ids[i]);
        const OEntry *L = LIST(c, ids[i]);
        int na = 0;
        for (int k = 0; k < c->len[ids[i]]; k++) na += n_aux_of(&L[k], c->m.usingErrorRate);
        n_ent[i] = c->len[ids[i]]; n_aux[i] = na;
    }
    return MAPLE_OK;
}

int maple_lists_download(maple_ctx *c, int32_t n, const int32_t *ids, const int64_t *ent_off, int32_t *pos, uint32_t *meta,
                         const int64_t *aux_off, double *aux)
{
    if (!c || n < 0 || !ids || !ent_off || !aux_off) return MAPLE_ERR_ARG;
    const int u = c->m.usingErrorRate;
    for (int i = 0; i < n; i++) {
        if (!ok_id(c, ids[i])) return fail(c, MAPLE_ERR_ARG, "list %d is not a list id", ids[i]);
        const OEntry *L = LIST(c, ids[i]);
        int p = 0;
        uint32_t ao = 0;
        double *a = aux + aux_off[i];
        for (int k = 0; k < c->len[ids[i]]; k++) {
            const OEntry *e = &L[k];
            uint32_t mt = (uint32_t)e->type;
            if (e->type == 4 || e->type == 5) p = e->x; else { p += 1; mt |= (uint32_t)e->x << 3; }
            const uint32_t at = ao;
            if (e->type == 6) {
                if (e->len == 4) { mt |= MAPLE_META_HASD0; a[ao++] = e->d0; }
                for (int j = 0; j < 4; j++) a[ao++] = e->vec[j];
            } else if (e->type != 5) {
                const int tails = e->len - 2 - ((u && e->len > 2) ? 1 : 0);
                if (tails >= 1) { mt |= MAPLE_META_HASD0; a[ao++] = e->d0; }
                if (tails >= 2) { mt |= MAPLE_META_HASD1; a[ao++] = e->d1; }
                if (u && e->len > 2 && e->flag) mt |= MAPLE_META_FLAG;
            }
            pos[ent_off[i] + k] = p;
            meta[ent_off[i] + k] = mt | (at << 8);
        }
    }
    return MAPLE_OK;
}

int maple_arena_mark(maple_ctx *c, int64_t *mark)
{
    if (!c || !mark) return MAPLE_ERR_ARG;
    *mark = (int64_t)c->nLists | ((int64_t)c->nMut << 40);
    return MAPLE_OK;
}
int maple_arena_release(maple_ctx *c, int64_t markBoth)
{
    if (!c) return MAPLE_ERR_ARG;
    const int64_t mark = markBoth & (((int64_t)1 << 40) - 1), mmark = markBoth >> 40;
    if (mark > c->nLists || mmark > c->nMut) return MAPLE_ERR_ARG;
    if (mmark < c->nMut) { c->usedMut = c->moff[mmark]; c->nMut = mmark; }
    if (mark < c->nLists) {
        long long ue = c->off[mark];
        for (long long i = 0; i < mark; i++) if (c->off[i] + c->len[i] > ue) ue = c->off[i] + c->len[i];   /* (lists updated later keep their room) */
        c->usedEnt = ue; c->nLists = mark;
    }
    return MAPLE_OK;
}
int maple_arena_stats(maple_ctx *c, int64_t *n_lists, int64_t *n_entries, int64_t *n_aux, int64_t *cap_entries)
{
    if (!c) return MAPLE_ERR_ARG;
    if (n_lists) *n_lists = c->nLists;
    if (n_entries) *n_entries = c->usedEnt;
    if (n_aux) *n_aux = 0;
    if (cap_entries) *cap_entries = c->capEnt;
    return MAPLE_OK;
}

int maple_mutations_upload(maple_ctx *c, int32_t n, const int64_t *off, const int32_t *mut3, int32_t *first_id)
{
    if (!c || n < 0 || !off || !first_id) return MAPLE_ERR_ARG;
    const long long tot = off[n];
    if (c->usedMut + tot > c->capMut) {
        long long cap = c->capMut ? c->capMut : 1 << 12;
        while (cap < c->usedMut + tot) cap *= 2;
        c->mut3 = (int *)realloc(c->mut3, (size_t)cap * 3 * sizeof(int)); c->capMut = cap;
    }
    if (c->nMut + n > c->capMutLists) {
        long long cap = c->capMutLists ? c->capMutLists : 1 << 10;
        while (cap < c->nMut + n) cap *= 2;
        c->moff = (long long *)realloc(c->moff, (size_t)cap * sizeof(long long)); c->mcnt = (int *)realloc(c->mcnt, (size_t)cap * sizeof(int));
        c->capMutLists = cap;
    }
    *first_id = (int32_t)c->nMut;
    for (int i = 0; i < n; i++) {
        const int cnt = (int)(off[i + 1] - off[i]);
        if (cnt) memcpy(c->mut3 + 3 * c->usedMut, mut3 + 3 * off[i], (size_t)cnt * 3 * sizeof(int));
        c->moff[c->nMut] = c->usedMut; c->mcnt[c->nMut] = cnt;
        c->usedMut += cnt; c->nMut++;
    }
    return MAPLE_OK;
}

/* ---- batched operators ------------------------------------------------------------------------------------------ */
#define NEED_MODEL(c) do { if (!(c)->model_set) return fail((c), MAPLE_ERR_STATE, "maple_set_model has not been called"); } while (0)
#define NEED_ID(c, id, what) do { if (!ok_id((c), (id))) return fail((c), MAPLE_ERR_ARG, "%s %d is not a list id", (what), (id)); } while (0)

int maple_append_batch(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl, const uint8_t *tip, const double *bl, double *out)
{
    if (!c || n < 0 || !pl || !cl || !tip || !bl || !out) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, pl[i], "parentList"); NEED_ID(c, cl[i], "childList");
        if (omo_appendProbNode(&c->m, LIST(c, pl[i]), c->len[pl[i]], LIST(c, cl[i]), c->len[cl[i]], tip[i], bl[i], &out[i]) < 0)
            return fail(c, MAPLE_ERR_FATAL, "appendProbNode: item %d hit a state the reference raises on", i);
    }
    return MAPLE_OK;
}

int maple_merge_batch(maple_ctx *c, int32_t n, const int32_t *l1, const double *b1, const uint8_t *t1, const int32_t *l2, const double *b2,
                      const uint8_t *t2, const uint8_t *ud, const int32_t *nm1, const int32_t *nm2, int32_t *outList, double *outLK)
{
    if (!c || n < 0 || !l1 || !b1 || !t1 || !l2 || !b2 || !t2 || !ud || !outList) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, l1[i], "list1"); NEED_ID(c, l2[i], "list2");
        const int n1 = c->len[l1[i]], n2 = c->len[l2[i]];
        int rc = room(c, n1 + n2 + 2, 1);
        if (rc) return rc;
        double lk = 0.0;
        const int r = omo_mergeVectors(&c->m, LIST(c, l1[i]), n1, b1[i], t1[i], LIST(c, l2[i]), n2, b2[i], t2[i], outLK != NULL, ud[i],
                                       nm1 ? nm1[i] : 0, nm2 ? nm2[i] : 0, c->ent + c->usedEnt, &lk);
        if (r == -1) { outList[i] = -1; if (outLK) outLK[i] = 0.0; continue; }
        if (r < 0) { if (c->tolerate) { outList[i] = -2; continue; } return fail(c, MAPLE_ERR_FATAL, "mergeVectors: item %d hit a state the reference raises on", i); }
        outList[i] = commit(c, r);
        if (outLK) outLK[i] = lk;
    }
    return MAPLE_OK;
}

int maple_blen_batch(maple_ctx *c, int32_t n, const int32_t *pl, const int32_t *cl, const uint8_t *tip, double *t, uint8_t *isFalse)
{
    if (!c || n < 0 || !pl || !cl || !tip || !t || !isFalse) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, pl[i], "parentList"); NEED_ID(c, cl[i], "childList");
        const int nP = c->len[pl[i]], nC = c->len[cl[i]];
        double *scratch = (double *)malloc((size_t)(nP + nC + 4) * sizeof(double));
        int f = 0;
        const int rc = omo_estimateBranchLength(&c->m, LIST(c, pl[i]), nP, LIST(c, cl[i]), nC, tip[i], &t[i], &f, scratch);
        free(scratch);
        if (rc < 0) return fail(c, MAPLE_ERR_FATAL, "estimateBranchLengthWithDerivative: item %d", i);
        isFalse[i] = (uint8_t)f;
        if (f) t[i] = 0.0;
    }
    return MAPLE_OK;
}

int maple_differ_batch(maple_ctx *c, int32_t n, const int32_t *l1, const int32_t *l2, uint8_t *out)
{
    if (!c || n < 0 || !l1 || !l2 || !out) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, l1[i], "list1");
        if (l2[i] == -1) { out[i] = 1; continue; }
        NEED_ID(c, l2[i], "list2");
        out[i] = (uint8_t)(omo_areVectorsDifferent(&c->m, LIST(c, l1[i]), c->len[l1[i]], LIST(c, l2[i]), c->len[l2[i]]) != 0);
    }
    return MAPLE_OK;
}

int maple_pass_branch_batch(maple_ctx *c, int32_t n, const int32_t *l, const int32_t *ml, const uint8_t *up, int32_t *outList)
{
    if (!c || n < 0 || !l || !ml || !up || !outList) return MAPLE_ERR_ARG;
    for (int i = 0; i < n; i++) {
        NEED_ID(c, l[i], "list");
        if (ml[i] < 0 || ml[i] >= c->nMut) return fail(c, MAPLE_ERR_ARG, "mutation list %d is not an id", ml[i]);
        const int nl = c->len[l[i]], cnt = c->mcnt[ml[i]];
        int rc = room(c, nl + 2 * cnt + 2, 1);
        if (rc) return rc;
        const int r = omo_passGenomeListThroughBranch(&c->m, LIST(c, l[i]), nl, c->mut3 + 3 * c->moff[ml[i]], cnt, up[i], c->ent + c->usedEnt);
        if (r < 0) return fail(c, MAPLE_ERR_FATAL, "passGenomeListThroughBranch: item %d", i);
        outList[i] = commit(c, r);
    }
    return MAPLE_OK;
}

int maple_shorten_batch(maple_ctx *c, int32_t n, const int32_t *l, int32_t *outList)
{
    if (!c || n < 0 || !l || !outList) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, l[i], "list");
        const int nl = c->len[l[i]];
        int rc = room(c, nl, 1);
        if (rc) return rc;
        memcpy(c->ent + c->usedEnt, LIST(c, l[i]), (size_t)nl * sizeof(OEntry));
        outList[i] = commit(c, omo_shorten(&c->m, c->ent + c->usedEnt, nl));
        /* (commit reserved nl entries' worth of room at most; the shortened list is a prefix of it) */
    }
    return MAPLE_OK;
}

int maple_root_vector_batch(maple_ctx *c, int32_t n, const int32_t *l, const double *bLen, const uint8_t *isFromTip, const int64_t *pathOff,
                            const int32_t *pathMutLists, int32_t *outList)
{
    if (!c || n < 0 || !l || !bLen || !isFromTip || !pathOff || !outList) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, l[i], "list");
        const int nPath = (int)(pathOff[i + 1] - pathOff[i]);
        int tot = 0;
        for (int k = 0; k < nPath; k++) tot += c->mcnt[pathMutLists[pathOff[i] + k]];
        int *m3 = (int *)malloc((size_t)(tot + 1) * 3 * sizeof(int)), *po = (int *)malloc((size_t)(nPath + 1) * sizeof(int));
        int at = 0;
        for (int k = 0; k < nPath; k++) {
            const int id = pathMutLists[pathOff[i] + k];
            po[k] = at;
            memcpy(m3 + 3 * at, c->mut3 + 3 * c->moff[id], (size_t)c->mcnt[id] * 3 * sizeof(int));
            at += c->mcnt[id];
        }
        po[nPath] = at;
        const int nl = c->len[l[i]], cap = nl + 4 * tot + 8;
        int rc = room(c, cap, 1);
        OEntry *tmp = (OEntry *)malloc((size_t)cap * sizeof(OEntry));
        int r = rc ? -9 : omo_rootVector(&c->m, LIST(c, l[i]), nl, bLen[i], isFromTip[i], m3, po, nPath, c->ent + c->usedEnt, tmp, cap);
        free(m3); free(po); free(tmp);
        if (rc) return rc;
        if (r < 0) return fail(c, MAPLE_ERR_FATAL, "rootVector: item %d", i);
        outList[i] = commit(c, r);
    }
    return MAPLE_OK;
}

int maple_root_prob_batch(maple_ctx *c, int32_t n, const int32_t *l, double *out)
{
    if (!c || n < 0 || !l || !out) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    const int po[1] = {0};
    for (int i = 0; i < n; i++) {
        NEED_ID(c, l[i], "list");
        const int nl = c->len[l[i]], cap = nl + 8;
        OEntry *a = (OEntry *)malloc((size_t)cap * sizeof(OEntry)), *b = (OEntry *)malloc((size_t)cap * sizeof(OEntry));
        const int rc = omo_findProbRoot(&c->m, LIST(c, l[i]), nl, NULL, po, 0, c->cumBases, c->rflec, a, b, cap, &out[i]);
        free(a); free(b);
        if (rc < 0) return fail(c, MAPLE_ERR_FATAL, "findProbRoot: item %d", i);
    }
    return MAPLE_OK;
}

int maple_evaluate_placement_batch(maple_ctx *c, int32_t n, const int32_t *mid, const int32_t *down, const int32_t *up, const double *distance,
                                   const int32_t *rem, const uint8_t *isRemovedTip, const uint8_t *fromTip1, double *out4)
{
    if (!c || n < 0 || !mid || !down || !up || !distance || !rem || !isRemovedTip || !fromTip1 || !out4) return MAPLE_ERR_ARG;
    NEED_MODEL(c);
    for (int i = 0; i < n; i++) {
        NEED_ID(c, mid[i], "midTot"); NEED_ID(c, down[i], "downVect"); NEED_ID(c, up[i], "upVect"); NEED_ID(c, rem[i], "removedPartials");
        const int cap = c->len[down[i]] + c->len[up[i]] + c->len[rem[i]] + 4;
        OEntry *tmp = (OEntry *)malloc((size_t)3 * cap * sizeof(OEntry));
        double *scratch = (double *)malloc((size_t)(c->len[mid[i]] + cap + 4) * sizeof(double));
        const int rc = omo_evaluatePlacement(&c->m, LIST(c, mid[i]), c->len[mid[i]], LIST(c, down[i]), c->len[down[i]], LIST(c, up[i]), c->len[up[i]],
                                             distance[i], LIST(c, rem[i]), c->len[rem[i]], isRemovedTip[i], fromTip1[i], c->params.defaultBLen,
                                             out4 + 4 * i, tmp, cap, scratch);
        free(tmp); free(scratch);
        if (rc < 0) return fail(c, MAPLE_ERR_FATAL, "evaluatePlacement: item %d", i);
    }
    return MAPLE_OK;
}

/* ---- tree mirror + SPR search batch --------------------------------------------------------------------------------- */
int maple_tree_upload(maple_ctx *c, int32_t n, int32_t root, const int32_t *up, const int32_t *c0, const int32_t *c1, const double *dist,
                      const uint8_t *isTip, const int32_t *lower, const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp,
                      const int32_t *mutList)
{
    if (!c || n <= 0 || root < 0 || root >= n || !up || !c0 || !c1 || !dist || !isTip || !lower || !upRight || !upLeft || !totUp || !mutList)
        return MAPLE_ERR_ARG;
    free(c->tup); free(c->tc0); free(c->tc1); free(c->tminor); free(c->tdist); free(c->ttip); free(c->tmut);
    for (int k = 0; k < 4; k++) free(c->tcol[k]);
    const size_t b = (size_t)n * sizeof(int);
    c->tup = (int *)malloc(b); c->tc0 = (int *)malloc(b); c->tc1 = (int *)malloc(b); c->tminor = (int *)malloc(b); c->tmut = (int *)malloc(b);
    c->tdist = (double *)malloc((size_t)n * sizeof(double)); c->ttip = (unsigned char *)malloc((size_t)n);
    for (int k = 0; k < 4; k++) c->tcol[k] = (int *)malloc(b);
    const int32_t *cols[4] = {lower, upRight, upLeft, totUp};
    for (int i = 0; i < n; i++) {
        c->tup[i] = up[i]; c->tc0[i] = c0[i]; c->tc1[i] = c1[i]; c->tdist[i] = dist[i]; c->ttip[i] = isTip[i]; c->tmut[i] = mutList[i];
        /* (isTip = leaf without minor sequences: a leaf that is no tip has some) */
        c->tminor[i] = (c0[i] < 0 && !isTip[i]) ? 1 : 0;
        for (int k = 0; k < 4; k++) {
            if (cols[k][i] >= 0 && !ok_id(c, cols[k][i])) return fail(c, MAPLE_ERR_ARG, "node %d: list %d is not a list id", i, cols[k][i]);
            c->tcol[k][i] = cols[k][i];
        }
        if (mutList[i] >= c->nMut) return fail(c, MAPLE_ERR_ARG, "node %d: mutation list %d is not an id", i, mutList[i]);
    }
    c->tn = n; c->troot = root; c->tree_set = 1;
    return MAPLE_OK;
}

int maple_spr_search_batch(maple_ctx *c, int32_t n, const int32_t *nodes, const maple_search_params *sp, int32_t ws_entries_per_lane,
                           int32_t *bestNode, double *bestScore, double *blen3, int32_t *placement, double *improvement, double *currentLK,
                           int32_t *nAppend, int32_t *status, int32_t *outRprList)
{
    (void)ws_entries_per_lane;
    if (!c || n < 0 || !nodes || !sp || !bestNode || !bestScore || !blen3 || !placement || !improvement || !currentLK || !nAppend || !status)
        return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    NEED_MODEL(c);
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    const int N = c->tn;
    /* the tree in the oracle's layout: the four list columns as (start, len) into the arena, mutations as a CSR by node */
    long long *start[4], *moff = (long long *)calloc((size_t)N + 1, sizeof(long long));
    int *len[4];
    for (int k = 0; k < 4; k++) {
        start[k] = (long long *)calloc((size_t)N, sizeof(long long)); len[k] = (int *)calloc((size_t)N, sizeof(int));
        for (int i = 0; i < N; i++) if (c->tcol[k][i] >= 0) { start[k][i] = c->off[c->tcol[k][i]]; len[k][i] = c->len[c->tcol[k][i]]; }
    }
    for (int i = 0; i < N; i++) moff[i + 1] = moff[i] + (c->tmut[i] >= 0 ? c->mcnt[c->tmut[i]] : 0);
    int *m3 = (int *)malloc((size_t)(moff[N] + 1) * 3 * sizeof(int));
    for (int i = 0; i < N; i++)
        if (c->tmut[i] >= 0 && c->mcnt[c->tmut[i]]) memcpy(m3 + 3 * moff[i], c->mut3 + 3 * c->moff[c->tmut[i]], (size_t)c->mcnt[c->tmut[i]] * 3 * sizeof(int));
    OTree t;
    t.n = N; t.root = c->troot; t.up = c->tup; t.c0 = c->tc0; t.c1 = c->tc1; t.dist = c->tdist; t.nMinor = c->tminor; t.ent = c->ent;
    for (int k = 0; k < 4; k++) { t.start[k] = start[k]; t.len[k] = len[k]; }
    t.mut3 = m3; t.mutOff = moff;
    OSearchParams p;
    p.strict = sp->strictTopologyStopRules; p.allowedFails = sp->allowedFailsTopology; p.thrLKtopology = sp->thresholdLogLKtopology;
    p.thrPlacement = sp->thresholdTopologyPlacement; p.thrOptTopo = sp->thresholdLogLKoptimizationTopology;
    p.thrConsec = sp->thresholdLogLKconsecutivePlacement; p.effNon0 = sp->effectivelyNon0BLen; p.defaultBLen = c->params.defaultBLen;
    OSearchResult *res = (OSearchResult *)calloc((size_t)n, sizeof(OSearchResult));
    const int rprCap = 8192;
    OEntry *rprBuf = outRprList ? (OEntry *)malloc((size_t)n * rprCap * sizeof(OEntry)) : NULL;
    if (outRprList) for (int i = 0; i < n; i++) { res[i].rpr = rprBuf + (size_t)i * rprCap; res[i].rprCap = rprCap; }
    const size_t arenaBytes = (size_t)512 << 20;
    void *arena = malloc(arenaBytes);
    int *nd = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; i++) nd[i] = nodes[i];
    const int rc = omo_sprWorker(&c->m, &t, &p, n, nd, res, arena, arenaBytes);
    for (int i = 0; i < n && rc >= 0; i++) {
        bestNode[i] = res[i].bestNode; bestScore[i] = res[i].bestScore; placement[i] = res[i].placement; improvement[i] = res[i].improvement;
        currentLK[i] = res[i].currentLK; nAppend[i] = res[i].nAppend; status[i] = res[i].status;
        for (int k = 0; k < 3; k++) blen3[3 * i + k] = res[i].blen[k];
    }
    int rc2 = MAPLE_OK;
    if (outRprList && rc >= 0)
        for (int i = 0; i < n; i++) {
            outRprList[i] = -1;
            if (res[i].status != 0 || res[i].rprN <= 0) continue;
            rc2 = room(c, res[i].rprN, 1);
            if (rc2) break;
            memcpy(c->ent + c->usedEnt, res[i].rpr, (size_t)res[i].rprN * sizeof(OEntry));
            outRprList[i] = commit(c, res[i].rprN);
        }
    free(arena); free(nd); free(res); free(rprBuf); free(m3); free(moff);
    for (int k = 0; k < 4; k++) { free(start[k]); free(len[k]); }
    if (rc < 0) return fail(c, MAPLE_ERR_FATAL, "SPR search: the oracle's worker failed (%d)", rc);
    return rc2;
}
