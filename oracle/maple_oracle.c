/*
 * oracle/maple_oracle.c -- TEST INFRASTRUCTURE ONLY (see maple_oracle.h).
 *
 * CPU restatement of the reference's genome-list functions.  Every function
 * names the reference lines it follows (M: = MAPLEv0.7.5.4.py).  Arithmetic
 * keeps the reference's operand order; build with -ffp-contract=off.
 *
 * Parity: pinned to golden call records harvested from the reference
 * (tests/golden/, tests/test_oracle_golden.py).
 */
#include "maple_oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

#define U_(m) ((m)->usingErrorRate)
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline double pymin(double a, double b) { return (b < a) ? b : a; } /* Python min(a,b) */

/* per-site matrix: mutMatrices[pos] = Q*siteRates[pos] (M:6361-6366) or Q */
static inline void site_mat(const OModel *m, int pos, double *M)
{
    if (m->useRateVariation) {
        double r = m->siteRates[pos];
        for (int k = 0; k < 16; k++) M[k] = m->Q[k] * r;
    } else {
        memcpy(M, m->Q, 16 * sizeof(double));
    }
}

static inline OEntry mk(int type, int x, int len, double d0, double d1, int flag)
{
    OEntry e;
    memset(&e, 0, sizeof e);
    e.type = type; e.x = x; e.len = len; e.d0 = d0; e.d1 = d1; e.flag = flag;
    return e;
}
static inline OEntry mkO(int ref, int hasD0, double d0, const double *v)
{
    OEntry e = mk(6, ref, hasD0 ? 4 : 3, d0, 0.0, 0);
    for (int i = 0; i < 4; i++) e.vec[i] = v[i];
    return e;
}

/* ---- getPartialVec, M:4073-4141 ------------------------------------------ */
void omo_getPartialVec(const OModel *m, int i12, double totLen, const double *M, double errorRate,
                       const double *vect, int upNode, int flag, double *out)
{
    static const double quarter[4] = {0.25, 0.25, 0.25, 0.25};
    if (i12 == 6) {
        if (totLen == 0.0) { memcpy(out, vect, 4 * sizeof(double)); return; }   /* M:4086-4088 */
        double nv[4];
        for (int i = 0; i < 4; i++) {
            double tot = 0.0;
            for (int j = 0; j < 4; j++)
                tot += (upNode ? M[j * 4 + i] : M[i * 4 + j]) * vect[j];        /* M:4093 / 4103 */
            tot *= totLen;
            tot += vect[i];
            if (tot < 0) { memcpy(out, quarter, sizeof quarter); return; }
            nv[i] = tot;
        }
        memcpy(out, nv, sizeof nv);
    } else if (U_(m) && flag) {                                                  /* M:4109-4125 */
        double nv[4], mv[4];
        for (int i = 0; i < 4; i++) nv[i] = errorRate * 0.33333;
        nv[i12] = 1.0 - errorRate;
        if (totLen == 0.0) { memcpy(out, nv, sizeof nv); return; }
        for (int j = 0; j < 4; j++) {
            double tot = 0.0;
            for (int i = 0; i < 4; i++) tot += M[j * 4 + i] * nv[i];
            tot *= totLen;
            tot += nv[j];
            if (tot < 0) { memcpy(out, quarter, sizeof quarter); return; }
            mv[j] = tot;
        }
        memcpy(out, mv, sizeof mv);
    } else {                                                                     /* M:4126-4141 */
        double nv[4] = {0.0, 0.0, 0.0, 0.0};
        if (totLen == 0.0) { nv[i12] += 1.0; memcpy(out, nv, sizeof nv); return; }
        for (int i = 0; i < 4; i++) nv[i] = (upNode ? M[i12 * 4 + i] : M[i * 4 + i12]) * totLen;
        nv[i12] += 1.0;
        if (nv[i12] < 0) { memcpy(out, quarter, sizeof quarter); return; }
        memcpy(out, nv, sizeof nv);
    }
}

/* ---- simplify, M:3697-3717 ------------------------------------------------ */
int omo_simplify(const OModel *m, const double *vec, int refA, int *state)
{
    double maxP = 0.0; int maxI = 0, numA = 0;
    double thr4 = (m->thresholdProb * m->thresholdProb) * (m->thresholdProb * m->thresholdProb); /* M:3692-3693 */
    for (int i = 0; i < 4; i++) {
        if (vec[i] > maxP) { maxP = vec[i]; maxI = i; }
        if (vec[i] > m->thresholdProb) numA++;
    }
    if (maxP < thr4) return -2;
    if (numA == 1) *state = (maxI == refA) ? 4 : maxI;
    else *state = 6;
    return 0;
}

/* ---- shorten, M:3721-3745 (in place; returns the new length) --------------- */
int omo_shorten(const OModel *m, OEntry *vec, int n)
{
    /* the reference pops vec[index] when vec[index+1] can absorb it */
    int index = 0;
    OEntry old = vec[0];
    while (index < n - 1) {
        OEntry nw = vec[index + 1];
        int pop = 0;
        if (nw.type == 4 && old.type == 4 && nw.len == old.len) {
            /* element [2], [3], [4] of the tuples: d0, then d1-or-flag, then flag */
            if (nw.len == 2) pop = 1;
            else if (fabs(nw.d0 - old.d0) > m->thresholdProb) pop = 0;
            else if (nw.len == 3) pop = 1;
            else {
                /* element [3]: d1 when len==4 without error model or len==5; flag when len==4 with it */
                double a3, b3;
                if (U_(m) && nw.len == 4) { a3 = nw.flag; b3 = old.flag; }
                else { a3 = nw.d1; b3 = old.d1; }
                if (fabs(a3 - b3) > m->thresholdProb) pop = 0;
                else if (nw.len == 4 || nw.flag == old.flag) pop = 1;
                else pop = 0;
            }
        }
        if (pop) {
            memmove(&vec[index], &vec[index + 1], (size_t)(n - index - 1) * sizeof(OEntry));
            n--;
            /* entryOld keeps its value (M:3728-3729: pop without refreshing entryOld) */
        } else {
            index++;
            old = vec[index];
        }
    }
    return n;
}

/* ---- passGenomeListThroughBranch, M:3749-3877 ------------------------------ */
int omo_passGenomeListThroughBranch(const OModel *m, const OEntry *pv, int n, const int *mut, int lMut,
                                    int dirIsUp, OEntry *out)
{
    int lRef = m->lRef;
    int iM = 0, iE = 0, lastPos = 0, no = 0;
    const OEntry *e = &pv[iE];
    (void)n;
    for (;;) {
        if (e->type == 5) {                                          /* M:3758-3768 */
            out[no++] = *e;
            lastPos = e->x;
            if (lastPos == lRef) break;
            while (iM < lMut && mut[iM * 3] <= lastPos) iM++;
            e = &pv[++iE];
        } else if (e->type < 4) {                                    /* M:3770-3806 */
            lastPos += 1;
            if (iM < lMut && mut[iM * 3] <= lastPos) {
                int from = mut[iM * 3 + 1], to = mut[iM * 3 + 2];
                int cmp = dirIsUp ? from : to;
                OEntry ne = *e;
                if (e->type == cmp) { ne.type = 4; ne.x = lastPos; }
                else { ne.type = e->type; ne.x = cmp; }
                iM++;
                out[no++] = ne;
            } else out[no++] = *e;
            if (lastPos == lRef) break;
            e = &pv[++iE];
        } else if (e->type == 4) {                                   /* M:3808-3852 */
            while (iM < lMut && mut[iM * 3] <= e->x) {
                if (mut[iM * 3] > lastPos + 1) {
                    lastPos = mut[iM * 3] - 1;
                    OEntry ne = *e; ne.x = lastPos;
                    out[no++] = ne;
                }
                lastPos += 1;
                OEntry ne = *e;
                if (dirIsUp) { ne.type = mut[iM * 3 + 2]; ne.x = mut[iM * 3 + 1]; }
                else { ne.type = mut[iM * 3 + 1]; ne.x = mut[iM * 3 + 2]; }
                iM++;
                out[no++] = ne;
            }
            if (lastPos < e->x) { lastPos = e->x; out[no++] = *e; }
            if (lastPos == lRef) break;
            e = &pv[++iE];
        } else {                                                     /* O, M:3854-3873 */
            lastPos += 1;
            if (iM < lMut && mut[iM * 3] <= lastPos) {
                OEntry ne = *e;
                ne.x = dirIsUp ? mut[iM * 3 + 1] : mut[iM * 3 + 2];
                iM++;
                out[no++] = ne;
            } else out[no++] = *e;
            if (lastPos == lRef) break;
            e = &pv[++iE];
        }
    }
    return no;
}

/* ---- appendProbNode, M:6505-6785 ------------------------------------------- */
int omo_appendProbNode(const OModel *m, const OEntry *P, int nP, const OEntry *C, int nC, int isTipC,
                       double bLen, double *outLK)
{
    const int U = U_(m), SS = m->errorRateSiteSpecific, RV = m->useRateVariation, lRef = m->lRef;
    const double *rf = m->rootFreqs;
    int ix1 = 0, ix2 = 0, pos = 0;
    double totalFactor = 1.0;
    const OEntry *e1 = &P[0], *e2 = &C[0];
    double contribLength = bLen;
    double Lkcost = bLen * m->globalTotRate;                         /* M:6541 */
    double M[16], tot2[4], tot3[4];
    double errorRate = m->errorRate;
    (void)nP; (void)nC;
    memcpy(M, m->Q, sizeof M);
    if (U && isTipC) Lkcost += m->totError;                          /* M:6542-6543 */
    for (;;) {
        if (e2->type == 5) {                                         /* M:6545-6560 */
            if (e1->type == 4 || e1->type == 5) {
                pos = imin(e1->x, e2->x);
                if (pos == lRef) break;
                if (e1->x == pos) e1 = &P[++ix1];
            } else {
                pos += 1;
                if (pos == lRef) break;
                e1 = &P[++ix1];
            }
            if (e2->x == pos) e2 = &C[++ix2];
        } else if (e1->type == 5) {                                  /* M:6562-6582 */
            if (e2->type == 4) {
                pos = imin(e1->x, e2->x);
                if (pos == lRef) break;
                if (e2->x == pos) e2 = &C[++ix2];
            } else {
                pos += 1;
                if (pos == lRef) break;
                e2 = &C[++ix2];
            }
            if (e1->x == pos) e1 = &P[++ix1];
        } else {
            if (e1->type != e2->type || e1->type == 6) {             /* M:6586-6599 */
                contribLength = bLen;
                if (e1->type < 5) {
                    if (e1->len == 3 + U) contribLength += e1->d0;
                    else if (e1->len == 4 + U) contribLength += e1->d1;
                } else if (e1->len == 4) contribLength += e1->d0;
                if (e2->type < 5) {
                    if (e2->len == 3 + U) contribLength += e2->d0;
                } else if (e2->len == 4) contribLength += e2->d0;
            }
            if (e1->type == 4) {
                if (e2->type == 4) {                                 /* M:6602-6608 */
                    pos = imin(e1->x, e2->x);
                    if (pos == lRef) break;
                    if (e2->x == pos) e2 = &C[++ix2];
                } else if (e2->type == 6) {                          /* M:6611-6638 */
                    if (RV) site_mat(m, pos, M);
                    int i1 = e2->x;
                    if (e2->vec[i1] > 0.02) totalFactor *= e2->vec[i1];
                    else {
                        double tot;
                        if (e1->len == 4 + U) {
                            int flag1 = (U && e1->len > 2 && e1->flag);
                            tot = 0.0;
                            if (U && SS) errorRate = m->errorRates[pos];
                            omo_getPartialVec(m, 6, contribLength, M, 0.0, e2->vec, 0, 0, tot3);
                            omo_getPartialVec(m, i1, e1->d0, M, errorRate, NULL, 0, flag1, tot2);
                            for (int i = 0; i < 4; i++) tot += tot3[i] * tot2[i] * rf[i];
                            tot /= rf[i1];
                        } else {
                            if (contribLength != 0.0) {
                                omo_getPartialVec(m, 6, contribLength, M, 0.0, e2->vec, 0, 0, tot3);
                                tot = tot3[i1];
                            } else tot = e2->vec[i1];
                        }
                        totalFactor *= tot;
                    }
                    pos += 1;
                    if (pos == lRef) break;
                    e2 = &C[++ix2];
                } else {                                             /* M:6640-6668 */
                    int flag2 = (U && (isTipC || (e2->len > 2 && e2->flag)));
                    if (RV) site_mat(m, pos, M);
                    if (e1->len == 4 + U) {
                        int flag1 = (U && e1->len > 2 && e1->flag);
                        int i1 = e2->x, i2 = e2->type;
                        if (U && SS) errorRate = m->errorRates[pos];
                        omo_getPartialVec(m, i2, contribLength, M, errorRate, NULL, 0, flag2, tot3);
                        omo_getPartialVec(m, i1, e1->d0, M, errorRate, NULL, 0, flag1, tot2);
                        double tot = 0.0;
                        for (int i = 0; i < 4; i++) tot += tot3[i] * tot2[i] * rf[i];
                        totalFactor *= tot / rf[i1];
                    } else {
                        if (flag2) {
                            if (U && SS) errorRate = m->errorRates[pos];
                            totalFactor *= pymin(0.25, M[e2->x * 4 + e2->type] * contribLength) + errorRate * 0.33333;
                        } else {
                            if (contribLength != 0.0)
                                totalFactor *= pymin(0.25, M[e2->x * 4 + e2->type] * contribLength);
                            else { *outLK = -INFINITY; return 0; }
                        }
                    }
                    pos += 1;
                    if (pos == lRef) break;
                    e2 = &C[++ix2];
                }
                if (e1->x == pos) e1 = &P[++ix1];                    /* M:6669-6671 */
            } else if (e1->type == 6) {                              /* M:6674-6711 */
                if (RV) site_mat(m, pos, M);
                if (e2->type == 6) {
                    double tot = 0.0;
                    if (contribLength != 0.0) {
                        omo_getPartialVec(m, 6, contribLength, M, 0.0, e2->vec, 0, 0, tot3);
                        for (int j = 0; j < 4; j++) tot += e1->vec[j] * tot3[j];
                    } else {
                        for (int j = 0; j < 4; j++) tot += e1->vec[j] * e2->vec[j];
                    }
                    totalFactor *= tot;
                } else {
                    int i2 = (e2->type == 4) ? e1->x : e2->type;
                    if (e1->vec[i2] > 0.02) totalFactor *= e1->vec[i2];
                    else {
                        if (U && (isTipC || (e2->len > 2 && e2->flag))) {
                            if (SS) errorRate = m->errorRates[pos];
                            omo_getPartialVec(m, i2, contribLength, M, errorRate, NULL, 0, 1, tot3);
                        } else omo_getPartialVec(m, i2, contribLength, M, 0.0, NULL, 0, 0, tot3);
                        double tot = 0.0;
                        for (int j = 0; j < 4; j++) tot += e1->vec[j] * tot3[j];
                        totalFactor *= tot;
                    }
                }
                pos += 1;
                if (pos == lRef) break;
                e1 = &P[++ix1];
                if (e2->type != 4 || e2->x == pos) e2 = &C[++ix2];
            } else {                                                 /* entry1 non-ref nuc, M:6713-6770 */
                if (e2->type != e1->type) {
                    int flag1 = (U && e1->len > 2 && e1->flag);
                    if (RV) site_mat(m, pos, M);
                    int i1 = e1->type;
                    if (e2->type < 5) {
                        int i2 = (e2->type == 4) ? e1->x : e2->type;
                        int flag2 = (U && (isTipC || (e2->len > 2 && e2->flag)));
                        if (e1->len == 4 + U) {
                            if (U && SS) errorRate = m->errorRates[pos];
                            omo_getPartialVec(m, i2, contribLength, M, errorRate, NULL, 0, flag2, tot3);
                            omo_getPartialVec(m, i1, e1->d0, M, errorRate, NULL, 0, flag1, tot2);
                            double tot = 0.0;
                            for (int j = 0; j < 4; j++) tot += rf[j] * tot3[j] * tot2[j];
                            totalFactor *= tot / rf[i1];
                        } else {
                            if (flag1 || flag2) {
                                if (SS) errorRate = m->errorRates[pos];
                                totalFactor *= (pymin(0.25, M[i1 * 4 + i2] * contribLength)
                                                + (double)(flag1 + flag2) * 0.33333 * errorRate);
                            } else {
                                if (contribLength != 0.0)
                                    totalFactor *= pymin(0.25, M[i1 * 4 + i2] * contribLength);
                                else { *outLK = -INFINITY; return 0; }
                            }
                        }
                    } else {                                         /* entry2 is O, M:6744-6761 */
                        if (U && SS) errorRate = m->errorRates[pos];
                        if (e2->vec[i1] > 0.02) totalFactor *= e2->vec[i1];
                        else {
                            if (e1->len == 4 + U) {
                                omo_getPartialVec(m, i1, e1->d0, M, errorRate, NULL, 0, flag1, tot2);
                                omo_getPartialVec(m, 6, contribLength, M, errorRate, e2->vec, 0, 0, tot3);
                                double tot = 0.0;
                                for (int i = 0; i < 4; i++) tot += tot2[i] * tot3[i] * rf[i];
                                totalFactor *= (tot / rf[i1]);
                            } else {
                                if (contribLength != 0.0) {
                                    omo_getPartialVec(m, 6, contribLength, M, 0.0, e2->vec, 0, 0, tot3);
                                    totalFactor *= tot3[i1];
                                } else totalFactor *= e2->vec[i1];
                            }
                        }
                    }
                }
                pos += 1;
                if (pos == lRef) break;
                e1 = &P[++ix1];
                if (e2->type != 4 || e2->x == pos) e2 = &C[++ix2];
            }
        }
        if (totalFactor <= m->minimumCarryOver) {                    /* M:6772-6783 */
            if (totalFactor < DBL_MIN) { *outLK = -INFINITY; return 0; }
            Lkcost += log(totalFactor);
            totalFactor = 1.0;
        }
    }
    /* the reference would raise on log(0) here; report -inf instead */
    *outLK = (totalFactor > 0.0) ? Lkcost + log(totalFactor) : -INFINITY;
    return 0;
}

/* ---- mergeVectors, M:4446-4859 ------------------------------------------------ */
int omo_mergeVectors(const OModel *m, const OEntry *pv1, int n1, double bLen1, int fromTip1,
                     const OEntry *pv2, int n2, double bLen2, int fromTip2, int returnLK, int isUpDown,
                     int numMinor1, int numMinor2, OEntry *out, double *outLK)
{
    const int U = U_(m), SS = m->errorRateSiteSpecific, RV = m->useRateVariation, lRef = m->lRef;
    const double *rf = m->rootFreqs, *cr = m->cumulativeRate, *cer = m->cumulativeErrorRate;
    int ix1 = 0, ix2 = 0, pos = 0, newPos = 0, no = 0;
    double totalFactor = 1.0, cumulPartLk = 0.0, cumErrorRate = 0.0, totSum;
    const OEntry *e1 = &pv1[0], *e2 = &pv2[0];
    double M[16], newVec[4], newVec2[4];
    double errorRate = m->errorRate;
    (void)n1; (void)n2;
    memcpy(M, m->Q, sizeof M);
    if (returnLK) {                                                  /* M:4486-4494 */
        cumulPartLk = (bLen1 + bLen2) * m->globalTotRate;
        if (U) {
            if (fromTip1 || numMinor1) cumulPartLk += m->totError * (1 + numMinor1);
            if (fromTip2 || numMinor2) cumulPartLk += m->totError * (1 + numMinor2);
        }
    }
    for (;;) {
        if (e1->type == 5) {
            if (e2->type == 5) {                                     /* M:4498-4500 */
                newPos = imin(e1->x, e2->x);
                out[no++] = mk(5, newPos, 2, 0, 0, 0);
            } else if (e2->type < 5) {                               /* M:4501-4548 */
                int newEl;
                if (e2->type < 4) { newPos = pos + 1; newEl = e2->x; }
                else { newPos = imin(e1->x, e2->x); newEl = newPos; }
                if (isUpDown) {
                    if (U) {
                        if (e2->len == 2) {
                            if (bLen2 != 0.0 || fromTip2) out[no++] = mk(e2->type, newEl, 5, bLen2, 0.0, fromTip2);
                            else out[no++] = mk(e2->type, newEl, 2, 0, 0, 0);
                        } else if (e2->len == 3) return -3;          /* unreachable in the reference (M:4515-4516) */
                        else out[no++] = mk(e2->type, newEl, 5, e2->d0 + bLen2, 0.0, e2->flag);
                    } else {
                        if (e2->len > 2) out[no++] = mk(e2->type, newEl, 4, e2->d0 + bLen2, 0.0, 0);
                        else if (bLen2 != 0.0) out[no++] = mk(e2->type, newEl, 4, bLen2, 0.0, 0);
                        else out[no++] = mk(e2->type, newEl, 2, 0, 0, 0);
                    }
                } else {
                    if (U) {
                        if (e2->len == 2) {
                            if (bLen2 != 0.0 || fromTip2) out[no++] = mk(e2->type, newEl, 4, bLen2, 0, fromTip2);
                            else out[no++] = mk(e2->type, newEl, 2, 0, 0, 0);
                        } else if (e2->len == 3) return -3;
                        else out[no++] = mk(e2->type, newEl, 4, e2->d0 + bLen2, 0, e2->flag);
                    } else {
                        if (e2->len > 2) out[no++] = mk(e2->type, newEl, 3, e2->d0 + bLen2, 0, 0);
                        else if (bLen2 != 0.0) out[no++] = mk(e2->type, newEl, 3, bLen2, 0, 0);
                        else out[no++] = mk(e2->type, newEl, 2, 0, 0, 0);
                    }
                }
            } else {                                                 /* N x O, M:4550-4576 */
                newPos = pos + 1;
                if (isUpDown) {
                    if (RV) site_mat(m, pos, M);
                    double totBLen = bLen2;
                    if (e2->len > 3) totBLen += e2->d0;
                    if (totBLen != 0.0) omo_getPartialVec(m, 6, totBLen, M, 0, e2->vec, 0, 0, newVec);
                    else memcpy(newVec, e2->vec, sizeof newVec);
                    for (int i = 0; i < 4; i++) newVec[i] *= rf[i];
                    totSum = 0.0; for (int i = 0; i < 4; i++) totSum += newVec[i];
                    for (int i = 0; i < 4; i++) newVec[i] /= totSum;
                    out[no++] = mkO(e2->x, 0, 0, newVec);
                } else {
                    if (e2->len > 3) out[no++] = mkO(e2->x, 1, e2->d0 + bLen2, e2->vec);
                    else if (bLen2 != 0.0) out[no++] = mkO(e2->x, 1, bLen2, e2->vec);
                    else out[no++] = mkO(e2->x, 0, 0, e2->vec);
                }
            }
            if (returnLK) {                                          /* M:4578-4587 */
                cumulPartLk += (bLen1 + bLen2) * (cr[pos] - cr[newPos]);
                if (U) {
                    if (fromTip1 || fromTip2) {
                        if (SS) cumErrorRate = cer[newPos] - cer[pos];
                        else cumErrorRate = errorRate * (newPos - pos);
                    }
                    if (fromTip1) cumulPartLk += cumErrorRate;
                    if (fromTip2) cumulPartLk += cumErrorRate;
                }
            }
            pos = newPos;
        } else if (e2->type == 5) {
            if (e1->type < 5) {                                      /* M:4590-4643 */
                int newEl;
                if (e1->type < 4) { newPos = pos + 1; newEl = e1->x; }
                else { newPos = imin(e1->x, e2->x); newEl = newPos; }
                if (isUpDown) {
                    if (U) {
                        if (e1->len == 2) {
                            if (bLen1 != 0.0) out[no++] = mk(e1->type, newEl, 4, bLen1, 0, 0);
                            else out[no++] = mk(e1->type, newEl, 2, 0, 0, 0);
                        } else if (e1->len == 3) return -3;
                        else if (e1->len == 4) out[no++] = mk(e1->type, newEl, 4, e1->d0 + bLen1, 0, e1->flag);
                        else out[no++] = mk(e1->type, newEl, 5, e1->d0, e1->d1 + bLen1, e1->flag);
                    } else {
                        if (e1->len == 2) {
                            if (bLen1 != 0.0) out[no++] = mk(e1->type, newEl, 3, bLen1, 0, 0);
                            else out[no++] = mk(e1->type, newEl, 2, 0, 0, 0);
                        } else if (e1->len == 3) out[no++] = mk(e1->type, newEl, 3, e1->d0 + bLen1, 0, 0);
                        else out[no++] = mk(e1->type, newEl, 4, e1->d0, e1->d1 + bLen1, 0);
                    }
                } else {
                    if (U) {
                        if (e1->len == 2) {
                            if (bLen1 != 0.0 || fromTip1) out[no++] = mk(e1->type, newEl, 4, bLen1, 0, fromTip1);
                            else out[no++] = mk(e1->type, newEl, 2, 0, 0, 0);
                        } else if (e1->len == 3) return -3;
                        else out[no++] = mk(e1->type, newEl, 4, e1->d0 + bLen1, 0, e1->flag);
                    } else {
                        if (e1->len > 2) out[no++] = mk(e1->type, newEl, 3, e1->d0 + bLen1, 0, 0);
                        else if (bLen1 != 0.0) out[no++] = mk(e1->type, newEl, 3, bLen1, 0, 0);
                        else out[no++] = mk(e1->type, newEl, 2, 0, 0, 0);
                    }
                }
            } else {                                                 /* O x N, M:4644-4668 */
                newPos = pos + 1;
                if (isUpDown && ((e1->len == 4 && e1->d0 > 0) || bLen1 != 0.0)) {
                    if (RV) site_mat(m, pos, M);
                    double totBLen = bLen1;
                    if (e1->len > 3) totBLen += e1->d0;
                    if (totBLen != 0.0) omo_getPartialVec(m, 6, totBLen, M, 0, e1->vec, 1, 0, newVec);
                    else memcpy(newVec, e1->vec, sizeof newVec);
                    totSum = 0.0; for (int i = 0; i < 4; i++) totSum += newVec[i];
                    for (int i = 0; i < 4; i++) newVec[i] /= totSum;
                    out[no++] = mkO(e1->x, 0, 0, newVec);
                } else {
                    if (e1->len > 3) out[no++] = mkO(e1->x, 1, e1->d0 + bLen1, e1->vec);
                    else if (bLen1 != 0.0) out[no++] = mkO(e1->x, 1, bLen1, e1->vec);
                    else out[no++] = mkO(e1->x, 0, 0, e1->vec);
                }
            }
            if (returnLK) {                                          /* M:4670-4679 */
                cumulPartLk += (bLen1 + bLen2) * (cr[pos] - cr[newPos]);
                if (U) {
                    if (fromTip1 || fromTip2) {
                        if (SS) cumErrorRate = cer[newPos] - cer[pos];
                        else cumErrorRate = errorRate * (newPos - pos);
                    }
                    if (fromTip1) cumulPartLk += cumErrorRate;
                    if (fromTip2) cumulPartLk += cumErrorRate;
                }
            }
            pos = newPos;
        } else {                                                     /* M:4682-4828 */
            double totLen1 = bLen1, totLen2 = bLen2;
            int refNucToPass;
            if (e1->type == 6) { if (e1->len > 3) totLen1 += e1->d0; }
            else if (e1->len > 2 + U) { totLen1 += e1->d0; if (e1->len > 3 + U) totLen1 += e1->d1; }
            if (e2->len > (2 + ((U || e2->type == 6) ? 1 : 0))) totLen2 += e2->d0;
            int flag1 = (U && e1->type != 6 && ((e1->len > 2 && e1->flag) || fromTip1));
            int flag2 = (U && e2->type != 6 && ((e2->len > 2 && e2->flag) || fromTip2));
            if (e1->type == 4 && e2->type == 4) newPos = imin(e1->x, e2->x);
            else newPos = pos + 1;
            if (returnLK) {                                          /* M:4703-4732 */
                if (e1->type == 4 && e2->type == 4) {
                    if (totLen2 > bLen2 || totLen1 > bLen1) {
                        cumulPartLk += (totLen2 - bLen2 + totLen1 - bLen1) * (cr[newPos] - cr[pos]);
                        if (U) {
                            if ((!fromTip1 && flag1) || (!fromTip2 && flag2)) {
                                if (SS) cumErrorRate = cer[pos] - cer[newPos];
                                else cumErrorRate = errorRate * (pos - newPos);
                                if (!fromTip1 && flag1) cumulPartLk += cumErrorRate;
                                if (!fromTip2 && flag2) cumulPartLk += cumErrorRate;
                            }
                        }
                    }
                } else {
                    refNucToPass = (e1->type != 4) ? e1->x : e2->x;
                    double dq = m->Q[refNucToPass * 4 + refNucToPass];
                    if (RV) dq = dq * m->siteRates[pos];
                    cumulPartLk -= dq * (bLen2 + bLen1);
                    if (U && ((e1->type != e2->type) || e1->type == 6) && (fromTip1 || fromTip2)) {
                        if (SS) cumErrorRate = m->errorRates[pos];
                        else cumErrorRate = errorRate;
                        if (fromTip1) cumulPartLk += cumErrorRate;
                        if (fromTip2) cumulPartLk += cumErrorRate;
                    }
                }
            }
            if (e2->type == e1->type && e2->type < 5) {              /* M:4734-4755 */
                if (e1->type == 4) out[no++] = mk(4, newPos, 2, 0, 0, 0);
                else {
                    out[no++] = mk(e1->type, e1->x, 2, 0, 0, 0);
                    if (returnLK) {
                        double dq = m->Q[e1->type * 4 + e1->type];
                        if (RV) dq = dq * m->siteRates[pos];
                        cumulPartLk += dq * (totLen1 + totLen2);
                        if (U) {
                            if ((!fromTip1 && flag1) || (!fromTip2 && flag2)) {
                                if (SS) cumErrorRate = m->errorRates[pos];
                                else cumErrorRate = errorRate;
                                if (!fromTip1 && flag1) cumulPartLk -= cumErrorRate;
                                if (!fromTip2 && flag2) cumulPartLk -= cumErrorRate;
                            }
                        }
                    }
                }
            } else if (totLen1 == 0.0 && totLen2 == 0.0 && e1->type < 5 && e2->type < 5 && !flag1 && !flag2) {
                if (returnLK) return -2;                             /* M:4757-4762 */
                return -1;
            } else {                                                 /* M:4763-4826 */
                int i1, i2;
                if (U && SS) errorRate = m->errorRates[pos];
                if (RV) site_mat(m, pos, M);
                if (e1->type == 4) { refNucToPass = e2->x; i1 = refNucToPass; }
                else { refNucToPass = e1->x; i1 = e1->type; }
                if (i1 <= 4) {
                    if (totLen1 != 0.0 || flag1) {
                        if (isUpDown && e1->len > 3 + U) {
                            omo_getPartialVec(m, i1, e1->d0, M, errorRate, NULL, 0, flag1, newVec);
                            for (int i = 0; i < 4; i++) newVec[i] *= rf[i];
                            if (e1->d1 + bLen1 != 0.0) {
                                double tmp[4];
                                omo_getPartialVec(m, 6, e1->d1 + bLen1, M, 0, newVec, 1, 0, tmp);
                                memcpy(newVec, tmp, sizeof tmp);
                            }
                        } else omo_getPartialVec(m, i1, totLen1, M, errorRate, NULL, isUpDown, flag1, newVec);
                    } else {
                        for (int i = 0; i < 4; i++) newVec[i] = 0.0;
                        newVec[i1] = 1.0;
                    }
                } else {
                    if (totLen1 != 0.0) omo_getPartialVec(m, 6, totLen1, M, 0, e1->vec, isUpDown, 0, newVec);
                    else memcpy(newVec, e1->vec, sizeof newVec);
                }
                i2 = (e2->type == 4) ? refNucToPass : e2->type;
                if (i2 == 6) {
                    if (totLen2 != 0.0) omo_getPartialVec(m, 6, totLen2, M, 0, e2->vec, 0, 0, newVec2);
                    else memcpy(newVec2, e2->vec, sizeof newVec2);
                } else {
                    if (totLen2 != 0.0 || flag2) omo_getPartialVec(m, i2, totLen2, M, errorRate, NULL, 0, flag2, newVec2);
                    else { for (int i = 0; i < 4; i++) newVec2[i] = 0.0; newVec2[i2] = 1.0; }
                }
                for (int j = 0; j < 4; j++) newVec[j] *= newVec2[j];
                totSum = 0.0; for (int i = 0; i < 4; i++) totSum += newVec[i];
                if (totSum == 0.0) { if (returnLK) return -2; return -1; }
                for (int i = 0; i < 4; i++) newVec[i] /= totSum;
                int state;
                if (omo_simplify(m, newVec, refNucToPass, &state) < 0) return -2;
                if (state == 6) out[no++] = mkO(refNucToPass, 0, 0, newVec);
                else if (state == 4) out[no++] = mk(4, newPos, 2, 0, 0, 0);
                else out[no++] = mk(state, refNucToPass, 2, 0, 0, 0);
                if (returnLK) totalFactor *= totSum;
            }
            pos = newPos;
        }
        if (returnLK && totalFactor <= m->minimumCarryOver) {        /* M:4830-4839 */
            if (totalFactor < DBL_MIN) return -2;
            cumulPartLk += log(totalFactor);
            totalFactor = 1.0;
        }
        if (pos == lRef) break;                                      /* M:4841-4854 */
        if (e1->type < 4 || e1->type == 6) e1 = &pv1[++ix1];
        else if (pos == e1->x) e1 = &pv1[++ix1];
        if (e2->type < 4 || e2->type == 6) e2 = &pv2[++ix2];
        else if (pos == e2->x) e2 = &pv2[++ix2];
    }
    if (returnLK && outLK) *outLK = cumulPartLk + log(totalFactor);
    return no;
}

/* ---- estimateBranchLengthWithDerivative, M:5040-5358 ---------------------------- */
int omo_estimateBranchLength(const OModel *m, const OEntry *P, int nP, const OEntry *C, int nC, int fromTipC,
                             double *tOut, int *isFalse, double *ais)
{
    const int U = U_(m), SS = m->errorRateSiteSpecific, RV = m->useRateVariation, lRef = m->lRef;
    const double *rf = m->rootFreqs, *cr = m->cumulativeRate;
    double c1 = m->globalTotRate;
    int nA = 0, nZeros = 0, ix1 = 0, ix2 = 0, pos = 0;
    const OEntry *e1 = &P[0], *e2 = &C[0];
    double M[16];
    double errorRate = m->errorRate;
    (void)nP; (void)nC;
    memcpy(M, m->Q, sizeof M);
    *isFalse = 0;
    for (;;) {
        if (e2->type == 5) {                                         /* M:5077-5083 */
            int end = (e1->type == 4 || e1->type == 5) ? imin(e1->x, e2->x) : pos + 1;
            c1 += (cr[pos] - cr[end]);
            pos = end;
        } else if (e1->type == 5) {                                  /* M:5084-5092 */
            int end = (e2->type == 4) ? imin(e1->x, e2->x) : pos + 1;
            c1 += (cr[pos] - cr[end]);
            pos = end;
        } else {
            if (e1->type == 4 && e2->type == 4) pos = imin(e1->x, e2->x);
            else {
                double coeff0 = 0.0, coeff1 = 0.0;
                if (RV) site_mat(m, pos, M);
                if (e1->type == 4) c1 -= M[e2->x * 4 + e2->x];
                else c1 -= M[e1->x * 4 + e1->x];
                int flag1 = (U && e1->type != 6 && e1->len > 2 && e1->flag);
                int flag2 = (U && e2->type != 6 && (fromTipC || (e2->len > 2 && e2->flag)));
                if (U && SS) errorRate = m->errorRates[pos];
                double contribLength = 0.0;                          /* M:5109-5124 (False) */
                if (e1->type < 5) {
                    if (e1->len == 3 + U) contribLength = e1->d0;
                    else if (e1->len == 4 + U) contribLength = e1->d1;
                } else if (e1->len > 3) contribLength = e1->d0;
                if (e2->type < 5) { if (e2->len > 2 + U) contribLength += e2->d0; }
                else if (e2->len > 3) contribLength += e2->d0;

                if (e1->type == 4) {
                    if (e2->type == 6) {                             /* M:5128-5155 */
                        int i1 = e2->x;
                        if (e1->len == 4 + U) {
                            coeff0 = rf[i1] * e2->vec[i1];
                            coeff1 = 0.0;
                            for (int i = 0; i < 4; i++) {
                                coeff0 += rf[i] * M[i * 4 + i1] * e1->d0 * e2->vec[i];
                                coeff1 += M[i1 * 4 + i] * e2->vec[i];
                            }
                            coeff1 *= rf[i1];
                            if (contribLength != 0.0) coeff0 += coeff1 * contribLength;
                            if (flag1) {
                                coeff0 -= 1.33333 * errorRate * rf[i1] * e2->vec[i1];
                                for (int i = 0; i < 4; i++) coeff0 += rf[i] * e2->vec[i] * 0.33333 * errorRate;
                            }
                        } else {
                            coeff0 = e2->vec[i1];
                            coeff1 = 0.0;
                            for (int j = 0; j < 4; j++) coeff1 += M[i1 * 4 + j] * e2->vec[j];
                            if (contribLength != 0.0) coeff0 += coeff1 * contribLength;
                        }
                        if (coeff1 < 0.0) c1 += coeff1 / coeff0;
                        else if (coeff1 != 0.0) { coeff0 = coeff0 / coeff1; ais[nA++] = coeff0; }
                        pos += 1;
                    } else {                                         /* M:5157-5185 */
                        int none = 0;
                        if (e1->len == 4 + U) {
                            int i1 = e2->x, i2 = e2->type;
                            coeff0 = rf[i2] * M[i2 * 4 + i1] * e1->d0;
                            if (contribLength != 0.0) coeff0 += rf[i1] * M[i1 * 4 + i2] * contribLength;
                            if (flag2) coeff0 += rf[i1] * 0.33333 * errorRate;
                            if (flag1) coeff0 += rf[i2] * 0.33333 * errorRate;
                            coeff1 = rf[i1] * M[i1 * 4 + i2];
                            if (coeff1 != 0.0) coeff0 = coeff0 / coeff1;
                            else none = 1;
                        } else {
                            coeff0 = contribLength;
                            if (flag2) {
                                double q = M[e2->x * 4 + e2->type];
                                if (q != 0.0) coeff0 += errorRate * 0.33333 / q;
                                else none = 1;
                            }
                        }
                        if (!none) { if (coeff0 != 0.0) ais[nA++] = coeff0; else nZeros++; }
                        pos += 1;
                    }
                } else if (e1->type == 6) {                          /* M:5188-5212 */
                    if (e2->type == 6) {
                        coeff0 = e1->vec[0] * e2->vec[0] + e1->vec[1] * e2->vec[1] + e1->vec[2] * e2->vec[2]
                                 + e1->vec[3] * e2->vec[3];
                        coeff1 = 0.0;
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 4; j++) coeff1 += e1->vec[i] * e2->vec[j] * M[i * 4 + j];
                        if (contribLength != 0.0) coeff0 += coeff1 * contribLength;
                    } else {
                        int i2 = (e2->type == 4) ? e1->x : e2->type;
                        coeff0 = e1->vec[i2];
                        coeff1 = 0.0;
                        for (int i = 0; i < 4; i++) coeff1 += e1->vec[i] * M[i * 4 + i2];
                        if (contribLength != 0.0) coeff0 += coeff1 * contribLength;
                        if (flag2) coeff0 += errorRate * 0.33333;
                    }
                    if (coeff1 < 0.0) c1 += coeff1 / coeff0;
                    else if (coeff1 != 0.0) { coeff0 = coeff0 / coeff1; ais[nA++] = coeff0; }
                    pos += 1;
                } else {                                             /* entry1 non-ref nuc, M:5216-5278 */
                    if (e2->type == e1->type) c1 += M[e1->type * 4 + e1->type];
                    else {
                        int i1 = e1->type;
                        if (e2->type < 5) {
                            int i2 = (e2->type == 4) ? e1->x : e2->type;
                            int none = 0;
                            if (e1->len == 4 + U) {
                                coeff0 = rf[i2] * M[i2 * 4 + i1] * e1->d0;
                                if (contribLength != 0.0) coeff0 += rf[i1] * M[i1 * 4 + i2] * contribLength;
                                if (flag2) coeff0 += rf[i1] * 0.33333 * errorRate;
                                if (flag1) coeff0 += rf[i2] * 0.33333 * errorRate;
                                coeff1 = rf[i1] * M[i1 * 4 + i2];
                                if (coeff1 != 0.0) coeff0 = coeff0 / coeff1;
                                else none = 1;
                            } else {
                                coeff0 = contribLength;
                                if (flag2) coeff0 += errorRate * 0.33333 / M[i1 * 4 + i2];
                            }
                            if (!none) { if (coeff0 != 0.0) ais[nA++] = coeff0; else nZeros++; }
                        } else {
                            if (e1->len == 4 + U) {
                                coeff0 = rf[i1] * e2->vec[i1];
                                coeff1 = 0.0;
                                for (int i = 0; i < 4; i++) {
                                    coeff0 += rf[i] * M[i * 4 + i1] * e1->d0 * e2->vec[i];
                                    coeff1 += M[i1 * 4 + i] * e2->vec[i];
                                }
                                coeff1 *= rf[i1];
                                if (contribLength != 0.0) coeff0 += coeff1 * contribLength;
                                if (flag1) {
                                    coeff0 -= 1.33333 * errorRate * rf[i1] * e2->vec[i1];
                                    for (int i = 0; i < 4; i++) coeff0 += rf[i] * e2->vec[i] * 0.33333 * errorRate;
                                }
                            } else {
                                coeff0 = e2->vec[i1];
                                coeff1 = 0.0;
                                for (int j = 0; j < 4; j++) coeff1 += M[i1 * 4 + j] * e2->vec[j];
                                if (contribLength != 0.0) coeff0 += coeff1 * contribLength;
                            }
                            if (coeff1 < 0.0) c1 += coeff1 / coeff0;
                            else if (coeff1 != 0.0) { coeff0 = coeff0 / coeff1; ais[nA++] = coeff0; }
                        }
                    }
                    pos += 1;
                }
            }
        }
        if (pos == lRef) break;                                      /* M:5281-5295 */
        if (e1->type < 4 || e1->type == 6) e1 = &P[++ix1];
        else if (pos == e1->x) e1 = &P[++ix1];
        if (e2->type < 4 || e2->type == 6) e2 = &C[++ix2];
        else if (pos == e2->x) e2 = &C[++ix2];
    }
    /* M:5298-5358 */
    const double sens = m->minBLenSensitivity;
    c1 = -c1;
    int n = nA + nZeros;
    double minAis, maxAis, tDown, tUp, vDown, vUp;
    if (n == 0) { *isFalse = 1; *tOut = 0.0; return 0; }
    if (nA) { minAis = ais[0]; for (int i = 1; i < nA; i++) if (ais[i] < minAis) minAis = ais[i]; }
    else minAis = 0.0;
    if (nZeros) minAis = pymin(0.0, minAis);
    if (minAis < 0.0) { *tOut = 0.1; return 0; }
    tDown = pymin(0.1, n / c1 - minAis);
    if (tDown <= 0.0) { *isFalse = 1; *tOut = 0.0; return 0; }
    vDown = nZeros ? nZeros / tDown : 0.0;
    for (int i = 0; i < nA; i++) vDown += 1.0 / (ais[i] + tDown);
    if (nA) { maxAis = ais[0]; for (int i = 1; i < nA; i++) if (ais[i] > maxAis) maxAis = ais[i]; }
    else maxAis = 0.0;
    tUp = pymin(0.1, n / c1 - maxAis);
    if (tUp >= 0.1) { *tOut = 0.1; return 0; }
    if (tUp <= sens) { if (minAis != 0.0) tUp = 0.0; else tUp = sens; }
    vUp = nZeros ? nZeros / tUp : 0.0;
    for (int i = 0; i < nA; i++) vUp += 1.0 / (ais[i] + tUp);
    if (vDown > c1 + sens || vUp < c1 - sens) {
        if (vUp < c1 - sens && tUp == 0.0) { *isFalse = 1; *tOut = 0.0; return 0; }
        if (vDown > c1 + sens && tDown >= 0.1) { *tOut = 0.1; return 0; }
    }
    while (tDown - tUp > sens) {
        double tMiddle = (tUp + tDown) / 2;
        double vMiddle = nZeros ? nZeros / tMiddle : 0.0;
        for (int i = 0; i < nA; i++) vMiddle += 1.0 / (ais[i] + tMiddle);
        if (vMiddle > c1) tUp = tMiddle; else tDown = tMiddle;
    }
    *tOut = tUp;
    return 0;
}

/* ---- areVectorsDifferent, M:5419-5472 --------------------------------------------- */
int omo_areVectorsDifferent(const OModel *m, const OEntry *pv1, int n1, const OEntry *pv2, int n2)
{
    const int U = U_(m), lRef = m->lRef;
    int ix1 = 0, ix2 = 0, pos = 0;
    if (pv2 == NULL || n2 == 0) return 1;
    const OEntry *e1 = &pv1[0], *e2 = &pv2[0];
    (void)n1;
    for (;;) {
        if (e1->type != e2->type) return 1;
        if (e1->len != e2->len) return 1;
        if (e1->type < 5) {
            if (e1->len > 2) {
                if (fabs(e1->d0 - e2->d0) > m->thresholdProb) return 1;
                if (e1->len > 3) {
                    /* element [3] is d1 (no error model, or len 5) or the flag (error model, len 4) */
                    double a3 = (U && e1->len == 4) ? (double)e1->flag : e1->d1;
                    double b3 = (U && e2->len == 4) ? (double)e2->flag : e2->d1;
                    if (fabs(a3 - b3) > m->thresholdProb) return 1;
                    if (e1->len > 4) {
                        if (fabs((double)e1->flag - (double)e2->flag) > m->thresholdProb) return 1;
                    }
                }
            }
            if (e1->type < 4) pos += 1;
            else pos = imin(e1->x, e2->x);
        } else if (e1->type == 6) {
            if (e1->len == 4) { if (fabs(e1->d0 - e2->d0) > m->thresholdProb) return 1; }
            for (int i = 0; i < 4; i++) {
                double diffVal = fabs(e1->vec[i] - e2->vec[i]);
                if (diffVal != 0.0) {
                    if (e1->vec[i] == 0.0 || e2->vec[i] == 0.0) return 1;
                    if (diffVal > m->thresholdDiffForUpdate
                        || (diffVal > m->thresholdProb
                            && ((diffVal / e1->vec[i] > m->thresholdFoldChangeUpdate)
                                || (diffVal / e2->vec[i] > m->thresholdFoldChangeUpdate))))
                        return 1;
                }
            }
            pos += 1;
        } else pos = imin(e1->x, e2->x);
        if (pos == lRef) break;
        if (e1->type < 4 || e1->type == 6) e1 = &pv1[++ix1];
        else if (pos == e1->x) e1 = &pv1[++ix1];
        if (e2->type < 4 || e2->type == 6) e2 = &pv2[++ix2];
        else if (pos == e2->x) e2 = &pv2[++ix2];
    }
    return 0;
}

/* ---- rootVector, M:4916-4996 ------------------------------------------------------- */
int omo_rootVector(const OModel *m, const OEntry *pv, int n, double bLen, int isFromTip, const int *mut3,
                   const int *pathOff, int nPath, OEntry *out, OEntry *tmp, int cap)
{
    const int U = U_(m);
    const double *rf = m->rootFreqs;
    double M[16];
    OEntry *a = out, *b = tmp;    /* ping-pong buffers */
    (void)cap;
    memcpy(a, pv, (size_t)n * sizeof(OEntry));
    for (int k = 0; k < nPath; k++) {                                /* M:4930-4940: up to the root frame */
        int lm = pathOff[k + 1] - pathOff[k];
        if (lm) {
            n = omo_passGenomeListThroughBranch(m, a, n, mut3 + 3 * pathOff[k], lm, 1, b);
            OEntry *t = a; a = b; b = t;
        }
    }
    int no = 0, newPos = 0;
    for (int k = 0; k < n; k++) {                                    /* M:4941-4986 */
        const OEntry *e = &a[k];
        if (e->type == 5) { b[no++] = *e; newPos = e->x; }
        else if (e->type == 6) {
            double nv[4], totSum = 0.0;
            double totBLen = bLen;
            if (e->len > 3) totBLen += e->d0;
            if (totBLen != 0.0) {
                site_mat(m, newPos, M);
                omo_getPartialVec(m, 6, totBLen, M, 0, e->vec, 0, 0, nv);
                for (int i = 0; i < 4; i++) nv[i] *= rf[i];
            } else for (int i = 0; i < 4; i++) nv[i] = e->vec[i] * rf[i];
            for (int i = 0; i < 4; i++) totSum += nv[i];
            for (int i = 0; i < 4; i++) nv[i] /= totSum;
            b[no++] = mkO(e->x, 0, 0, nv);
            newPos += 1;
        } else {
            if (U) {
                int flag1 = ((e->len > 2) && e->flag) || isFromTip;
                if (e->len > 3) b[no++] = mk(e->type, e->x, 5, e->d0 + bLen, 0.0, flag1);
                else if (bLen != 0.0 || flag1) b[no++] = mk(e->type, e->x, 5, bLen, 0.0, flag1);
                else b[no++] = mk(e->type, e->x, 2, 0, 0, 0);
            } else {
                if (e->len == 3) b[no++] = mk(e->type, e->x, 4, e->d0 + bLen, 0.0, 0);
                else if (bLen != 0.0) b[no++] = mk(e->type, e->x, 4, bLen, 0.0, 0);
                else b[no++] = mk(e->type, e->x, 2, 0, 0, 0);
            }
            if (e->type < 4) newPos += 1; else newPos = e->x;
        }
    }
    n = no;
    { OEntry *t = a; a = b; b = t; }
    for (int k = nPath - 1; k >= 0; k--) {                           /* M:4988-4993: back down */
        int lm = pathOff[k + 1] - pathOff[k];
        if (lm) {
            n = omo_passGenomeListThroughBranch(m, a, n, mut3 + 3 * pathOff[k], lm, 0, b);
            OEntry *t = a; a = b; b = t;
        }
    }
    n = omo_shorten(m, a, n);
    if (a != out) memcpy(out, a, (size_t)n * sizeof(OEntry));
    return n;
}

/* ---- findProbRoot, M:4865-4912 ------------------------------------------------------ */
int omo_findProbRoot(const OModel *m, const OEntry *pv, int n, const int *mut3, const int *pathOff, int nPath,
                     const int *cb, const double *rflec, OEntry *a, OEntry *b, int cap, double *outLK)
{
    const int U = U_(m), SS = m->errorRateSiteSpecific;
    const double *rf = m->rootFreqs;
    double rfLog[4];
    (void)cap;
    for (int i = 0; i < 4; i++) rfLog[i] = log(rf[i]);
    memcpy(a, pv, (size_t)n * sizeof(OEntry));
    for (int k = 0; k < nPath; k++) {
        int lm = pathOff[k + 1] - pathOff[k];
        if (lm) {
            n = omo_passGenomeListThroughBranch(m, a, n, mut3 + 3 * pathOff[k], lm, 1, b);
            OEntry *t = a; a = b; b = t;
        }
    }
    double errorRate = m->errorRate, logLK = 0.0, logFactor = 1.0;
    int pos = 0;
    for (int k = 0; k < n; k++) {
        const OEntry *e = &a[k];
        if (U && e->type < 5 && e->len > 2 && e->flag) {
            if (e->type == 4) { logLK += rflec[e->x] - rflec[pos]; pos = e->x; }
            else {
                if (SS) errorRate = m->errorRates[pos];
                logFactor *= (rf[e->type] * (1.0 - 1.33333 * errorRate) + 0.33333 * errorRate);
                pos += 1;
            }
        } else {
            if (e->type == 4) {
                for (int i = 0; i < 4; i++) logLK += rfLog[i] * (cb[e->x * 4 + i] - cb[pos * 4 + i]);
                pos = e->x;
            } else if (e->type < 4) { logLK += rfLog[e->type]; pos += 1; }
            else if (e->type == 6) {
                double tot = 0.0;
                for (int i = 0; i < 4; i++) tot += rf[i] * e->vec[i];
                logFactor *= tot;
                pos += 1;
            } else pos = e->x;
        }
        if (logFactor <= m->minimumCarryOver) {
            if (logFactor < DBL_MIN) { *outLK = -INFINITY; return 0; }
            logLK += log(logFactor);
            logFactor = 1.0;
        }
    }
    logLK += log(logFactor);
    *outLK = logLK;
    return 0;
}

/* ---- evaluatePlacement, M:6790-6806 --------------------------------------------------- */
int omo_evaluatePlacement(const OModel *m, const OEntry *midTot, int nMid, const OEntry *down, int nDown,
                          const OEntry *up, int nUp, double distance, const OEntry *rem, int nRem,
                          int isRemovedTip, int fromTip1, double defaultBLen, double *out4, OEntry *tmp, int cap,
                          double *scratch)
{
    OEntry *midLower = tmp, *midTop = tmp + cap, *newMid = tmp + 2 * cap;
    double bestApp, bestTop, bestBottom, cost;
    int f, nL, nT, nN;
    omo_estimateBranchLength(m, midTot, nMid, rem, nRem, isRemovedTip, &bestApp, &f, scratch);
    nL = omo_mergeVectors(m, down, nDown, distance / 2, fromTip1, rem, nRem, bestApp, isRemovedTip, 0, 0, 0, 0,
                          midLower, NULL);
    if (nL < 0) return -4;  /* the reference would fail on None here */
    omo_estimateBranchLength(m, up, nUp, midLower, nL, 0, &bestTop, &f, scratch);
    nT = omo_mergeVectors(m, up, nUp, bestTop, 0, rem, nRem, bestApp, isRemovedTip, 0, 1, 0, 0, midTop, NULL);
    if (nT == -1) {
        bestTop = defaultBLen * 0.1;
        nT = omo_mergeVectors(m, up, nUp, bestTop, 0, rem, nRem, bestApp, isRemovedTip, 0, 1, 0, 0, midTop, NULL);
    }
    if (nT < 0) return -4;
    omo_estimateBranchLength(m, midTop, nT, down, nDown, fromTip1, &bestBottom, &f, scratch);
    nN = omo_mergeVectors(m, up, nUp, bestTop, 0, down, nDown, bestBottom, fromTip1, 0, 1, 0, 0, newMid, NULL);
    if (nN < 0) return -4;
    omo_appendProbNode(m, newMid, nN, rem, nRem, isRemovedTip, bestApp, &cost);
    out4[0] = cost; out4[1] = bestBottom; out4[2] = bestTop; out4[3] = bestApp;
    return 0;
}

/* ---- batch driver for timing the CPU baseline (bench.py cpu_baseline leg only) --------------------- */
/* threads <= 1: the scalar loop; > 1: the same loop dealt to that many OpenMP threads (the pairs are independent and
 * the model is read-only), which is how a CPU would run this workload on all of the host's cores */
int omo_appendProbNode_batch_mt(const OModel *m, const OEntry *all, const long long *off, int n, const int *pl,
                                const int *cl, const unsigned char *tip, const double *bl, double *out, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 2048)
    for (int i = 0; i < n; i++) {
        const OEntry *P = all + off[pl[i]], *C = all + off[cl[i]];
        omo_appendProbNode(m, P, (int)(off[pl[i] + 1] - off[pl[i]]), C, (int)(off[cl[i] + 1] - off[cl[i]]), tip[i],
                           bl[i], &out[i]);
    }
    return 0;
}

int omo_appendProbNode_batch(const OModel *m, const OEntry *all, const long long *off, int n, const int *pl,
                             const int *cl, const unsigned char *tip, const double *bl, double *out)
{
    for (int i = 0; i < n; i++) {
        const OEntry *P = all + off[pl[i]], *C = all + off[cl[i]];
        omo_appendProbNode(m, P, (int)(off[pl[i] + 1] - off[pl[i]]), C, (int)(off[cl[i] + 1] - off[cl[i]]), tip[i],
                           bl[i], &out[i]);
    }
    return 0;
}

/* Lists in the HIP library's packed form (include/maple_hip.h: word = {pos, meta}, a per-list stream of doubles) as OEntry
 * tuples -- what oracle_py.packed_to_entries does with numpy, for whole trees of 10^8 entries (a conversion, no arithmetic):
 * meta = type | ref << 3 | hasD0 << 5 | hasD1 << 6 | flag << 7 | auxoff << 8; `len` = the Python tuple length (M:378-390). */
int omo_entries_from_packed(long long nLists, const long long *entOff, const int *pos, const unsigned *meta,
                            const long long *auxOff, const double *aux, int usingErrorRate, OEntry *out, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static, 4096)
    for (long long l = 0; l < nLists; l++) {
        const double *a = aux + auxOff[l];
        for (long long k = entOff[l]; k < entOff[l + 1]; k++) {
            const unsigned mt = meta[k];
            const int type = (int)(mt & 7u), ref = (int)((mt >> 3) & 3u);
            const int h0 = (int)((mt >> 5) & 1u), h1 = (int)((mt >> 6) & 1u), fl = (int)((mt >> 7) & 1u);
            const double *q = a + (mt >> 8);
            OEntry e;
            e.type = type;
            e.x = (type == 4 || type == 5) ? pos[k] : ref;
            e.flag = usingErrorRate ? fl : 0;
            e.d0 = h0 ? q[0] : 0.0;
            e.d1 = h1 ? q[h0] : 0.0;
            for (int j = 0; j < 4; j++) e.vec[j] = type == 6 ? q[h0 + h1 + j] : 0.0;
            if (type == 6) e.len = 3 + h0;
            else if (type == 5) e.len = 2;
            else e.len = 2 + h0 + h1 + ((h0 + h1) > 0 && usingErrorRate ? 1 : 0);
            out[k] = e;
        }
    }
    return 0;
}
