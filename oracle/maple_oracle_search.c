/*
 * oracle/maple_oracle_search.c -- TEST INFRASTRUCTURE ONLY (see maple_oracle.h).
 *
 * CPU restatement of the SPR regraft search of MAPLE v0.7.5.4: findBestParentTopology (M:6817-7724) and the worker
 * body of startTopologyUpdatesParallel (M:9615-9711), written against the oracle's tuple-like lists and a plain
 * struct-of-arrays tree.  HnZ, time trees, SPRTA and --deeperSearchForLongBranches (all off by default) are not
 * restated.  Pinned to the reference's own records of these functions (tests/golden/search_*.json.gz,
 * tests/test_oracle_golden.py).  Frozen-tree semantics: the tree is never modified (the reference's in-place
 * shorten of the removed list acts on a private copy).
 */
#include "maple_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- a tiny arena of lists ------------------------------------------------------------------------------------- */
typedef struct { OEntry *e; int n; } OL;          /* a genome list; e == NULL means the reference's None */

typedef struct {
    char *mem; size_t cap, used;
    int failed;
} OArena;

static void *oa_alloc(OArena *a, size_t bytes)
{
    bytes = (bytes + 15) & ~(size_t)15;
    if (a->used + bytes > a->cap) {
        a->failed = 1;
        return NULL;
    }
    void *p = a->mem + a->used;
    a->used += bytes;
    return p;
}
static OL *ol_new(OArena *a, int cap)
{
    OL *l = (OL *)oa_alloc(a, sizeof(OL));
    if (!l) return NULL;
    l->e = (OEntry *)oa_alloc(a, (size_t)(cap > 0 ? cap : 1) * sizeof(OEntry));
    l->n = 0;
    return l->e ? l : NULL;
}

typedef struct {
    const OModel *m;
    const OTree *t;
    const OSearchParams *p;
    OArena *a;
    double *scratch;
    int nAppend;
} OS;

static OL *tree_list(OS *s, int kind, int node)
{
    const OTree *t = s->t;
    if (t->len[kind][node] <= 0) return NULL;
    OL *l = (OL *)oa_alloc(s->a, sizeof(OL));
    if (!l) return NULL;
    l->e = (OEntry *)(t->ent + t->start[kind][node]);   /* read-only view */
    l->n = t->len[kind][node];
    return l;
}
static const int *mut_of(const OTree *t, int node, int *cnt)
{
    *cnt = (int)(t->mutOff[node + 1] - t->mutOff[node]);
    return t->mut3 + 3 * t->mutOff[node];
}
static int is_tip(const OTree *t, int v) { return t->c0[v] < 0 && t->nMinor[v] == 0; }

/* passGenomeListThroughBranch(list, mutations[node], dirIsUp) -- returns the SAME object when there is nothing to do */
static OL *pass(OS *s, OL *l, int node, int up)
{
    int cnt;
    const int *mu = mut_of(s->t, node, &cnt);
    if (!l || cnt == 0) return l;
    OL *o = ol_new(s->a, l->n + 2 * cnt + 2);
    if (!o) return NULL;
    o->n = omo_passGenomeListThroughBranch(s->m, l->e, l->n, mu, cnt, up, o->e);
    return o;
}
static OL *merge(OS *s, OL *l1, double b1, int t1, OL *l2, double b2, int t2, int upDown)
{
    if (!l1 || !l2) return NULL;
    OL *o = ol_new(s->a, l1->n + l2->n + 2);
    if (!o) return NULL;
    int n = omo_mergeVectors(s->m, l1->e, l1->n, b1, t1, l2->e, l2->n, b2, t2, 0, upDown, 0, 0, o->e, NULL);
    if (n < 0) return NULL;
    o->n = n;
    return o;
}
static double append(OS *s, OL *P, OL *C, int tipC, double bLen)
{
    double lk;
    s->nAppend++;
    omo_appendProbNode(s->m, P->e, P->n, C->e, C->n, tipC, bLen, &lk);
    return lk;
}
static double blen(OS *s, OL *P, OL *C, int tipC)
{
    double t; int f;
    omo_estimateBranchLength(s->m, P->e, P->n, C->e, C->n, tipC, &t, &f, s->scratch);
    return t;    /* the reference's False is 0.0 wherever it is used as a number */
}
static int differ(OS *s, OL *a, OL *b)
{
    if (!b) return 1;
    return omo_areVectorsDifferent(s->m, a->e, a->n, b->e, b->n);
}
/* shorten(list) in place: the object is re-pointed to private storage first if it still views the tree */
static void shorten_inplace(OS *s, OL *l)
{
    OEntry *copy = (OEntry *)oa_alloc(s->a, (size_t)l->n * sizeof(OEntry));
    if (!copy) return;
    memcpy(copy, l->e, (size_t)l->n * sizeof(OEntry));
    l->e = copy;
    l->n = omo_shorten(s->m, l->e, l->n);
}
static OL *root_vector(OS *s, OL *l, double bLen, int fromTip, int node)
{
    const OTree *t = s->t;
    int depth = 0, total = 0;
    for (int v = node; v >= 0; v = t->up[v]) { int c; mut_of(t, v, &c); depth++; total += c; }
    int *off = (int *)oa_alloc(s->a, (size_t)(depth + 1) * sizeof(int));
    int *mu = (int *)oa_alloc(s->a, (size_t)(total > 0 ? total : 1) * 3 * sizeof(int));
    if (!off || !mu || !l) return NULL;
    int k = 0, w = 0;
    off[0] = 0;
    for (int v = node; v >= 0; v = t->up[v]) {
        int c; const int *src = mut_of(t, v, &c);
        memcpy(mu + 3 * w, src, (size_t)c * 3 * sizeof(int));
        w += c; off[++k] = w;
    }
    int cap = l->n + 4 * total + 4;
    OL *o = ol_new(s->a, cap);
    OEntry *tmp = (OEntry *)oa_alloc(s->a, (size_t)cap * sizeof(OEntry));
    if (!o || !tmp) return NULL;
    o->n = omo_rootVector(s->m, l->e, l->n, bLen, fromTip, mu, off, depth, o->e, tmp, cap);
    return o;
}

/* ---- findBestParentTopology ------------------------------------------------------------------------------------------ */
typedef struct { int t1, dir, upd, fails; OL *passed, *rpr; double distance, lastLK; } Item;
typedef struct { int t1; double score; OL *up, *down, *mid, *rpr; double distance; } Rec;

int omo_findBestParentTopology(const OModel *m, const OTree *t, const OSearchParams *p, int node, int child,
                               double bestLKdiff, double removedBLen, OSearchResult *res, void *arenaMem, size_t arenaBytes)
{
    OArena A = {(char *)arenaMem, arenaBytes, 0, 0};
    OS S = {m, t, p, &A, NULL, 0};
    S.scratch = (double *)oa_alloc(&A, 65536 * sizeof(double));
    const int capI = 8192, capR = 8192;
    Item *st = (Item *)oa_alloc(&A, (size_t)capI * sizeof(Item));
    Rec *rec = (Rec *)oa_alloc(&A, (size_t)capR * sizeof(Rec));
    if (!S.scratch || !st || !rec) return -3;
    int sp = 0, nR = 0;
    const int removed = child == 0 ? t->c0[node] : t->c1[node];
    const int sibling = child == 0 ? t->c1[node] : t->c0[node];
    int bestNode = sibling;
    /* M:6838-6850 */
    OL *rpr = tree_list(&S, 0, removed);
    if (rpr) {                                   /* private object: shorten() may edit it in place */
        OL *own = ol_new(&A, rpr->n);
        if (!own) return -3;
        memcpy(own->e, rpr->e, (size_t)rpr->n * sizeof(OEntry));
        own->n = rpr->n;
        rpr = own;
    }
    rpr = pass(&S, rpr, removed, 1);
    OL *bestRpr = pass(&S, rpr, sibling, 0);
    const int isRemovedTip = is_tip(t, removed);
    const double originalLK = bestLKdiff;
    double bl[3];
#define PUSH(T1, DIR, UPD, PASSED, DIST, LASTLK, FAILS, RPR)                                   \
    do { if (sp >= capI) return -3;                                                            \
         st[sp].t1 = (T1); st[sp].dir = (DIR); st[sp].upd = (UPD); st[sp].passed = (PASSED);   \
         st[sp].distance = (DIST); st[sp].lastLK = (LASTLK); st[sp].fails = (FAILS); st[sp].rpr = (RPR); sp++; } while (0)
    if (t->up[node] >= 0) {                                           /* M:6855-6914 */
        const int parent = t->up[node];
        const int first = t->c0[parent] == node;
        OL *vectUpUp = tree_list(&S, first ? 1 : 2, parent);
        OL *pv1 = pass(&S, tree_list(&S, 0, sibling), sibling, 1);
        OL *rpr1 = rpr;
        int cnt; mut_of(t, node, &cnt);
        if (cnt) { pv1 = pass(&S, pv1, node, 1); rpr1 = pass(&S, rpr, node, 1); }
        const double d = t->dist[sibling] + t->dist[node];
        PUSH(parent, first ? 1 : 2, 1, pv1, d, bestLKdiff, 0, rpr1);
        vectUpUp = pass(&S, vectUpUp, node, 0);
        rpr1 = rpr;
        mut_of(t, sibling, &cnt);
        if (cnt) { vectUpUp = pass(&S, vectUpUp, sibling, 0); rpr1 = pass(&S, rpr, sibling, 0); }
        PUSH(sibling, 0, 1, vectUpUp, d, bestLKdiff, 0, rpr1);
        bl[0] = t->dist[node]; bl[1] = t->dist[sibling]; bl[2] = removedBLen;
    } else {                                                          /* node is the root, M:6916-6960 */
        if (t->c0[sibling] >= 0) {
            const int ch1 = t->c0[sibling], ch2 = t->c1[sibling];
            OL *v1 = pass(&S, tree_list(&S, 0, ch2), ch2, 1);
            v1 = root_vector(&S, v1, t->dist[ch2], is_tip(t, ch2), node);
            OL *r1 = bestRpr;
            int cnt; mut_of(t, ch1, &cnt);
            if (cnt) { r1 = pass(&S, bestRpr, ch1, 0); v1 = pass(&S, v1, ch1, 0); }
            PUSH(ch1, 0, 1, v1, t->dist[ch1], bestLKdiff, 0, r1);
            OL *v2 = pass(&S, tree_list(&S, 0, ch1), ch1, 1);
            v2 = root_vector(&S, v2, t->dist[ch1], is_tip(t, ch1), node);
            OL *r2 = bestRpr;
            mut_of(t, ch2, &cnt);
            if (cnt) { r2 = pass(&S, bestRpr, ch2, 0); v2 = pass(&S, v2, ch2, 0); }
            PUSH(ch2, 0, 1, v2, t->dist[ch2], bestLKdiff, 0, r2);
        }
        bl[0] = 0.0; bl[1] = t->dist[sibling]; bl[2] = removedBLen;
    }
    while (sp > 0) {                                                  /* M:6964-7434 */
        if (A.failed) return -3;
        Item it = st[--sp];
        const int t1 = it.t1;
        int upd = it.upd, fails = it.fails;
        double distance = it.distance, midProb;
        if (it.dir == 0) {
            const int upT = t->up[t1];
            if (!(upT == node || upT < 0) && (t->dist[t1] > p->effNon0 || t->up[upT] < 0)) {
                OL *midTot;
                if (upd) {
                    midTot = merge(&S, it.passed, distance / 2, 0, tree_list(&S, 0, t1), distance / 2, is_tip(t, t1), 1);
                    if (!midTot) continue;
                    if (!differ(&S, midTot, tree_list(&S, 3, t1))) upd = 0;
                } else { midTot = tree_list(&S, 3, t1); distance = t->dist[t1]; }
                if (!midTot) continue;
                midProb = append(&S, midTot, it.rpr, isRemovedTip, removedBLen);
                if (midProb > bestLKdiff - p->thrOptTopo) {
                    if (nR >= capR) return -3;
                    if (upd) { Rec r = {t1, midProb, it.passed, tree_list(&S, 0, t1), midTot, it.rpr, distance}; rec[nR++] = r; }
                    else { Rec r = {t1, midProb, NULL, NULL, NULL, it.rpr, 0.0}; rec[nR++] = r; }
                }
                if (midProb > bestLKdiff) { bestLKdiff = midProb; fails = 0; shorten_inplace(&S, it.rpr); }
                else if (midProb < (it.lastLK - p->thrConsec)) fails++;
            } else midProb = it.lastLK;
            int go;
            if (p->strict) go = fails <= p->allowedFails && midProb > (bestLKdiff - p->thrLKtopology) && t->c0[t1] >= 0;
            else go = (fails <= p->allowedFails || midProb > (bestLKdiff - p->thrLKtopology)) && t->c0[t1] >= 0;
            if (go) {
                for (int k = 0; k < 2; k++) {
                    const int ch = k == 0 ? t->c0[t1] : t->c1[t1], other = k == 0 ? t->c1[t1] : t->c0[t1];
                    OL *vUp;
                    if (upd) {
                        OL *opv = pass(&S, tree_list(&S, 0, other), other, 1);
                        vUp = merge(&S, it.passed, distance, 0, opv, t->dist[other], is_tip(t, other), 1);
                    } else vUp = tree_list(&S, k == 0 ? 1 : 2, t1);
                    if (vUp) {
                        OL *r1 = pass(&S, it.rpr, ch, 0);
                        if (upd) { vUp = pass(&S, vUp, ch, 0); PUSH(ch, 0, 1, vUp, t->dist[ch], midProb, fails, r1); }
                        else PUSH(ch, 0, 0, NULL, 0.0, midProb, fails, r1);
                    }
                }
            }
        } else {
            const int other = it.dir == 1 ? t->c1[t1] : t->c0[t1];
            const int upT = t->up[t1];
            OL *midBottom = NULL, *vectUp = NULL;
            if (upT >= 0 && (t->dist[t1] > p->effNon0 || t->up[upT] < 0)) {
                OL *midTot;
                if (upd) {
                    OL *opv = pass(&S, tree_list(&S, 0, other), other, 1);
                    midBottom = merge(&S, it.passed, distance, 0, opv, t->dist[other], is_tip(t, other), 0);
                    if (!midBottom) continue;
                    vectUp = pass(&S, tree_list(&S, t->c0[upT] == t1 ? 1 : 2, upT), t1, 0);
                    midTot = merge(&S, vectUp, t->dist[t1] / 2, 0, midBottom, t->dist[t1] / 2, 0, 1);
                    OL *cached = tree_list(&S, 3, t1);
                    if (!cached)                                      /* M:7198-7200 */
                        cached = merge(&S, vectUp, t->dist[t1] / 2, 0, tree_list(&S, 0, t1), t->dist[t1] / 2, 0, 1);
                    if (!midTot) continue;
                    if (!differ(&S, midTot, cached)) upd = 0;
                } else midTot = tree_list(&S, 3, t1);
                if (!midTot) continue;
                midProb = append(&S, midTot, it.rpr, isRemovedTip, removedBLen);
                if (midProb >= (bestLKdiff - p->thrOptTopo)) {
                    if (nR >= capR) return -3;
                    if (upd) { Rec r = {t1, midProb, vectUp, midBottom, midTot, it.rpr, t->dist[t1]}; rec[nR++] = r; }
                    else { Rec r = {t1, midProb, NULL, NULL, NULL, it.rpr, 0.0}; rec[nR++] = r; }
                }
                if (midProb > bestLKdiff) { bestLKdiff = midProb; fails = 0; }
                else if (midProb < (it.lastLK - p->thrConsec)) fails++;
            } else midProb = it.lastLK;
            int go;
            if (p->strict) go = fails <= p->allowedFails && midProb > (bestLKdiff - p->thrLKtopology);
            else go = fails <= p->allowedFails || midProb > (bestLKdiff - p->thrLKtopology);
            if (!go) continue;
            if (upT >= 0) {
                const int upChild = t->c0[upT] == t1 ? 0 : 1;
                OL *vUp;
                if (upd) {
                    OL *vUpUp = pass(&S, tree_list(&S, upChild == 0 ? 1 : 2, upT), t1, 0);
                    vUp = merge(&S, vUpUp, t->dist[t1], 0, it.passed, distance, 0, 1);
                } else vUp = tree_list(&S, it.dir == 1 ? 2 : 1, t1);
                if (!vUp) continue;
                OL *r1 = pass(&S, it.rpr, other, 0);
                if (upd) { vUp = pass(&S, vUp, other, 0); PUSH(other, 0, 1, vUp, t->dist[other], midProb, fails, r1); }
                else PUSH(other, 0, 0, NULL, 0.0, midProb, fails, r1);
                if (upd && !midBottom) {
                    OL *opv = pass(&S, tree_list(&S, 0, other), other, 1);
                    midBottom = merge(&S, it.passed, distance, 0, opv, t->dist[other], is_tip(t, other), 0);
                    if (!midBottom) continue;
                }
                r1 = pass(&S, it.rpr, t1, 1);
                if (upd) { midBottom = pass(&S, midBottom, t1, 1); PUSH(upT, upChild + 1, 1, midBottom, t->dist[t1], midProb, fails, r1); }
                else PUSH(upT, upChild + 1, 0, NULL, 0.0, midProb, fails, r1);
            } else {
                OL *r1 = pass(&S, it.rpr, other, 0);
                if (upd) {
                    OL *vUp = root_vector(&S, it.passed, distance, 0, t1);
                    vUp = pass(&S, vUp, other, 0);
                    PUSH(other, 0, 1, vUp, t->dist[other], midProb, fails, r1);
                } else PUSH(other, 0, 0, NULL, 0.0, midProb, fails, r1);
            }
        }
    }
    /* refinement, M:7460-7639 */
    double bestScore = originalLK;
    for (int i = 0; i < nR; i++) {
        if (A.failed) return -3;
        Rec *r = &rec[i];
        if (!(r->score >= originalLK - p->thrOptTopo)) continue;
        const int t1 = r->t1;
        OL *upV, *downV, *midTot;
        double distance;
        if (!r->up) {
            const int upT = t->up[t1];
            upV = pass(&S, tree_list(&S, t->c0[upT] == t1 ? 1 : 2, upT), t1, 0);
            downV = tree_list(&S, 0, t1);
            distance = t->dist[t1];
            midTot = tree_list(&S, 3, t1);
        } else { upV = r->up; downV = r->down; distance = r->distance; midTot = r->mid; }
        if (!upV || !downV || !midTot) { res->nAppend = S.nAppend; return -1; }
        const int ft = is_tip(t, t1);
        const size_t save = A.used;
        const double app = blen(&S, midTot, r->rpr, isRemovedTip);
        OL *midLower = merge(&S, downV, distance / 2, ft, r->rpr, app, isRemovedTip, 0);
        if (!midLower) { res->nAppend = S.nAppend; return -1; }                                     /* the reference raises; its worker swallows it */
        double top = blen(&S, upV, midLower, 0);
        OL *midTop = merge(&S, upV, top, 0, r->rpr, app, isRemovedTip, 1);
        if (!midTop) { top = p->defaultBLen * 0.1; midTop = merge(&S, upV, top, 0, r->rpr, app, isRemovedTip, 1); }
        if (!midTop) { res->nAppend = S.nAppend; return -1; }
        const double bottom = blen(&S, midTop, downV, ft);
        OL *newMid = merge(&S, upV, top, 0, downV, bottom, ft, 1);
        if (!newMid) { res->nAppend = S.nAppend; return -1; }
        const double cost = append(&S, newMid, r->rpr, isRemovedTip, app);
        const double initialCost = append(&S, upV, downV, ft, distance);
        const double newPartialCost = append(&S, upV, downV, ft, bottom + top);
        const double optimized = cost + newPartialCost - initialCost;
        A.used = save;
        if (optimized >= bestScore) { bestNode = t1; bestScore = optimized; bl[0] = top; bl[1] = bottom; bl[2] = app; bestRpr = r->rpr; }
    }
    if (A.failed) return -3;
    res->bestNode = bestNode; res->bestScore = bestScore;
    res->blen[0] = bl[0]; res->blen[1] = bl[1]; res->blen[2] = bl[2];
    res->nAppend = S.nAppend;
    res->rprN = 0;
    if (bestRpr && res->rpr && bestRpr->n <= res->rprCap) {
        memcpy(res->rpr, bestRpr->e, (size_t)bestRpr->n * sizeof(OEntry));
        res->rprN = bestRpr->n;
    }
    return 0;
#undef PUSH
}

/* ---- worker body of startTopologyUpdatesParallel, M:9615-9711 ------------------------------------------------------ */
int omo_sprWorker(const OModel *m, const OTree *t, const OSearchParams *p, int n, const int *nodes, OSearchResult *out,
                  void *arenaMem, size_t arenaBytes)
{
    for (int i = 0; i < n; i++) {
        OSearchResult *r = &out[i];
        const int node = nodes[i];
        r->bestNode = -1; r->placement = -1; r->status = 0; r->nAppend = 0;
        r->bestScore = r->improvement = r->currentLK = 0.0;
        r->blen[0] = r->blen[1] = r->blen[2] = 0.0;
        const int parent = t->up[node];
        if (parent < 0) { r->status = 1; continue; }
        OArena A = {(char *)arenaMem, arenaBytes, 0, 0};
        OS S = {m, t, p, &A, NULL, 0};
        const int child = t->c0[parent] == node ? 0 : 1;
        OL *vectUp = pass(&S, tree_list(&S, child == 0 ? 1 : 2, parent), node, 0);
        OL *low = tree_list(&S, 0, node);
        if (!vectUp || !low) { r->status = -1; continue; }
        double cur;
        omo_appendProbNode(m, vectUp->e, vectUp->n, low->e, low->n, is_tip(t, node), t->dist[node], &cur);
        r->currentLK = cur;
        if (!(cur < p->thrPlacement || t->dist[node] != 0.0)) { r->status = 2; continue; }
        int rc = omo_findBestParentTopology(m, t, p, parent, child, cur, t->dist[node], r, arenaMem, arenaBytes);
        if (rc != 0) { r->status = rc; continue; }                    /* (-1: nAppend = the calls issued before the reference raises) */
        if (r->bestScore + p->thrPlacement > cur) {                   /* M:9681-9700 */
            int updated = 1;
            int topNode = t->up[node];
            if (r->bestNode == topNode) updated = 0;
            while (t->dist[topNode] == 0.0 && t->up[topNode] >= 0) topNode = t->up[topNode];
            if (r->bestNode == topNode && r->blen[1] == 0.0) updated = 0;
            const int sib = t->c0[parent] == node ? t->c1[parent] : t->c0[parent];
            if (r->bestNode == sib) updated = 0;
            if (t->up[r->bestNode] == sib && r->blen[0] == 0.0) updated = 0;
            if (updated) { r->improvement = r->bestScore - cur; r->placement = r->bestNode; }
        }
    }
    return 0;
}

/* The same worker dealt to `threads` OpenMP threads (searches are independent given the frozen tree -- the reference's own
 * Pool.map over numCores, M:12283-12293); every thread owns arenaBytesPerThread bytes of arenaMem.  threads <= 1: the
 * scalar loop above.  Results are those of omo_sprWorker whatever the thread count. */
int omo_sprWorker_mt(const OModel *m, const OTree *t, const OSearchParams *p, int n, const int *nodes, OSearchResult *out,
                     void *arenaMem, size_t arenaBytesPerThread, int threads)
{
    if (threads <= 1) return omo_sprWorker(m, t, p, n, nodes, out, arenaMem, arenaBytesPerThread);
#pragma omp parallel num_threads(threads)
    {
        int tid = 0;
#ifdef _OPENMP
        extern int omp_get_thread_num(void);
        tid = omp_get_thread_num();
#endif
        char *mine = (char *)arenaMem + (size_t)tid * arenaBytesPerThread;
#pragma omp for schedule(dynamic, 4)
        for (int i = 0; i < n; i++) omo_sprWorker(m, t, p, 1, nodes + i, out + i, mine, arenaBytesPerThread);
    }
    return 0;
}
