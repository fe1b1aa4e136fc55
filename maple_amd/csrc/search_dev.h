// maple_amd/csrc/search_dev.h -- device-resident SPR regraft search.
//
// One lane runs one findBestParentTopology (M:6817-7724) as a state machine: every loop iteration pops one
// item of the reference's LIFO stack `nodesToVisit`, so the order-dependent pruning (bestLKdiff, failedPasses)
// is reproduced exactly.  The "as-if-removed" genome lists live in a per-lane bump arena in HBM; list handles
// are indirect so that the reference's in-place shorten() (M:7087) is seen by every holder of the list.
// The wrapper around it is the worker body of startTopologyUpdatesParallel (M:9615-9711).
// Not implemented (flags that are off by default and outside BASELINE's configs): HnZ, time trees, SPRTA,
// --deeperSearchForLongBranches.
#pragma once
#include "genome_dev.h"

// MAPLE_SEARCH_OP decides whether the list operations of the search are real calls or inlined into the kernel.
// Measured (10 000-tip tree, deep round): inlined = 243 ms at 1 wave/SIMD, 2 min compile, 3 MB binary; calls = 257 ms
// at 2 waves/SIMD, 25 s compile.  The search is bound by the dependent-instruction latency of single lanes either way,
// so the smaller kernel with the higher occupancy is kept.
#ifndef MAPLE_SEARCH_OP
#define MAPLE_SEARCH_OP __noinline__
#endif

namespace maple {

struct ArenaViewS {                    // same fields as ArenaView in maple_hip.hip (kept POD for kernel args)
    const uint2 *words;
    const double *aux;
    const int64_t *ent_off;
    const int64_t *aux_off;
    const int32_t *n_ent;
    const int32_t *n_aux;
};

struct MutViewS {
    const int32_t *mut3;
    const int64_t *off;
    const int32_t *cnt;
};

// One 64-byte record per node: a search step touches a node and its relatives, and with one record per cache line
// that is one memory transaction per node instead of ten (the search is a chain of dependent loads).
struct alignas(64) NodeRec {
    int32_t up, c0, c1;                // -1 = none
    int32_t lower, upRight, upLeft, totUp;   // list ids, -1 = None
    int32_t mutId;                     // mutation-list id of the branch above the node, -1 = empty
    double dist;
    uint8_t isTip;                     // leaf without minor sequences
    uint8_t upIsRoot;                  // the parent is the root (saves the dependent nd[up].up load of M:7182 / 6984)
    uint8_t whichChild;                // 0 / 1: this node is child 0 / 1 of its parent
    uint8_t pad0;
    int32_t preRank;                   // rank in the search's own depth-first order (child 1's subtree first): the column of
                                       // this node in a cached score row, so that a descent reads the row sequentially
    int32_t frameOf;                   // index of the MAT reference frame the node's lists are expressed in (0 = the root's)
    int32_t c0Frame, c1Frame, upFrame; // the frames of its relatives: a frame change is visible without loading them
};

// The tree once more, in the order a search descends (NodeRec::preRank: the child pushed last, child 1, first): rank
// r + 1 is the first node visited after rank r and r + size skips r's clade, so the cached-regime descent into a clade is a
// forward scan over this array and over the search's score row (both read sequentially) instead of a chase through
// 64-byte node records and a stack in memory.
struct alignas(16) SScan {
    int32_t node;                      // node id
    int32_t size;                      // nodes in the clade rooted here (itself included)
    int32_t depth;                     // root = 0
    uint32_t ff;                       // frame << 4 | flags
};
enum { SS_SCORED = 1,                  // up != None and (dist > effectivelyNon0BLen or the parent is the root), M:6984
       SS_TOTUP = 2,                   // has a probVectTotUp
       SS_INNER = 4,                   // has children
       SS_ENTER = 8 };                 // the parent pushes it: its probVectUpRight / probVectUpLeft for this child exists (M:7105-7160)

struct DevTree {
    int32_t n, root;
    const NodeRec *nd;
    const int32_t *totUp;              // also kept as a flat column: the candidate list of the batch-scoring kernel
    const SScan *scan;                 // [n] by preRank, or null (then the cached regime pops one node at a time)
    const int32_t *scanParent;         // [n] by preRank: the rank of the parent
    int32_t scanDepthCap;              // per-depth slots a searching lane owns in LDS
    // for score rows that come with a bitmap of their finite scores (whole-tree searches, FiniteRows below):
    const int32_t *candBefore;         // [n + 1] by preRank: scored candidates (nodes with a probVectTotUp) of smaller rank
    const int32_t *cladeVisits;        // [n] by preRank: candidate placements counted below the node when its whole clade is
                                       // walked with every score -inf (the non-strict rule descends everywhere, M:7095)
};

// A search's row of the score table together with the bitmap of its finite scores (one bit per scored candidate, in rank
// order; one word per tile of 64 candidates of the dense kernel) and the number of finite scores before each word: only the
// finite scores are stored in the row, and a clade without a single finite score is counted, not walked.
struct FiniteRows {
    const unsigned long long *mask;    // [rows][nWords] or null (then every score of the row is stored)
    const int32_t *prefix;             // [rows][nWords + 1]
    int32_t nWords;
};
__device__ __forceinline__ bool fin_bit(const unsigned long long *m, int cb) { return (m[cb >> 6] >> (cb & 63)) & 1ull; }
__device__ __forceinline__ int fin_count_before(const unsigned long long *m, const int32_t *pf, int cb)
{
    const int w = cb >> 6, b = cb & 63;
    return pf[w] + (b ? __popcll(m[w] & ((1ull << b) - 1ull)) : 0);
}

struct SearchParams {
    int32_t strict;                    // strictTopologyStopRules
    int32_t allowedFails;              // allowedFailsTopology
    double thrLKtopology;              // thresholdLogLKtopology (already x log lRef)
    double thrPlacement;               // thresholdTopologyPlacement
    double thrOptTopo;                 // thresholdLogLKoptimizationTopology
    double thrConsec;                  // thresholdLogLKconsecutivePlacement
    double effNon0;                    // effectivelyNon0BLen
};

struct TList { const uint2 *w; const double *aux; int32_t n, na; };

struct StackItem {
    int32_t t1;
    int8_t dir, upd;
    int16_t fails;
    int32_t hPassed, hRpr;
    double distance, lastLK;
};

struct BestRec { int32_t t1, hUp, hDown, hMid, hRpr; double score, distance; };

struct SearchOut {                     // per query
    int32_t bestNode, placement, status, nAppend;
    double bestScore, improvement, currentLK;
    double blen[3];
    int64_t rprWoff, rprAoff;          // bestRemovedPartials inside the output pool (-1 = not stored)
    int32_t rprN, rprNA;
    int32_t nShortList, nSteps;        // profile: short-listed branches refined, state-machine steps (updating regime)
    int64_t tStep, tReplay, tRefine;   // profile (MAPLE_SPR_PROFILE builds): wall_clock64 ticks (10 ns) per phase
};

struct WsLayout {                      // per-lane workspace capacities
    int32_t capW, capA, capH, capS, capB, capAis;
};

struct LaneWs {
    uint2 *w; double *aux; TList *h; StackItem *st; BestRec *best; double *ais;
    int32_t usedW, usedA, nH, sp, nB;
    WsLayout L;
    int overflow;                      // 0, or which capacity ran out: 1 handles, 2 list words/aux, 3 coefficients, 4 stack, 5 short list, 6 other
    // When the lane's own room for lists is used up it carries on in a chunk of a pool the whole launch shares (lists are
    // referred to by pointer, so they stay where they are): the few searches that update lists for hundreds of steps no longer
    // come back for a second launch with a larger workspace.
    unsigned long long *ovfUsed = nullptr;
    uint2 *ovfW = nullptr;
    double *ovfA = nullptr;
    long long ovfChunks = 0;
    int chunkSerial = 0;
    __device__ inline int newHandle(const uint2 *w_, const double *a_, int n, int na)
    {
        if (nH >= L.capH) { overflow = 1; return -2; }
        h[nH] = TList{w_, a_, n, na};
        return nH++;
    }
    __device__ inline bool reserve(int nw)
    {
        if (usedW + nw > L.capW || usedA + 5 * nw > L.capA) {
            if (!ovfW || nw > L.capW || 5 * nw > L.capA) { overflow = 2; return false; }
            const unsigned long long k = atomicAdd(ovfUsed, 1ull);
            if ((long long)k >= ovfChunks) { overflow = 2; return false; }
            w = ovfW + k * (size_t)L.capW; aux = ovfA + k * (size_t)L.capA;
            usedW = usedA = 0;
            chunkSerial++;
        }
        return true;
    }
    __device__ inline int commit(const Writer &wr)
    {
        int hid = newHandle(w + usedW, aux + usedA, wr.n, wr.na);
        usedW += wr.n; usedA += wr.na;
        return hid;
    }
};

// LEAN: the kernel this search runs in has the wave-assisted path (lean visits, wave_append) compiled in
template <bool RV, bool U, bool SS, bool LEAN = false> struct Search {
    typedef Ctx<RV, U, SS> CT;
    const CT &c;
    const ArenaViewS &av;
    const MutViewS &mv;
    const DevTree &T;
    const SearchParams &P;
    LaneWs &ws;
    int nAppend;
    // Cached regime: when the search has stopped updating partials (needsUpdating False) the score of a branch is
    // appendProbNode(probVectTotUp[t1], removed list, ...) -- a pure function of (query, branch) that a batch kernel
    // computes ~50x faster per placement.  `cached` (if set) is this query's row of such scores, indexed by node.
    const double *cached = nullptr;
    const unsigned long long *finMask = nullptr;   // this row's bitmap of finite scores (only those are stored), or null
    __device__ __forceinline__ double cachedAt(int rank) const
    {
        if (finMask && !fin_bit(finMask, T.candBefore[rank])) return -INFINITY;
        return cached[rank];
    }
    // with MAT local references: this query's removed list in every reference frame (arena list ids, one per frame),
    // prepared by the host along the same up-then-down paths the traversal takes
    const int32_t *rTable = nullptr;
    int fShort[4] = {-1, -1, -1, -1};  // frames whose frame-table list the reference would have shortened in place (M:7087)
    // hand-over of a cached-regime descent to the whole wavefront (wave_scan_clade)
    StackItem scanItem;
    int scanRank = 0, scanSeedFrame = 0;
    bool scanFirstScored = false, wantScan = false;
    int budget = 0;                    // > 0: give up (status -5) after this many traversal placements without a cache
    bool overBudget = false;
    // lean: no score table, but the items that arrive with needsUpdating == False still go through the register-resident
    // loop of replayCached().  The score of a branch is then ASKED FOR: the loop puts the item back, leaves (wantApp), the
    // whole wavefront computes appendProbNode(probVectTotUp[appT1], removed list) (wave_append, wave_dev.h) and hands it over
    // in the mailbox; the loop is entered again and carries on with that item.  (Trees without MAT local references: the
    // removed list is one object for the whole search.)
    bool lean = false, wantApp = false, haveMail = false;
    bool stepOnce = false;             // the item on top crosses into another MAT reference frame: step() takes it (it passes the
                                       // removed list through the branch, M:7119 / 7342 / 7392), the lean loop carries on after it
    int appT1 = -1, appHRpr = -1, mailNode = -1;
    double mailScore = 0.0;
    // optional visit trace of ONE query (debugging / parity of the visit sequence)
    int32_t *trI = nullptr; double *trD = nullptr; int trN = 0, trCap = 0;
    __device__ inline void trace(int t1, int dir, int upd, int fails, double lastLK, double midProb)
    {
        if (!trI || trN >= trCap) return;
        trI[4 * trN] = t1; trI[4 * trN + 1] = dir; trI[4 * trN + 2] = upd; trI[4 * trN + 3] = fails;
        trD[2 * trN] = lastLK; trD[2 * trN + 1] = midProb;
        trN++;
    }

    __device__ Search(const CT &c_, const ArenaViewS &av_, const MutViewS &mv_, const DevTree &T_, const SearchParams &P_, LaneWs &ws_)
        : c(c_), av(av_), mv(mv_), T(T_), P(P_), ws(ws_), nAppend(0) {}

    // ---- list helpers (every op returns a handle; -1 = the reference's None; -2 = out of workspace) ----
    // handles >= 0 index the lane's handle table; handles <= -10 name a list of the tree arena directly
    // (read-only, never shortened in place)
    __device__ inline int treeList(int listId) const { return listId < 0 ? -1 : -(listId + 10); }
    __device__ inline TList L(int hid) const
    {
        if (hid >= 0) return ws.h[hid];
        int id = -hid - 10;
        return TList{av.words + av.ent_off[id], av.aux + av.aux_off[id], av.n_ent[id], av.n_aux[id]};
    }
    __device__ inline ListRef ref(int hid) const { TList t = L(hid); return ListRef{t.w, t.aux}; }
    __device__ inline int len(int hid) const { return hid >= 0 ? ws.h[hid].n : av.n_ent[-hid - 10]; }
    __device__ inline bool valid(int hid) const { return hid >= 0 || hid <= -10; }
    // a real (repointable) handle for a tree list: needed for the list that shorten() may edit in place
    __device__ inline int ownHandle(int listId)
    {
        if (listId < 0) return -1;
        return ws.newHandle(av.words + av.ent_off[listId], av.aux + av.aux_off[listId], av.n_ent[listId], av.n_aux[listId]);
    }
    // (the test that nearly always says "nothing to do" stays inline: a real call costs a register spill to scratch)
    __device__ __forceinline__ int opPass(int hid, int mutId, bool dirUp)
    {
        if (!valid(hid) || mutId < 0) return hid;
        return opPassReal(hid, mutId, dirUp);
    }
    __device__ MAPLE_SEARCH_OP int opPassReal(int hid, int mutId, bool dirUp)
    {
        int cnt = mv.cnt[mutId];
        if (cnt == 0) return hid;
        if (!ws.reserve(len(hid) + 2 * cnt)) return -2;
        Writer wr;
        wr.init(ws.w + ws.usedW, ws.aux + ws.usedA);
        pass_walk(c.m.lRef, ref(hid), mv.mut3 + 3 * mv.off[mutId], cnt, dirUp, wr);
        return ws.commit(wr);
    }
    __device__ MAPLE_SEARCH_OP int opMerge(int h1, double b1, bool t1, int h2, double b2, bool t2, bool upDown)
    {
        if (!valid(h1) || !valid(h2)) return -2;
        if (!ws.reserve(len(h1) + len(h2))) return -2;
        Writer wr;
        wr.init(ws.w + ws.usedW, ws.aux + ws.usedA);
        int r = merge_walk(c, ref(h1), b1, t1, ref(h2), b2, t2, upDown, false, 0, 0, wr, nullptr);
        if (r == -1) return -1;
        if (r < 0) return -2;
        return ws.commit(wr);
    }
    __device__ MAPLE_SEARCH_OP void opShortenInPlace(int hid)
    {
        if (hid < 0) return;                                          // only real handles can be repointed
        int n = ws.h[hid].n;
        if (!ws.reserve(n)) return;
        Writer wr;
        wr.init(ws.w + ws.usedW, ws.aux + ws.usedA);
        shorten_walk(c, ref(hid), n, wr);
        if (wr.n == n) return;                                        // nothing merged: keep the old storage
        ws.h[hid] = TList{ws.w + ws.usedW, ws.aux + ws.usedA, wr.n, wr.na};
        ws.usedW += wr.n; ws.usedA += wr.na;
    }
    // shorten() of a list this lane does not own (a tree / frame-table list): a shortened copy under a new handle
    __device__ MAPLE_SEARCH_OP int opShortenCopy(int hid)
    {
        const int n = len(hid);
        if (!ws.reserve(n)) return hid;
        Writer wr;
        wr.init(ws.w + ws.usedW, ws.aux + ws.usedA);
        shorten_walk(c, ref(hid), n, wr);
        if (wr.n == n) return hid;
        const int h = ws.newHandle(ws.w + ws.usedW, ws.aux + ws.usedA, wr.n, wr.na);
        if (h < 0) return hid;
        ws.usedW += wr.n; ws.usedA += wr.na;
        return h;
    }
    __device__ MAPLE_SEARCH_OP double opAppend(int hP, int hC, bool isTipC, double bLen)
    {
        nAppend++;
#ifdef MAPLE_SPR_PROFILE
        const long long t0 = wall_clock64();
        const ListRef rp = ref(hP), rc = ref(hC);
        const long long t1 = wall_clock64();
        const double v = append_walk(c, rp, rc, isTipC, bLen);
        tRefSetup += t1 - t0; tWalk += wall_clock64() - t1;
        return v;
#else
        return append_walk(c, ref(hP), ref(hC), isTipC, bLen);
#endif
    }
#ifdef MAPLE_SPR_PROFILE
    long long tWalk = 0, tRefSetup = 0;
#endif
    __device__ MAPLE_SEARCH_OP double opBlen(int hP, int hC, bool fromTipC)
    {
        bool f;
        if (len(hP) + len(hC) > ws.L.capAis) { ws.overflow = 3; return 0.0; }
        return blen_walk(c, ref(hP), ref(hC), fromTipC, ws.ais, 1, &f);
    }
    __device__ MAPLE_SEARCH_OP bool opDiffer(int h1, int h2)
    {
        if (!valid(h2)) return true;
        return differ_walk(c, ref(h1), ref(h2));
    }
    // rootVector(probVect, bLen, isFromTip, tree, node), M:4916-4996
    __device__ MAPLE_SEARCH_OP int opRootVector(int hid, double bLen, bool isFromTip, int node)
    {
        int cur = hid;
        if (!valid(cur)) return -2;
        for (int v = node; v >= 0; v = T.nd[v].up) { cur = opPass(cur, T.nd[v].mutId, true); if (!valid(cur)) return -2; }
        if (!ws.reserve(len(cur))) return -2;
        Writer wr;
        wr.init(ws.w + ws.usedW, ws.aux + ws.usedA);
        root_walk(c, ref(cur), bLen, isFromTip, wr);
        cur = ws.commit(wr);
        if (cur < 0) return -2;
        // back down: root first ... node last (M:4988-4993); walk the path again from the top
        int depth = 0;
        for (int v = node; v >= 0; v = T.nd[v].up) depth++;
        for (int k = depth - 1; k >= 0; k--) {
            int v = node;
            for (int s = 0; s < k; s++) v = T.nd[v].up;
            cur = opPass(cur, T.nd[v].mutId, false);
            if (cur < 0) return -2;                                  // always a real handle after root_walk
        }
        opShortenInPlace(cur);
        return cur;
    }

    __device__ inline void push(int t1, int dir, bool upd, int hPassed, double distance, double lastLK, int fails, int hRpr)
    {
        if (ws.sp >= ws.L.capS) { ws.overflow = 4; return; }
        StackItem &s = ws.st[ws.sp++];
        s.t1 = t1; s.dir = (int8_t)dir; s.upd = upd ? 1 : 0; s.fails = (int16_t)fails;
        s.hPassed = hPassed; s.hRpr = hRpr; s.distance = distance; s.lastLK = lastLK;
    }
    __device__ inline void record(int t1, double score, int hUp, int hDown, double distance, int hMid, int hRpr)
    {
        if (ws.nB >= ws.L.capB) { ws.overflow = 5; return; }
        ws.best[ws.nB++] = BestRec{t1, hUp, hDown, hMid, hRpr, score, distance};
    }
    __device__ inline int child(int v, int k) const { return k == 0 ? T.nd[v].c0 : T.nd[v].c1; }
    __device__ inline int upVectOf(int t1) const { return T.nd[t1].whichChild ? T.nd[T.nd[t1].up].upLeft : T.nd[T.nd[t1].up].upRight; }

    // search state
    int node, removed, sibling;
    bool isRemovedTip;
    double removedBLen, bestLKdiff, originalLK;
    int hBestRpr, bestNode;
    double bestScore, bl0, bl1, bl2;
    int refineIdx;

    // seeding of nodesToVisit, M:6855-6962
    __device__ MAPLE_SEARCH_OP void begin(int node_, int childIdx, double bestLK, double remBLen)
    {
        node = node_;
        removed = child(node, childIdx);
        sibling = child(node, 1 - childIdx);
        removedBLen = remBLen;
        bestLKdiff = originalLK = bestLK;
        bestNode = sibling;
        refineIdx = 0;
        int rpr = opPass(ownHandle(T.nd[removed].lower), T.nd[removed].mutId, true);
        hBestRpr = opPass(rpr, T.nd[sibling].mutId, false);
        isRemovedTip = T.nd[removed].isTip;
        if (T.nd[node].up >= 0) {
            int parent = T.nd[node].up;
            bool first = T.nd[parent].c0 == node;
            int vectUpUp = treeList(first ? T.nd[parent].upRight : T.nd[parent].upLeft);
            int pv1 = opPass(treeList(T.nd[sibling].lower), T.nd[sibling].mutId, true);
            int rpr1 = rpr;
            if (T.nd[node].mutId >= 0) { pv1 = opPass(pv1, T.nd[node].mutId, true); rpr1 = opPass(rpr, T.nd[node].mutId, true); }
            double d = T.nd[sibling].dist + T.nd[node].dist;
            push(parent, first ? 1 : 2, true, pv1, d, bestLK, 0, rpr1);
            vectUpUp = opPass(vectUpUp, T.nd[node].mutId, false);
            rpr1 = rpr;
            if (T.nd[sibling].mutId >= 0) { vectUpUp = opPass(vectUpUp, T.nd[sibling].mutId, false); rpr1 = opPass(rpr, T.nd[sibling].mutId, false); }
            push(sibling, 0, true, vectUpUp, d, bestLK, 0, rpr1);
            bl0 = T.nd[node].dist; bl1 = T.nd[sibling].dist; bl2 = remBLen;
        } else {
            if (T.nd[sibling].c0 >= 0) {                                // M:6916-6960, node is the root
                int ch1 = T.nd[sibling].c0, ch2 = T.nd[sibling].c1;
                int v1 = opPass(treeList(T.nd[ch2].lower), T.nd[ch2].mutId, true);
                v1 = opRootVector(v1, T.nd[ch2].dist, T.nd[ch2].isTip, node);
                int r1 = hBestRpr;
                if (T.nd[ch1].mutId >= 0) { r1 = opPass(hBestRpr, T.nd[ch1].mutId, false); v1 = opPass(v1, T.nd[ch1].mutId, false); }
                push(ch1, 0, true, v1, T.nd[ch1].dist, bestLK, 0, r1);
                int v2 = opPass(treeList(T.nd[ch1].lower), T.nd[ch1].mutId, true);
                v2 = opRootVector(v2, T.nd[ch1].dist, T.nd[ch1].isTip, node);
                int r2 = hBestRpr;
                if (T.nd[ch2].mutId >= 0) { r2 = opPass(hBestRpr, T.nd[ch2].mutId, false); v2 = opPass(v2, T.nd[ch2].mutId, false); }
                push(ch2, 0, true, v2, T.nd[ch2].dist, bestLK, 0, r2);
            }
            bl0 = 0.0; bl1 = T.nd[sibling].dist; bl2 = remBLen;
        }
        bestScore = originalLK;
    }

    // one iteration of "while nodesToVisit", M:6964-7434
    __device__ MAPLE_SEARCH_OP void step()
    {
        StackItem it = ws.st[--ws.sp];
        const int t1 = it.t1;
        bool upd = it.upd;
        int fails = it.fails;
        const int hPassed = it.hPassed, hRpr = it.hRpr;
        double distance = it.distance;
        double midProb;
        if (it.dir == 0) {                                           // moving from a parent to its child
            const int upT = T.nd[t1].up;
            if (!(upT == node || upT < 0) && (T.nd[t1].dist > P.effNon0 || T.nd[t1].upIsRoot)) {
                int midTot;
                if (upd) {
                    midTot = opMerge(hPassed, distance / 2, false, treeList(T.nd[t1].lower), distance / 2, T.nd[t1].isTip, true);
                    if (midTot < 0) return;
                    if (!opDiffer(midTot, treeList(T.nd[t1].totUp))) upd = false;
                } else {
                    midTot = treeList(T.nd[t1].totUp);
                    distance = T.nd[t1].dist;
                }
                if (!valid(midTot)) return;
                if (cached && !it.upd) { midProb = cachedAt(T.nd[t1].preRank); nAppend++; }   // only items that ARRIVED in the cached regime
                else midProb = opAppend(midTot, hRpr, isRemovedTip, removedBLen);
                if (budget > 0 && !cached && nAppend > budget) { overBudget = true; return; }
                if (midProb > bestLKdiff - P.thrOptTopo) {            // M:7071-7082
                    if (upd) record(t1, midProb, hPassed, treeList(T.nd[t1].lower), distance, midTot, hRpr);
                    else record(t1, midProb, -1, -1, 0.0, -1, hRpr);
                }
                if (midProb > bestLKdiff) { bestLKdiff = midProb; fails = 0; opShortenInPlace(hRpr); }
                else if (midProb < (it.lastLK - P.thrConsec)) fails++;
            } else midProb = it.lastLK;
            trace(t1, 0, upd, fails, it.lastLK, midProb);
            bool go;
            if (P.strict) go = fails <= P.allowedFails && midProb > (bestLKdiff - P.thrLKtopology) && T.nd[t1].c0 >= 0;
            else go = (fails <= P.allowedFails || midProb > (bestLKdiff - P.thrLKtopology)) && T.nd[t1].c0 >= 0;
            if (go) {
                for (int k = 0; k < 2; k++) {                         // child 0 uses vectUpRight, child 1 vectUpLeft
                    const int ch = child(t1, k), other = child(t1, 1 - k);
                    int vUp;
                    if (upd) {
                        int opv = opPass(treeList(T.nd[other].lower), T.nd[other].mutId, true);
                        vUp = opMerge(hPassed, distance, false, opv, T.nd[other].dist, T.nd[other].isTip, true);
                        if (vUp == -2) { if (!ws.overflow) ws.overflow = 6; return; }
                    } else vUp = treeList(k == 0 ? T.nd[t1].upRight : T.nd[t1].upLeft);
                    if (valid(vUp)) {
                        int r1 = opPass(hRpr, T.nd[ch].mutId, false);
                        if (upd) { vUp = opPass(vUp, T.nd[ch].mutId, false); push(ch, 0, true, vUp, T.nd[ch].dist, midProb, fails, r1); }
                        else push(ch, 0, false, -1, 0.0, midProb, fails, r1);
                    }
                }
            }
        } else {                                                     // crawling up from child (dir-1) to its parent t1
            const int other = child(t1, 2 - it.dir);
            const int upT = T.nd[t1].up;
            int midBottom = -1, vectUp = -1;
            if (upT >= 0 && (T.nd[t1].dist > P.effNon0 || T.nd[t1].upIsRoot)) {
                int midTot;
                if (upd) {
                    int opv = opPass(treeList(T.nd[other].lower), T.nd[other].mutId, true);
                    midBottom = opMerge(hPassed, distance, false, opv, T.nd[other].dist, T.nd[other].isTip, false);
                    if (midBottom < 0) return;
                    vectUp = opPass(treeList(upVectOf(t1)), T.nd[t1].mutId, false);
                    midTot = opMerge(vectUp, T.nd[t1].dist / 2, false, midBottom, T.nd[t1].dist / 2, false, true);
                    if (midTot < 0) return;
                    int cached = treeList(T.nd[t1].totUp);
                    if (!valid(cached))                               // "Node has no probVectTotUp ... calculating new one", M:7198-7200
                        cached = opMerge(vectUp, T.nd[t1].dist / 2, false, treeList(T.nd[t1].lower), T.nd[t1].dist / 2, false, true);
                    if (!opDiffer(midTot, cached)) upd = false;
                } else midTot = treeList(T.nd[t1].totUp);
                if (!valid(midTot)) return;
                if (cached && !it.upd) { midProb = cachedAt(T.nd[t1].preRank); nAppend++; }   // only items that ARRIVED in the cached regime
                else midProb = opAppend(midTot, hRpr, isRemovedTip, removedBLen);
                if (budget > 0 && !cached && nAppend > budget) { overBudget = true; return; }
                if (midProb >= (bestLKdiff - P.thrOptTopo)) {         // M:7293-7304 (>= here, > on the way down)
                    if (upd) record(t1, midProb, vectUp, midBottom, T.nd[t1].dist, midTot, hRpr);
                    else record(t1, midProb, -1, -1, 0.0, -1, hRpr);
                }
                if (midProb > bestLKdiff) { bestLKdiff = midProb; fails = 0; }
                else if (midProb < (it.lastLK - P.thrConsec)) fails++;
            } else midProb = it.lastLK;
            trace(t1, it.dir, upd, fails, it.lastLK, midProb);
            bool go;
            if (P.strict) go = fails <= P.allowedFails && midProb > (bestLKdiff - P.thrLKtopology);
            else go = fails <= P.allowedFails || midProb > (bestLKdiff - P.thrLKtopology);
            if (!go) return;
            if (upT >= 0) {
                const int upChild = T.nd[t1].whichChild;
                int vUp;
                if (upd) {
                    int vUpUp = opPass(treeList(upVectOf(t1)), T.nd[t1].mutId, false);
                    vUp = opMerge(vUpUp, T.nd[t1].dist, false, hPassed, distance, false, true);
                    if (vUp == -2) { if (!ws.overflow) ws.overflow = 6; return; }
                } else vUp = treeList(it.dir == 1 ? T.nd[t1].upLeft : T.nd[t1].upRight);
                if (!valid(vUp)) return;
                int r1 = opPass(hRpr, T.nd[other].mutId, false);
                if (upd) { vUp = opPass(vUp, T.nd[other].mutId, false); push(other, 0, true, vUp, T.nd[other].dist, midProb, fails, r1); }
                else push(other, 0, false, -1, 0.0, midProb, fails, r1);
                if (upd && midBottom < 0) {                           // M:7376-7384
                    int opv = opPass(treeList(T.nd[other].lower), T.nd[other].mutId, true);
                    midBottom = opMerge(hPassed, distance, false, opv, T.nd[other].dist, T.nd[other].isTip, false);
                    if (midBottom < 0) return;
                }
                r1 = opPass(hRpr, T.nd[t1].mutId, true);
                if (upd) { midBottom = opPass(midBottom, T.nd[t1].mutId, true); push(upT, upChild + 1, true, midBottom, T.nd[t1].dist, midProb, fails, r1); }
                else push(upT, upChild + 1, false, -1, 0.0, midProb, fails, r1);
            } else {                                                  // t1 is the root, M:7406-7432
                int r1 = opPass(hRpr, T.nd[other].mutId, false);
                if (upd) {
                    int vUp = opRootVector(hPassed, distance, false, t1);
                    vUp = opPass(vUp, T.nd[other].mutId, false);
                    if (!valid(vUp)) { if (!ws.overflow) ws.overflow = 6; return; }
                    push(other, 0, true, vUp, T.nd[other].dist, midProb, fails, r1);
                } else push(other, 0, false, -1, 0.0, midProb, fails, r1);
            }
        }
    }

    // The cached regime of step(), with the hot state in registers: items that
    // arrive with needsUpdating == False only compare cached scores and push their relatives, so a whole-tree search
    // is ~15 000 iterations of integer work.  Processes items until the stack is empty or its top item still needs
    // updating.  Semantics are those of step() (same order, same tie-breaks); the removed list is one shared object in
    // such trees, so the reference's in-place shorten() (M:7087) is done once, at the first improvement.
    __device__ __forceinline__ void replayCached()
    {
        if constexpr (!LEAN) replayCachedT<false>();                     // (no lean visits in this kernel: none of their code)
        else { if (cached) replayCachedT<false>(); else replayCachedT<true>(); }
    }
    // OWN: no score table -- a score is asked of the wavefront (lean lane searches); a template so that the replay over a
    // table does not carry the other form's code and registers
    template <bool OWN> __device__ __forceinline__ void replayCachedT()
    {
        int sp = ws.sp, nB = ws.nB, nApp = nAppend;
        double best = bestLKdiff;
        const double thrOpt = P.thrOptTopo, thrCons = P.thrConsec, thrLK = P.thrLKtopology, eff = P.effNon0;
        const int strict = P.strict, allowed = P.allowedFails, capS = ws.L.capS, capB = ws.L.capB;
        StackItem *st = ws.st;
        BestRec *br = ws.best;
        const NodeRec *nd = T.nd;
        const double *cs = cached;
        constexpr bool own = OWN;
        // the reference shortens the removed list in place at every improvement (M:7087); in the cached regime that has no
        // reader before the refinement, so the (few distinct) own handles are remembered and shortened on the way out
        int hShorten[4] = {-1, -1, -1, -1};
        const int32_t *rT = rTable;
        // the item pushed last is popped next: it stays in registers (`top`), so that the only load a visit depends on
        // is the node record itself
        StackItem top;
        bool haveTop = false;
        auto push = [&](int t1, int dir, int fails, int hRpr, double lastLK) {
            if (haveTop) st[sp++] = top;
            top.t1 = t1; top.dir = (int8_t)dir; top.upd = 0; top.fails = (int16_t)fails; top.hPassed = -1; top.hRpr = hRpr;
            top.distance = 0.0; top.lastLK = lastLK;
            haveTop = true;
        };
        for (;;) {
            StackItem it;
            if (haveTop) { it = top; haveTop = false; }
            else {
                if (sp <= 0 || st[sp - 1].upd) break;
                it = st[--sp];
            }
            const int t1 = it.t1;
            const NodeRec r1 = nd[t1];
            const int upT = r1.up;
            int fails = it.fails;
            double midProb = it.lastLK;
            const bool rootChild = r1.upIsRoot != 0;
            if (own && (r1.c0Frame != r1.frameOf || r1.c1Frame != r1.frameOf || r1.upFrame != r1.frameOf)) {
                top = it; haveTop = true;
                stepOnce = true;
                break;
            }
            if (it.dir == 0 && T.scan) {
                // The whole clade below this item goes to the wavefront (wave_scan_clade): this lane hands the item over
                // and applies the outcome; what the LIFO stack would do with the item and everything it pushes happens
                // there in the same order (a pushed clade is finished before anything older is popped).
                scanItem = it;
                scanRank = r1.preRank;
                scanSeedFrame = r1.frameOf;
                scanFirstScored = !(upT == node || upT < 0) && (r1.dist > eff || rootChild);
                wantScan = true;
                break;
            }
            if (it.dir == 0) {
                if (!(upT == node || upT < 0) && (r1.dist > eff || rootChild)) {
                    if (r1.totUp < 0) continue;
                    if (!own) midProb = cachedAt(r1.preRank);
                    else if (haveMail && mailNode == t1) { midProb = mailScore; haveMail = false; }
                    else {
                        // ask the wavefront for it -- after the shortenings the reference would have done by now (M:7087)
                        for (int k = 0; k < 4; k++) if (hShorten[k] >= 0) { opShortenInPlace(hShorten[k]); hShorten[k] = -1; }
                        top = it; haveTop = true;
                        wantApp = true; appT1 = t1; appHRpr = it.hRpr;
                        break;
                    }
                    nApp++;
                    if (own && budget > 0 && nApp > budget) { overBudget = true; break; }
                    if (midProb > best - thrOpt) {
                        if (nB >= capB) { ws.overflow = 5; break; }
                        br[nB++] = BestRec{t1, -1, -1, -1, it.hRpr, midProb, 0.0};
                    }
                    if (midProb > best) {
                        best = midProb; fails = 0;
                        if (it.hRpr >= 0 && hShorten[0] != it.hRpr && hShorten[1] != it.hRpr && hShorten[2] != it.hRpr && hShorten[3] != it.hRpr) {
                            hShorten[3] = hShorten[2]; hShorten[2] = hShorten[1]; hShorten[1] = hShorten[0]; hShorten[0] = it.hRpr;
                        } else if (it.hRpr <= -10 && rT) {
                            const int f = r1.frameOf;
                            if (fShort[0] != f && fShort[1] != f && fShort[2] != f && fShort[3] != f) {
                                fShort[3] = fShort[2]; fShort[2] = fShort[1]; fShort[1] = fShort[0]; fShort[0] = f;
                            }
                        }
                    } else if (midProb < (it.lastLK - thrCons)) fails++;
                }
                bool go;
                if (strict) go = fails <= allowed && midProb > (best - thrLK) && r1.c0 >= 0;
                else go = (fails <= allowed || midProb > (best - thrLK)) && r1.c0 >= 0;
                if (go) {
                    if (sp + 3 > capS) { ws.overflow = 4; break; }
                    if (r1.upRight >= 0) push(r1.c0, 0, fails, (rT && r1.c0Frame != r1.frameOf) ? treeList(rT[r1.c0Frame]) : it.hRpr, midProb);
                    if (r1.upLeft >= 0) push(r1.c1, 0, fails, (rT && r1.c1Frame != r1.frameOf) ? treeList(rT[r1.c1Frame]) : it.hRpr, midProb);
                }
            } else {
                const int other = (it.dir == 1) ? r1.c1 : r1.c0;
                if (upT >= 0 && (r1.dist > eff || rootChild)) {
                    if (r1.totUp < 0) continue;
                    if (!own) midProb = cachedAt(r1.preRank);
                    else if (haveMail && mailNode == t1) { midProb = mailScore; haveMail = false; }
                    else {
                        // ask the wavefront for it -- after the shortenings the reference would have done by now (M:7087)
                        for (int k = 0; k < 4; k++) if (hShorten[k] >= 0) { opShortenInPlace(hShorten[k]); hShorten[k] = -1; }
                        top = it; haveTop = true;
                        wantApp = true; appT1 = t1; appHRpr = it.hRpr;
                        break;
                    }
                    nApp++;
                    if (own && budget > 0 && nApp > budget) { overBudget = true; break; }
                    if (midProb >= (best - thrOpt)) {
                        if (nB >= capB) { ws.overflow = 5; break; }
                        br[nB++] = BestRec{t1, -1, -1, -1, it.hRpr, midProb, 0.0};
                    }
                    if (midProb > best) { best = midProb; fails = 0; }
                    else if (midProb < (it.lastLK - thrCons)) fails++;
                }
                bool go;
                if (strict) go = fails <= allowed && midProb > (best - thrLK);
                else go = fails <= allowed || midProb > (best - thrLK);
                if (!go) continue;
                if (sp + 3 > capS) { ws.overflow = 4; break; }
                if (upT >= 0) {
                    if (((it.dir == 1) ? r1.upLeft : r1.upRight) < 0) continue;
                    const int oFrame = (it.dir == 1) ? r1.c1Frame : r1.c0Frame;
                    push(other, 0, fails, (rT && oFrame != r1.frameOf) ? treeList(rT[oFrame]) : it.hRpr, midProb);
                    push(upT, (int)r1.whichChild + 1, fails, (rT && r1.upFrame != r1.frameOf) ? treeList(rT[r1.upFrame]) : it.hRpr, midProb);
                } else {
                    const int oFrame = (it.dir == 1) ? r1.c1Frame : r1.c0Frame;
                    push(other, 0, fails, (rT && oFrame != r1.frameOf) ? treeList(rT[oFrame]) : it.hRpr, midProb);
                }
            }
        }
        if (haveTop) st[sp++] = top;
        ws.sp = sp; ws.nB = nB; nAppend = nApp; bestLKdiff = best;
        for (int k = 0; k < 4; k++) if (hShorten[k] >= 0) opShortenInPlace(hShorten[k]);
    }

    // refinement of one short-listed branch, M:7460-7639 (evaluatePlacement M:6790-6806 inlined)
    __device__ MAPLE_SEARCH_OP int refine(const BestRec &r)
    {
        if (!(r.score >= originalLK - P.thrOptTopo)) return 0;
        const int t1 = r.t1;
        int upV, downV, midTot;
        double distance;
        if (r.hUp == -1) {
            upV = opPass(treeList(upVectOf(t1)), T.nd[t1].mutId, false);
            downV = treeList(T.nd[t1].lower);
            distance = T.nd[t1].dist;
            midTot = treeList(T.nd[t1].totUp);
        } else { upV = r.hUp; downV = r.hDown; distance = r.distance; midTot = r.hMid; }
        if (!valid(upV) || !valid(downV) || !valid(midTot)) return -1;
        const int rem = r.hRpr;
        const bool ft = T.nd[t1].isTip;
        const int saveW = ws.usedW, saveA = ws.usedA, saveH = ws.nH, saveSerial = ws.chunkSerial;
        double app = opBlen(midTot, rem, isRemovedTip);
        int midLower = opMerge(downV, distance / 2, ft, rem, app, isRemovedTip, false);
        if (midLower < 0) return -1;                                  // the reference fails here (caught by the worker)
        double top = opBlen(upV, midLower, false);
        int midTop = opMerge(upV, top, false, rem, app, isRemovedTip, true);
        if (midTop == -1) { top = c.m.defaultBLen * 0.1; midTop = opMerge(upV, top, false, rem, app, isRemovedTip, true); }
        if (midTop < 0) return -1;
        double bottom = opBlen(midTop, downV, ft);
        int newMid = opMerge(upV, top, false, downV, bottom, ft, true);
        if (newMid < 0) return -1;
        double cost = opAppend(newMid, rem, isRemovedTip, app);
        double initialCost = opAppend(upV, downV, ft, distance);
        double newPartialCost = opAppend(upV, downV, ft, bottom + top);
        double optimized = cost + newPartialCost - initialCost;
        // temporaries of this record are dead (after a move to a new chunk in between: everything in that chunk)
        if (ws.chunkSerial == saveSerial) { ws.usedW = saveW; ws.usedA = saveA; } else ws.usedW = ws.usedA = 0;
        ws.nH = saveH;
        if (optimized >= bestScore) {
            bestNode = t1; bestScore = optimized; bl0 = top; bl1 = bottom; bl2 = app; hBestRpr = rem;
        }
        return 0;
    }
};

// ---- cached-regime descent into one clade, by the whole wavefront ------------------------------------------------------
// All 64 lanes load 64 consecutive records of the tree in the search's depth-first order and the 64 scores of the
// search's row that go with them (two coalesced streams), then the reference's order-dependent rules -- running best,
// failedPasses, the short list, the strict / non-strict stop rule (M:7071-7103) -- are applied record by record from
// those registers (v_readlane with a wave-uniform index): no stack, no dependent global load per visit.  A record that is
// not descended into skips its clade (r += size).  (lastLK, failedPasses) handed from a node to its children live in
// one LDS slot per depth below the clade's root.  Every value that steers control flow is wave-uniform.
struct ScanState {                     // wave-uniform in / out
    double best;
    int nB, nApp, overflow;
    bool shortenSeed;                  // an improvement was found with the item's own removed list (shorten() it, M:7087)
    int fShort[4];
};

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readfirst_f64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ inline void wave_scan_clade(const SScan *__restrict__ SC, const int32_t *__restrict__ PR, const double *__restrict__ cs,
                                       const int32_t *__restrict__ rT, int r, bool firstScored, int seedFrame, int hSeed,
                                       double lastLK, int fails0, const SearchParams &P, BestRec *br, int capB, double *slotLK,
                                       int *slotFails, unsigned *slotOwner, int cap, ScanState &S,
                                       const unsigned long long *fm = nullptr, const int32_t *fp = nullptr, const int32_t *CB = nullptr,
                                       const int32_t *DV = nullptr)
{
    const int lane = threadIdx.x & 63;
    const double thrOpt = P.thrOptTopo, thrCons = P.thrConsec, thrLK = P.thrLKtopology;
    const int strict = P.strict, allowed = P.allowedFails;
    const SScan head = SC[r];
    const int end = r + head.size, d0 = head.depth;
    double best = S.best;
    int nB = S.nB, nApp = S.nApp;
    bool first = true;
    for (int k = lane; k < cap; k += 64) slotOwner[k] = 0u;
    unsigned serial = 1;                                                    // chunk number: slot ownership keys grow with it
    // the next chunk is nearly always the 64 records that follow: its three loads are issued before the current chunk is
    // resolved, so that the HBM latency of the score row (read once, never cached) overlaps with the rule evaluation
    int rAhead = -1;
    SScan recAhead = head;
    double scvAhead = 0.0;
    int prAhead = 0;
    while (r < end) {
        const int n = min(64, end - r);
        const bool valid = lane < n;
        const int idx = valid ? r + lane : end - 1;
        SScan rec;
        double scv;
        int prk;
        // (fm: only the finite scores of the row are stored; which ones they are is in the bitmap)
        auto scoreAt = [&](int at, const SScan &rc) -> double {
            if (!fm) return cs[at];
            if ((rc.ff & (SS_SCORED | SS_TOTUP)) != (SS_SCORED | SS_TOTUP)) return 0.0;   // (not a candidate: nothing is read)
            return fin_bit(fm, CB[at]) ? cs[at] : -INFINITY;
        };
        if (r == rAhead) { rec = recAhead; scv = scvAhead; prk = prAhead; }
        else { rec = SC[idx]; scv = scoreAt(idx, rec); prk = PR[idx]; }
        if (r + 64 < end) {
            rAhead = r + 64;
            const int idx2 = min(rAhead + lane, end - 1);
            recAhead = SC[idx2]; scvAhead = scoreAt(idx2, recAhead); prAhead = PR[idx2];
        }
        const int pl = prk - r;                                             // lane of the parent; < 0: an earlier chunk
        const int d = rec.depth, f = (int)(rec.ff >> 4);
        const uint32_t fl = rec.ff & 15u;
        const bool isFirst = first && lane == 0;
        const bool enter = isFirst || (fl & SS_ENTER);
        const bool scoredHere = isFirst ? firstScored : (fl & SS_SCORED) != 0;
        const bool dropped = scoredHere && !(fl & SS_TOTUP);                // visited, nothing scored, nothing pushed
        const bool counts = scoredHere && !dropped;                        // one appendProbNode evaluation when visited
        // ---- every record's (score it hands on, failedPasses, descended into?) under the CURRENT running best: the state
        // flows from parent to child.  Three prefix computations along the parent pointers inside the chunk, each by pointer jumping (log2 of the longest
        // chain rounds, one ds_bpermute per round): the score a record hands on (its own if it is scored, else that of the
        // nearest scored ancestor), failedPasses (a sum of per-record drops along the path) and "every ancestor was
        // descended into".  A record whose parent lies before the chunk starts from the parent's slot.
        const bool rootLane = !valid || isFirst || pl < 0;
        double inMp = 0.0;
        int inFails = 0;
        if (valid) {
            if (isFirst) { inMp = lastLK; inFails = fails0; }
            else if (pl < 0) { inMp = slotLK[d - d0 - 1]; inFails = slotFails[d - d0 - 1]; }
        }
        // (1) myMp: own score, or the nearest scored ancestor's (the parent's incoming score for a chain that leaves the chunk)
        double myMp = counts ? scv : inMp;
        {
            bool has = counts || rootLane;
            int ptr = rootLane ? lane : pl;
            while (__ballot(!has)) {
                const int src = has ? lane : ptr;
                const int oHasPtr = __shfl((has ? 64 : 0) | ptr, src, 64);   // the ancestor's (resolved?, its pointer)
                const double oMp = __shfl(myMp, src, 64);
                if (!has) {
                    if (oHasPtr & 64) { myMp = oMp; has = true; }
                    else ptr = oHasPtr & 63;
                }
            }
        }
        const double pMpIn = __shfl(myMp, rootLane ? lane : pl, 64);       // (every lane takes part: the source must be active)
        const double pMp = rootLane ? inMp : pMpIn;
        // (2) failedPasses = the parent's + (scored here and the score fell by more than thresholdLogLKconsecutivePlacement)
        int myFails = inFails + ((counts && myMp < (pMp - thrCons)) ? 1 : 0);
        {
            int ptr = rootLane ? -1 : pl;
            while (__ballot(ptr >= 0)) {
                const int o = __shfl((myFails << 8) | (ptr + 1), ptr >= 0 ? ptr : lane, 64);
                if (ptr >= 0) { myFails += o >> 8; ptr = (o & 255) - 1; }
            }
        }
        // (3) descended into: the record's own rule and every ancestor's
        const bool within = myMp > (best - thrLK);
        const bool rule = strict ? (myFails <= allowed && within) : (myFails <= allowed || within);
        bool ownGo = valid && enter && !dropped && (fl & SS_INNER) && rule;   // descended into, if it is visited at all
        // A record that is descended into with the score -inf and no finite score anywhere below it: every record of its clade
        // hands on (-inf, the same failedPasses), nothing is short-listed, the running best does not move, and the non-strict
        // rule keeps descending -- its clade is COUNTED (cladeVisits, static) instead of walked.
        bool skipClade = false;
        if (fm && !strict && ownGo && myMp == -INFINITY && rec.size > 1) {
            const int lo = CB[idx + 1], hi = CB[idx + rec.size];
            skipClade = fin_count_before(fm, fp, hi) == fin_count_before(fm, fp, lo);
            if (skipClade) ownGo = false;
        }
        bool go = ownGo;
        {
            int ptr = rootLane ? -1 : pl;
            while (__ballot(ptr >= 0)) {
                const int o = __shfl((go ? 256 : 0) | (ptr + 1), ptr >= 0 ? ptr : lane, 64);
                if (ptr >= 0) { go = go && (o & 256); ptr = (o & 255) - 1; }
            }
        }
        const int pGoIn = __shfl(go ? 1 : 0, rootLane ? lane : pl, 64);
        const bool pGo = rootLane || pGoIn != 0;
        const bool vis = valid && pGo && enter;
        // ---- a visited record that beats the running best changes the rules for everything after it: records before the
        // first such one are final; that one is applied on its own and the scan resumes behind it
        const unsigned long long imp = __ballot(valid && vis && counts && scv > best);
        const int nFinal = imp ? (int)__ffsll((long long)imp) - 1 : n;
        const bool fin = lane < nFinal;
        const unsigned long long cntM = __ballot(fin && vis && counts);
        nApp += __popcll(cntM);
        if (fm) {                                                           // the clades counted instead of walked
            int add = (fin && vis && skipClade) ? DV[idx] : 0;
            for (int off = 32; off > 0; off >>= 1) add += __shfl_xor(add, off, 64);
            nApp += add;
        }
        const bool rec1 = fin && vis && counts && myMp > (best - thrOpt);   // M:7071: the short list (on the way down: >)
        const unsigned long long recM = __ballot(rec1);
        const int nRec = __popcll(recM);
        if (nRec) {
            if (nB + nRec > capB) { S.overflow = 5; break; }
            if (rec1) {
                const int hr = (rT && f != seedFrame) ? -(rT[f] + 10) : hSeed;
                const int at = nB + __popcll(recM & ((1ull << lane) - 1ull));
                br[at] = BestRec{rec.node, -1, seedFrame, f, hr, myMp, 0.0};   // (hUp -1: a branch of the tree; the frames: frontier.hip)
            }
            nB += nRec;
        }
        // what a descended record hands to children in later chunks: the last such record of each depth owns the slot
        const bool keep = fin && go;
        if (__ballot(keep && (d - d0 >= cap))) { S.overflow = 4; break; }
        const unsigned key = serial * 64u + (unsigned)lane;
        if (keep) atomicMax(&slotOwner[d - d0], key);
        if (keep && slotOwner[d - d0] == key) { slotLK[d - d0] = myMp; slotFails[d - d0] = myFails; }
        serial++;
        int next;
        if (imp) {
            // the improving record, alone (its parent's state is final): M:7083-7091
            const int i = nFinal;
            const int di = __builtin_amdgcn_readlane(d, i), fi = __builtin_amdgcn_readlane(f, i);
            const int sizei = __builtin_amdgcn_readlane(rec.size, i);
            const uint32_t fli = (uint32_t)__builtin_amdgcn_readlane((int)fl, i);
            const double mp = readlane_f64(scv, i);
            const int hr = (rT && fi != seedFrame) ? -(rT[fi] + 10) : hSeed;
            nApp++;
            if (nB >= capB) { S.overflow = 5; break; }
            if (lane == 0) br[nB] = BestRec{__builtin_amdgcn_readlane(rec.node, i), -1, seedFrame, fi, hr, mp, 0.0};
            nB++;
            best = mp;
            if (hr >= 0) S.shortenSeed = true;
            else if (hr <= -10 && rT) {
                if (S.fShort[0] != fi && S.fShort[1] != fi && S.fShort[2] != fi && S.fShort[3] != fi) {
                    S.fShort[3] = S.fShort[2]; S.fShort[2] = S.fShort[1]; S.fShort[1] = S.fShort[0]; S.fShort[0] = fi;
                }
            }
            const bool goi = (fli & SS_INNER) != 0;                         // failedPasses = 0 and the score IS the best
            if (goi) {
                if (di - d0 >= cap) { S.overflow = 4; break; }
                if (lane == 0) { slotLK[di - d0] = mp; slotFails[di - d0] = 0; slotOwner[di - d0] = serial * 64u; }
                serial++;
                next = i + 1;
            } else next = i + sizei;
        } else {
            // skip the clade of the outermost record that is not descended into and reaches past this chunk
            const unsigned long long blk = __ballot(valid && !go && (lane + rec.size > n));
            next = blk ? ((int)__ffsll((long long)blk) - 1) : n;
            if (blk) next += __builtin_amdgcn_readlane(rec.size, next);
        }
        first = false;
        r += next;
    }
    S.best = best; S.nB = nB; S.nApp = nApp;
}

} // namespace maple
