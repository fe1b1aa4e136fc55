// maple_amd/csrc/spr_batch.hip -- the tree mirror on the device (maple_tree_upload / maple_tree_patch) and the SPR search of a
// batch of pruned nodes (maple_spr_search_batch: findBestParentTopology, M:6817-7724, inside the worker body of
// startTopologyUpdatesParallel, M:9580-9716): the one-lane-per-search kernels (k_spr_search: state machine in search_dev.h), the
// dense / witness scoring of the whole-tree searches, the hand-over to and from the frontier tier (frontier.hip).
// gfx950 only.  Split off maple_hip.hip (round 4).
#include "../../include/maple_hip.h"
#include "genome_dev.h"
#include "search_dev.h"
#include "placement_dev.h"
#include "append_lds.h"
#include "wave_dev.h"
#include "wave_update.h"

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace maple;

#include <hipcub/hipcub.hpp>
#include "ctx_host.h"
#include "batch_host.h"
#include "frontier.h"
#include "witness.h"

// 248 VGPRs let only 2 wavefronts share a SIMD; the search is latency-bound, so capping it at 128 VGPRs (4 wavefronts,
// some state spilled to scratch) is faster: deep round 134 -> 103 ms (3 waves 115, 5 waves 144, 8 waves 186).
#ifndef MAPLE_SPR_ATTR
#define MAPLE_SPR_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
// SPR regraft search: one lane = one query (state machine in search_dev.h) ---------------------------
struct LaneBytes { size_t w, aux, h, st, best, ais, total; };
static LaneBytes lane_bytes(const WsLayout &L)
{
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    LaneBytes b;
    b.w = al((size_t)L.capW * sizeof(uint2));
    b.aux = al((size_t)L.capA * sizeof(double));
    b.h = al((size_t)L.capH * sizeof(TList));
    b.st = al((size_t)L.capS * sizeof(StackItem));
    b.best = al((size_t)L.capB * sizeof(BestRec));
    b.ais = al((size_t)L.capAis * sizeof(double));
    b.total = b.w + b.aux + b.h + b.st + b.best + b.ais;
    return b;
}

// ASSIST: with the wave-assisted lane searches compiled in.  A kernel of its own (k_spr_search_assisted): inlined next to the
// rest, that path costs every launch registers (with an error model the whole kernel spilled four times as many, and the
// replay launches, which never use it, slowed by a quarter).
template <bool RV, bool U, bool SS, bool ASSIST>
__device__ __forceinline__ void spr_search_impl(const DevModel *__restrict__ mp, ArenaView av, MutView mv, DevTree T, SearchParams P, int n,
                                                   const int32_t *nodes, WsLayout L, LaneBytes LB, uint8_t *wsBase,
                                                   int32_t *counter, SearchOut *out, uint2 *poolW, double *poolA,
                                                   unsigned long long *poolUsed, long long poolCapW, long long poolCapA,
                                                   int traceQuery, int32_t *trI, double *trD, int trCap, int32_t *trN,
                                                   int activeLanes, const double *cacheS, int budget, const int32_t *rTable, int nF,
                                                   const int32_t *cacheRow, int leanVisits, unsigned long long *ovfUsed, uint8_t *ovfBase,
                                                   long long ovfChunks, FiniteRows fin)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    // only the first `activeLanes` lanes of every wavefront search: with few queries it is better to spread them
    // over many wavefronts (a wavefront executes the union of its lanes' control paths) than to fill 64-wide waves
    // Cached (whole-tree) launches run ONE search per wavefront (activeLanes == 1): lane 0 is the state machine, and all 64
    // lanes join it whenever the search descends into a clade in the cached regime (wave_scan_clade, search_dev.h).
    const bool coop = T.scan != nullptr;
    // lane searches on a tree without local references: the lanes that do not search stay and help -- every cached-regime
    // placement score a searching lane needs is computed by all 64 (wave_append), one request after the other
    // (not compiled into the kernels of the error models, where it is not used: their site factors are large, and with the
    // wavefront-wide walk inlined next to them the whole kernel spilled four times as many registers)
    const bool assist = ASSIST && !coop && leanVisits != 0 && cacheS == nullptr;
    const int coopMax = leanVisits > 1 ? leanVisits - 1 : 0;                 // (leanVisits = 1 + the most requests served one by one)
    if (!coop && !assist && (int)threadIdx.x >= activeLanes) return;
    const bool searcher = (int)threadIdx.x < activeLanes;
    const size_t lane = (size_t)blockIdx.x * activeLanes + (searcher ? threadIdx.x : 0);
    LaneWs ws;
    // the lane's own workspace (a search that outgrows its list room carries on in chunks of a shared pool: LaneWs::reserve)
    auto setWs = [&](uint8_t *base, const LaneBytes &B, const WsLayout &Lx) {
        ws.w = (uint2 *)base;
        ws.aux = (double *)(base + B.w);
        ws.h = (TList *)(base + B.w + B.aux);
        ws.st = (StackItem *)(base + B.w + B.aux + B.h);
        ws.best = (BestRec *)(base + B.w + B.aux + B.h + B.st);
        ws.ais = (double *)(base + B.w + B.aux + B.h + B.st + B.best);
        ws.L = Lx;
    };
    setWs(wsBase + lane * LB.total, LB, L);
    if (ovfBase && ovfChunks > 0) {
        ws.ovfUsed = ovfUsed; ws.ovfChunks = ovfChunks;
        ws.ovfW = (uint2 *)ovfBase; ws.ovfA = (double *)(ovfBase + (size_t)ovfChunks * L.capW * sizeof(uint2));
    }
    Search<RV, U, SS, ASSIST> S(c, av, mv, T, P, ws);
    extern __shared__ double dynLds[];   // coop: per-depth (lastLK, failedPasses) slots of the clade scan; assisted lane searches:
    WaveLds &wl = *(WaveLds *)dynLds;    // the staging area of the wavefront-wide appendProbNode (wave_dev.h)
    double *slotLK = dynLds;
    int *slotFails = (int *)(dynLds + T.scanDepthCap);
    unsigned *slotOwner = (unsigned *)(slotFails + T.scanDepthCap);
    bool active = false, done = !searcher;
    int q = -1, node = -1, curRow = 0;
    double curLK = 0.0;
    for (;;) {
        S.wantScan = false;
        if (!done) do {                 // (`continue` below ends this pass of the state machine)
        if (!active) {
            q = atomicAdd(counter, 1);
            if (q >= n) { done = true; break; }
            node = nodes[q];
            setWs(wsBase + lane * LB.total, LB, L);                           // (the last search may have moved into the shared pool)
            ws.usedW = ws.usedA = ws.nH = ws.sp = ws.nB = 0;
            ws.overflow = 0;
            S.nAppend = 0;
            SearchOut &o = out[q];
            o.bestNode = -1; o.placement = -1; o.status = 0; o.nAppend = 0;
            o.bestScore = 0.0; o.improvement = 0.0; o.currentLK = 0.0;
            o.blen[0] = o.blen[1] = o.blen[2] = 0.0;
            o.rprWoff = o.rprAoff = -1; o.rprN = o.rprNA = 0;
            o.nShortList = o.nSteps = 0; o.tStep = o.tReplay = o.tRefine = 0;
            const int parent = T.nd[node].up;
            if (parent < 0) { o.status = 1; continue; }               // the root cannot be re-placed (M:9626)
            // current placement cost, M:9629-9646
            const int childIdx = (T.nd[parent].c0 == node) ? 0 : 1;
            int vectUp = S.opPass(S.treeList(childIdx == 0 ? T.nd[parent].upRight : T.nd[parent].upLeft), T.nd[node].mutId, false);
            if (!S.valid(vectUp)) { o.status = ws.overflow ? -3 : -1; continue; }
            curLK = append_walk(c, S.ref(vectUp), S.ref(S.treeList(T.nd[node].lower)), T.nd[node].isTip != 0, T.nd[node].dist);
            o.currentLK = curLK;
            if (!(curLK < P.thrPlacement || T.nd[node].dist != 0.0)) { o.status = 2; continue; }   // M:9674
            ws.usedW = ws.usedA = ws.nH = 0;
            const size_t row = cacheRow ? (size_t)cacheRow[q] : (size_t)q;
            curRow = (int)row;
            S.cached = cacheS ? cacheS + row * T.n : nullptr;             // this query's row of the (queries x nodes) score table
            S.finMask = (cacheS && fin.mask) ? fin.mask + row * fin.nWords : nullptr;
            S.rTable = (cacheS && rTable) ? rTable + row * nF : nullptr;  // and of the (queries x frames) removed lists
            S.fShort[0] = S.fShort[1] = S.fShort[2] = S.fShort[3] = -1;
            // A pruned node on a zero-length branch is searched with removedBLen = 0 (M:9644).  Without an error model a
            // mismatch over zero length is impossible (-inf, M:6663), -inf never counts as a failed pass, and the search
            // walks the whole tree by the reference's own rules (tools/wide_stats.py, 100 000-tip bench tree: all 37 536
            // whole-tree searches of a deep round sit on zero-length branches and none of the other 136 494 does): it goes
            // to the dense tier after a token budget instead of spending the full one here first.  With an error model the
            // mismatch has a finite cost and such searches end like any other (1 000 000-tip run), so no hint then.  (A
            // routing hint only: the dense tier runs the same search from the start.)
            S.budget = (!U && budget > MAPLE_ZERO_DIST_BUDGET && T.nd[node].dist == 0.0) ? MAPLE_ZERO_DIST_BUDGET : budget;
            S.overBudget = false;
            S.lean = assist && q != traceQuery;                            // (a traced query keeps to step(), which records its visits)
            S.wantApp = S.haveMail = S.stepOnce = false;
            S.trI = nullptr;
            if (q == traceQuery && trI) { S.trI = trI; S.trD = trD; S.trCap = trCap; S.trN = 0; }
            S.begin(parent, childIdx, curLK, T.nd[node].dist);
            active = true;
        } else if (ws.overflow) {
            // workspace exhausted: in the budgeted pass the search is simply handed to the batch path like a wide one (it
            // restarts there with 4x the room and allocates far less once scores are cached); otherwise the host retries
            // it with more
            out[q].status = (S.budget > 0 && !S.cached) ? -5 : -3;
            out[q].nAppend = ws.overflow;                             // (which capacity, for MAPLE_DEBUG)
            out[q].bestNode = -2;                                     // marks "handed over because it ran out of room"
            active = false;
        } else if (S.overBudget) {
            out[q].status = -5;                                       // a wide search: the host batch-scores it and re-runs it
            active = false;
        } else if (ws.sp > 0) {
#ifdef MAPLE_SPR_PROFILE
            const long long t0 = wall_clock64();
            const bool notUpd = !ws.st[ws.sp - 1].upd;
            const bool rep = (S.cached || S.lean) && notUpd && !S.stepOnce;
            if (rep) S.replayCached(); else { S.stepOnce = false; S.step(); if (!notUpd) out[q].nSteps++; }
            if (notUpd) out[q].tReplay += wall_clock64() - t0; else out[q].tStep += wall_clock64() - t0;
#else
            if ((S.cached || S.lean) && !ws.st[ws.sp - 1].upd && !S.stepOnce) S.replayCached();
            else { S.stepOnce = false; S.step(); }
#endif
        } else if (S.refineIdx < ws.nB) {
#ifdef MAPLE_SPR_PROFILE
            const long long t0 = wall_clock64();
            out[q].nShortList++;
            int r = S.refine(ws.best[S.refineIdx++]);
            out[q].tRefine += wall_clock64() - t0;
#else
            int r = S.refine(ws.best[S.refineIdx++]);
#endif
            if (r < 0 && !ws.overflow) {                              // the reference raises here; its worker swallows it (M:9703)
                SearchOut &o = out[q];
                o.status = -1; o.nAppend = S.nAppend;
                active = false;
            }
        } else {
            SearchOut &o = out[q];
            o.bestNode = S.bestNode; o.bestScore = S.bestScore;
            o.blen[0] = S.bl0; o.blen[1] = S.bl1; o.blen[2] = S.bl2;
            o.nAppend = S.nAppend;
#ifdef MAPLE_SPR_PROFILE
            o.rprN = (int32_t)(S.tWalk / 100); o.rprNA = (int32_t)(S.tRefSetup / 100);   // (profile: microseconds inside append_walk / list lookup)
            S.tWalk = S.tRefSetup = 0;
#endif
            if (S.trI) *trN = S.trN;
            if (poolW) {                                             // hand bestRemovedPartials out through the pool
                int hOut = S.hBestRpr;
                if (hOut <= -10 && S.rTable)                          // a frame-table list the reference shortened in place
                    for (int k = 0; k < 4; k++)
                        if (S.fShort[k] >= 0 && -(hOut + 10) == S.rTable[S.fShort[k]]) { hOut = S.opShortenCopy(hOut); break; }
                TList rp = S.L(hOut);
                long long ow = (long long)atomicAdd(&poolUsed[0], (unsigned long long)rp.n);
                long long oa = (long long)atomicAdd(&poolUsed[1], (unsigned long long)rp.na);
                if (ow + rp.n <= poolCapW && oa + rp.na <= poolCapA) {
                    for (int k = 0; k < rp.n; k++) poolW[ow + k] = rp.w[k];
                    for (int k = 0; k < rp.na; k++) poolA[oa + k] = rp.aux[k];
                    o.rprWoff = ow; o.rprAoff = oa; o.rprN = rp.n; o.rprNA = rp.na;
                } else o.status = -4;
            }
            // accept rule and the four "same place" vetoes, M:9681-9700
            if (S.bestScore + P.thrPlacement > curLK) {
                bool updated = true;
                int topNode = T.nd[node].up;
                if (S.bestNode == topNode) updated = false;
                while (T.nd[topNode].dist == 0.0 && T.nd[topNode].up >= 0) topNode = T.nd[topNode].up;
                if (S.bestNode == topNode && S.bl1 == 0.0) updated = false;
                const int par = T.nd[node].up;
                const int sib = (T.nd[par].c0 == node) ? T.nd[par].c1 : T.nd[par].c0;
                if (S.bestNode == sib) updated = false;
                if (T.nd[S.bestNode].up == sib && S.bl0 == 0.0) updated = false;
                if (updated) { o.improvement = S.bestScore - curLK; o.placement = S.bestNode; }
            }
            active = false;
        }
        } while (0);
        if constexpr (ASSIST) if (assist) {
            // ---- every lane is here: the scores the searching lanes asked for, one wavefront-wide walk each ----
            unsigned long long req = __ballot(searcher && !done && S.wantApp);
            if (__popcll(req) > coopMax) {
                // many lanes ask at once: each walks its own pair, all of them in lockstep (one walk's latency for all of
                // them, a third of the instructions per score of the wavefront-wide form); the wavefront-wide form is for
                // the few lanes still searching when the others are done
                if (searcher && !done && S.wantApp) {
                    const double v = append_walk(c, S.ref(S.treeList(T.nd[S.appT1].totUp)), S.ref(S.appHRpr), S.isRemovedTip, S.removedBLen);
                    S.mailScore = v; S.mailNode = S.appT1; S.haveMail = true; S.wantApp = false;
                }
                req = 0;
            }
            // (every asking lane looks its two lists up first, all of them at once: the look-ups are chains of dependent loads)
            TList tp{nullptr, nullptr, 0, 0}, tc{nullptr, nullptr, 0, 0};
            if (req && searcher && !done && S.wantApp) { tp = S.L(S.treeList(T.nd[S.appT1].totUp)); tc = S.L(S.appHRpr); }
            while (req) {
                const int r = (int)__ffsll((long long)req) - 1;
                req &= req - 1;
                auto bc64 = [&](unsigned long long x) {
                    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(x >> 32), r) << 32)
                           | (uint32_t)__builtin_amdgcn_readlane((int)x, r);
                };
                const ListRef P{(const uint2 *)bc64((unsigned long long)tp.w), (const double *)bc64((unsigned long long)tp.aux)};
                const ListRef Cq{(const uint2 *)bc64((unsigned long long)tc.w), (const double *)bc64((unsigned long long)tc.aux)};
                const int nP = __builtin_amdgcn_readlane(tp.n, r), nC = __builtin_amdgcn_readlane(tc.n, r);
                const bool tipq = __builtin_amdgcn_readlane(S.isRemovedTip ? 1 : 0, r) != 0;
                const double blq = __longlong_as_double((long long)bc64((unsigned long long)__double_as_longlong(S.removedBLen)));
                const double v = wave_append(c, P, nP, Cq, nC, tipq, blq, wl);
                if ((int)threadIdx.x == r) { S.mailScore = v; S.mailNode = S.appT1; S.haveMail = true; S.wantApp = false; }
            }
            if (!__ballot(!done)) break;
            continue;
        }
        if (!coop) {
            if (done) break;
            continue;
        }
        // ---- wave-level part: every lane is here, lane 0 decides ----
        if (__builtin_amdgcn_readfirstlane(S.wantScan ? 1 : 0)) {
            const int rowU = __builtin_amdgcn_readfirstlane(curRow);
            const double *cs = cacheS + (size_t)rowU * T.n;
            const int32_t *rT = rTable ? rTable + (size_t)rowU * nF : nullptr;
            ScanState st;
            st.best = readfirst_f64(S.bestLKdiff);
            st.nB = __builtin_amdgcn_readfirstlane(ws.nB);
            st.nApp = __builtin_amdgcn_readfirstlane(S.nAppend);
            st.overflow = 0;
            st.shortenSeed = false;
            for (int k = 0; k < 4; k++) st.fShort[k] = __builtin_amdgcn_readfirstlane(S.fShort[k]);
            const int hSeed = __builtin_amdgcn_readfirstlane(S.scanItem.hRpr);
            const unsigned long long brBits = (unsigned long long)ws.best;  // (lane 0's short list)
            BestRec *br = (BestRec *)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(brBits >> 32)) << 32)
                                      | (uint32_t)__builtin_amdgcn_readfirstlane((int)brBits));
            const int capBnow = __builtin_amdgcn_readfirstlane(ws.L.capB);
#ifdef MAPLE_SPR_PROFILE
            const long long tScan0 = wall_clock64();
#endif
            wave_scan_clade(T.scan, T.scanParent, cs, rT, __builtin_amdgcn_readfirstlane(S.scanRank),
                            __builtin_amdgcn_readfirstlane(S.scanFirstScored ? 1 : 0) != 0,
                            __builtin_amdgcn_readfirstlane(S.scanSeedFrame), hSeed, readfirst_f64(S.scanItem.lastLK),
                            __builtin_amdgcn_readfirstlane((int)S.scanItem.fails), P, br, capBnow, slotLK, slotFails,
                            slotOwner, T.scanDepthCap, st, fin.mask ? fin.mask + (size_t)rowU * fin.nWords : nullptr,
                            fin.mask ? fin.prefix + (size_t)rowU * (fin.nWords + 1) : nullptr, T.candBefore, T.cladeVisits);
#ifdef MAPLE_SPR_PROFILE
            if (searcher) out[q].tReplay += wall_clock64() - tScan0;
#endif
            if (searcher) {
                S.bestLKdiff = st.best; ws.nB = st.nB; S.nAppend = st.nApp;
                if (st.overflow && !ws.overflow) ws.overflow = st.overflow;
                for (int k = 0; k < 4; k++) S.fShort[k] = st.fShort[k];
                if (st.shortenSeed) S.opShortenInPlace(hSeed);
            }
            continue;
        }
        if (__builtin_amdgcn_readfirstlane(done ? 1 : 0)) break;
    }
}

#define MAPLE_SPR_KERNEL_ARGS const DevModel *__restrict__ mp, ArenaView av, MutView mv, DevTree T, SearchParams P, int n,            \
    const int32_t *nodes, WsLayout L, LaneBytes LB, uint8_t *wsBase, int32_t *counter, SearchOut *out, uint2 *poolW, double *poolA, \
    unsigned long long *poolUsed, long long poolCapW, long long poolCapA, int traceQuery, int32_t *trI, double *trD, int trCap,     \
    int32_t *trN, int activeLanes, const double *cacheS, int budget, const int32_t *rTable, int nF, const int32_t *cacheRow,        \
    int leanVisits, unsigned long long *ovfUsed, uint8_t *ovfBase, long long ovfChunks, FiniteRows fin
#define MAPLE_SPR_KERNEL_PASS mp, av, mv, T, P, n, nodes, L, LB, wsBase, counter, out, poolW, poolA, poolUsed, poolCapW, poolCapA,   \
    traceQuery, trI, trD, trCap, trN, activeLanes, cacheS, budget, rTable, nF, cacheRow, leanVisits, ovfUsed, ovfBase, ovfChunks, fin
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) MAPLE_SPR_ATTR void k_spr_search(MAPLE_SPR_KERNEL_ARGS)
{
    spr_search_impl<RV, U, SS, false>(MAPLE_SPR_KERNEL_PASS);
}
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) MAPLE_SPR_ATTR void k_spr_search_assisted(MAPLE_SPR_KERNEL_ARGS)
{
    spr_search_impl<RV, U, SS, true>(MAPLE_SPR_KERNEL_PASS);
}


// MAT reference frames of the uploaded tree: frame 0 is the root's reference, every node whose branch carries mutations
// opens a new one for its clade.  Frames are numbered by nesting depth (a parent frame always has a smaller index).
static int compute_frames(maple_ctx *c)
{
    PlaceMeta &M = *c->place;
    const int32_t n = c->dtree.n, root = c->dtree.root;
    const auto &up = c->h_tree_up;
    const auto &c0 = c->h_tree_c0, &c1 = c->h_tree_c1, &mut = c->h_tree_mut;
    std::vector<int32_t> &order = M.order;
    std::vector<int32_t> depth(n, 0), fdepth;
    order.clear();
    order.reserve(n);
    std::vector<int32_t> st{root};
    M.frameOf.assign(n, -1);
    M.frameNode.assign(1, -1);
    M.frameParent.assign(1, -1);
    fdepth.assign(1, 0);
    M.maxDepth = 0;
    while (!st.empty()) {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        const int32_t pf = up[v] < 0 || v == root ? 0 : M.frameOf[up[v]];
        if (v != root) depth[v] = depth[up[v]] + 1;
        if (depth[v] > M.maxDepth) M.maxDepth = depth[v];
        if (mut[v] >= 0) {
            M.frameOf[v] = (int32_t)M.frameNode.size();
            M.frameNode.push_back(v);
            M.frameParent.push_back(pf);
            fdepth.push_back(fdepth[pf] + 1);
        } else M.frameOf[v] = pf;
        if (c0[v] >= 0) { st.push_back(c0[v]); st.push_back(c1[v]); }
    }
    // renumber frames by nesting depth so that a level is a contiguous range (parents always in earlier levels)
    const int32_t nF = (int32_t)M.frameNode.size();
    std::vector<int32_t> perm(nF), inv(nF);
    for (int i = 0; i < nF; i++) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return fdepth[a] < fdepth[b]; });
    for (int i = 0; i < nF; i++) inv[perm[i]] = i;
    std::vector<int32_t> fn(nF), fp(nF);
    M.levelStart.clear();
    for (int i = 0; i < nF; i++) {
        fn[i] = M.frameNode[perm[i]];
        fp[i] = M.frameParent[perm[i]] < 0 ? -1 : inv[M.frameParent[perm[i]]];
        if (i > 0 && fdepth[perm[i]] != fdepth[perm[i - 1]]) M.levelStart.push_back(i);
    }
    M.levelStart.push_back(nF);
    M.frameNode.swap(fn);
    M.frameParent.swap(fp);
    for (auto &f : M.frameOf) if (f >= 0) f = inv[f];
    M.nF = nF;
    for (auto &f : M.frameOf) if (f < 0) f = 0;                           // nodes not reachable from the root
    c->h_depth.swap(depth);                                               // (levels below the root; 0 for what the root does not reach)
    return MAPLE_OK;
}

// ---- tree mirror + SPR search ------------------------------------------------------------------------
extern "C" int maple_tree_upload(maple_ctx *c, int32_t n, int32_t root, const int32_t *up, const int32_t *child0,
                                 const int32_t *child1, const double *dist, const uint8_t *isTip, const int32_t *lower,
                                 const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp, const int32_t *mutList)
{
    if (!c || n <= 0 || root < 0 || root >= n || !up || !child0 || !child1 || !dist || !isTip || !lower || !upRight || !upLeft
        || !totUp || !mutList)
        return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    if (c->ahead) { c->ahead->join(); c->ahead->spec.row = -1; c->ahead->active = false; }
    const bool dbgU = c->tuning.verbose != 0;
    auto tU0 = std::chrono::steady_clock::now();
    auto lapU = [&](const char *what) {
        if (!dbgU) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[maple] tree upload: %s %.1f ms\n", what, std::chrono::duration_cast<std::chrono::microseconds>(t - tU0).count() * 1e-3);
        tU0 = t;
    };
    TRY(check_ids(c, n, lower, true, "lower"));
    TRY(check_ids(c, n, upRight, true, "upRight"));
    TRY(check_ids(c, n, upLeft, true, "upLeft"));
    TRY(check_ids(c, n, totUp, true, "totUp"));
    const int32_t nml = (int32_t)c->h_mut_cnt.size();
    for (int i = 0; i < n; i++)
        if (mutList[i] >= nml) return fail(c, MAPLE_ERR_ARG, "mutList[%d] is not a mutation-list id", i);
    // the topology arrays are trusted by everything below (depth-first orders, frames, the kernels): check them here
    if (up[root] >= 0) return fail(c, MAPLE_ERR_ARG, "the root (%d) has a parent", root);
    for (int i = 0; i < n; i++) {
        if (up[i] < -1 || up[i] >= n || child0[i] < -1 || child0[i] >= n || child1[i] < -1 || child1[i] >= n)
            return fail(c, MAPLE_ERR_ARG, "node %d: up / child index out of range", i);
        if ((child0[i] >= 0) != (child1[i] >= 0)) return fail(c, MAPLE_ERR_ARG, "node %d has exactly one child", i);
        if (child0[i] >= 0 && (child0[i] == child1[i] || child0[i] == i || child1[i] == i))
            return fail(c, MAPLE_ERR_ARG, "node %d: malformed children", i);
    }
    {   // every node reachable from the root must be the child its parent says it is, and be reached once (no cycles)
        std::vector<uint8_t> seen((size_t)n, 0);
        std::vector<int32_t> st{root};
        while (!st.empty()) {
            const int32_t v = st.back();
            st.pop_back();
            if (seen[v]) return fail(c, MAPLE_ERR_ARG, "node %d is reached twice from the root (cycle or shared child)", v);
            seen[v] = 1;
            if (child0[v] >= 0) {
                if (up[child0[v]] != v || up[child1[v]] != v) return fail(c, MAPLE_ERR_ARG, "children of node %d do not point back to it", v);
                st.push_back(child0[v]); st.push_back(child1[v]);
            }
        }
    }
    lapU("columns checked, reachability");
    const int32_t *src[9] = {up, child0, child1, lower, upRight, upLeft, totUp, mutList, nullptr};
    for (int k = 0; k < 8; k++) TRY(h2d(c, c->t_i32[k], src[k], (size_t)n));
    TRY(h2d(c, c->t_dist, dist, (size_t)n));
    TRY(h2d(c, c->t_tip, isTip, (size_t)n));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->h_tree_up.assign(up, up + n);
    c->h_tree_lower.assign(lower, lower + n);
    c->h_tree_dist.assign(dist, dist + n);
    c->h_tree_tip.assign(isTip, isTip + n);
    c->h_tree_c0.assign(child0, child0 + n);
    c->h_tree_c1.assign(child1, child1 + n);
    c->h_tree_mut.assign(mutList, mutList + n);
    c->h_tree_totUp.assign(totUp, totUp + n);
    c->h_tree_upRight.assign(upRight, upRight + n);
    c->h_tree_upLeft.assign(upLeft, upLeft + n);
    if (!c->place) c->place = new PlaceMeta();
    c->place->valid = false;
    c->place->rootVect = -1;
    c->dtree.n = n; c->dtree.root = root;
    lapU("columns on the device and in the host copy");
    TRY(compute_frames(c));
    lapU("reference frames");
    const PlaceMeta &F = *c->place;
    std::vector<NodeRec> recs((size_t)n);
    for (int i = 0; i < n; i++) {
        NodeRec &r = recs[i];
        memset(&r, 0, sizeof r);
        r.up = up[i]; r.c0 = child0[i]; r.c1 = child1[i];
        r.lower = lower[i]; r.upRight = upRight[i]; r.upLeft = upLeft[i]; r.totUp = totUp[i];
        r.mutId = mutList[i]; r.dist = dist[i]; r.isTip = isTip[i];
        r.upIsRoot = (up[i] >= 0 && up[up[i]] < 0) ? 1 : 0;
        r.whichChild = (up[i] >= 0 && child1[up[i]] == i) ? 1 : 0;
        r.preRank = i;
        r.frameOf = F.frameOf[i];
        r.c0Frame = child0[i] >= 0 ? F.frameOf[child0[i]] : r.frameOf;
        r.c1Frame = child1[i] >= 0 ? F.frameOf[child1[i]] : r.frameOf;
        r.upFrame = up[i] >= 0 ? F.frameOf[up[i]] : r.frameOf;
    }
    std::vector<int32_t> rankOrder;                                    // node of every depth-first rank
    {   // depth-first ranks in the order the searches descend (the child pushed last, child 1, is visited first): the order
        // compute_frames walked the tree in (PlaceMeta::order; it left the depths in h_depth) -- not walked a third time
        std::vector<uint8_t> seen((size_t)n, 0);
        int32_t next = 0;
        rankOrder.assign((size_t)n, 0);
        for (const int32_t v : F.order) { seen[v] = 1; recs[v].preRank = next; rankOrder[(size_t)next++] = v; }
        for (int i = 0; i < n; i++) if (!seen[i]) { recs[i].preRank = next; rankOrder[(size_t)next++] = i; }   // nodes not reachable from the root
        c->h_clade.clear();
    }
    lapU("node records, depth-first ranks, depths");
    HIPCK(c, c->t_nodes.reserve(((size_t)n + (size_t)n / 8 + 1024) * sizeof(NodeRec) + 64));   // (room for the nodes patches add)
    uint8_t *aligned = (uint8_t *)(((uintptr_t)c->t_nodes.p + 63) & ~(uintptr_t)63);
    HIPCK(c, hipMemcpy(aligned, recs.data(), (size_t)n * sizeof(NodeRec), hipMemcpyHostToDevice));
    c->h_nodes = recs;
    c->nodes_current = true;
    DevTree &T = c->dtree;
    T.n = n; T.root = root;
    T.nd = (const NodeRec *)aligned;
    T.totUp = c->t_i32[6].p;
    c->scan_valid = false;
    c->cand_root_end = -1;
    c->h_over_hint.clear();
    T.scan = nullptr; T.scanParent = nullptr; T.scanDepthCap = 0;
    c->tree_has_mut = false;
    c->tree_max_ent = 0;
    for (int i = 0; i < n; i++) {
        if (mutList[i] >= 0) c->tree_has_mut = true;
        for (const int32_t *col : {lower, upRight, upLeft, totUp})
            if (col[i] >= 0) c->tree_max_ent = std::max(c->tree_max_ent, c->h_n_ent[col[i]]);
    }
    lapU("node records on the device, longest list");
    {   // The dense scoring of the whole-tree searches takes its candidates in the searches' own depth-first order: the 64
        // scores of a tile then land next to each other in the search's row of the score table (a contiguous 512-byte
        // store instead of 64 partial-line stores, which WRITE_SIZE counts 4x), and neighbours in the tree have lists of
        // similar length anyway.  Measured at 100 000 tips: 1 017 -> 940 ms per round against candidates sorted by length.
        // (ranks are a permutation: the candidates in rank order are read off the rank table, and "by frame, then by rank" is one
        // stable counting pass over that -- two comparison sorts of 2 M nodes through their 100-byte records were 1.4 s of a
        // 1.7 s upload at 1 000 000 tips)
        std::vector<int32_t> byRank;
        byRank.reserve((size_t)n);
        for (int r = 0; r < n; r++) { const int v = rankOrder[r]; if (totUp[v] >= 0) byRank.push_back(v); }
        std::vector<int32_t> col;
        if (c->tree_has_mut) {                                        // by reference frame, then depth-first: a chunk of 64
            std::vector<int64_t> start((size_t)F.nF + 1, 0);          // candidates shares ONE copy of the query
            for (int v : byRank) start[(size_t)recs[v].frameOf + 1]++;
            for (int f = 0; f < F.nF; f++) start[f + 1] += start[f];
            col.resize(byRank.size());
            for (int v : byRank) col[(size_t)start[recs[v].frameOf]++] = v;
        } else col = byRank;
        {   // (in rank order whatever the tree: what the rows that come with bitmaps are indexed by, FiniteRows)
            c->h_cand_ids.resize(byRank.size()); c->h_cand_rank.resize(byRank.size()); c->h_cand_frame.resize(byRank.size());
            for (size_t i = 0; i < byRank.size(); i++) {
                c->h_cand_ids[i] = totUp[byRank[i]]; c->h_cand_rank[i] = recs[byRank[i]].preRank; c->h_cand_frame[i] = recs[byRank[i]].frameOf;
            }
            TRY(h2d(c, c->t_cand_rank, c->h_cand_rank.data(), c->h_cand_rank.size()));
        }
        std::vector<int32_t> ids(col.size()), rank(col.size()), fr(col.size());
        for (size_t i = 0; i < col.size(); i++) { ids[i] = totUp[col[i]]; rank[i] = recs[col[i]].preRank; fr[i] = recs[col[i]].frameOf; }
        c->n_frame_chunks = 0;
        if (c->tree_has_mut) {
            std::vector<int4> chunks;
            for (size_t i = 0; i < col.size();) {
                size_t j = i;
                while (j < col.size() && j - i < 64 && fr[j] == fr[i]) j++;
                chunks.push_back(make_int4((int)i, (int)(j - i), fr[i], 0));
                i = j;
            }
            TRY(h2d(c, c->t_frame_chunks, chunks.data(), chunks.size()));
            c->n_frame_chunks = (int32_t)chunks.size();
        }
        c->n_scored = (int32_t)col.size();
        c->scored_bytes_total = 0.0;                                       // 8 E + 8 A + 8 (result) per candidate
        for (size_t i = 0; i < col.size(); i++) c->scored_bytes_total += 8.0 * c->h_n_ent[ids[i]] + 8.0 * c->h_n_aux[ids[i]] + 8.0;
        TRY(h2d(c, c->t_i32[8], ids.data(), ids.size()));
        TRY(h2d(c, c->t_scored_col, rank.data(), rank.size()));
        TRY(h2d(c, c->t_scored_frame, fr.data(), fr.size()));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    lapU("candidate order");
    c->tree_set = true;
    c->tree_stale = false;
    return MAPLE_OK;
}

// every device table rebuilt from the host's own copy of the tree (after maple_tree_patch, before a search that needs them)
int tree_rebuild_from_host(maple_ctx *c)
{
    const std::vector<int32_t> up = c->h_tree_up, c0 = c->h_tree_c0, c1 = c->h_tree_c1, lower = c->h_tree_lower,
                               upRight = c->h_tree_upRight, upLeft = c->h_tree_upLeft, totUp = c->h_tree_totUp, mut = c->h_tree_mut;
    const std::vector<double> dist = c->h_tree_dist;
    const std::vector<uint8_t> tip = c->h_tree_tip;
    std::vector<uint8_t> hint;                                         // (the same tree, patched: what its nodes' searches did last time still holds)
    hint.swap(c->h_over_hint);
    const int rc = maple_tree_upload(c, (int32_t)up.size(), c->dtree.root, up.data(), c0.data(), c1.data(), dist.data(), tip.data(),
                                     lower.data(), upRight.data(), upLeft.data(), totUp.data(), mut.data());
    c->h_over_hint.swap(hint);
    return rc;
}

// maple_tree_patch's device writes in two launches instead of one small copy (and one synchronisation) per word: node records
// by index, then single words by address
struct PatchPoke { int32_t *p; int32_t v; int32_t pad; };
__global__ __launch_bounds__(256) void k_patch_nodes(int nR, const int32_t *idx, const NodeRec *recs, NodeRec *dn)
{
    constexpr int W = (int)(sizeof(NodeRec) / 4);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nR * W; k += gridDim.x * blockDim.x)
        ((int32_t *)(dn + idx[k / W]))[k % W] = ((const int32_t *)(recs + k / W))[k % W];
}
__global__ __launch_bounds__(256) void k_patch_pokes(int nP, const PatchPoke *pk)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nP; k += gridDim.x * blockDim.x) *pk[k].p = pk[k].v;
}

// A local change of the uploaded tree -- what placeSampleOnTree (M:8300-8722) and the updatePartials after it leave behind:
// a few nodes with new relatives, branch lengths or list ids, one or two new nodes.  nodes[i] gets the record
// (up, child0, child1, dist, isTip, lower, upRight, upLeft, totUp)[i]; ids >= the old node count are new nodes (all of them
// must be listed; nTotal = the new count).  Mutation lists (MAT reference nodes) and the root do not change this way:
// re-upload the tree for that.  The host copy of the tree and the candidate / leaf columns of the placement search are
// updated in place (a few 4-byte writes); the linearised tables of the batched placement search and of the SPR search are
// only marked stale and are rebuilt from the host copy before their next use -- so the serial phase (one placement, one
// patch, one placement, ...) never pays for the whole tree.
extern "C" int maple_tree_patch(maple_ctx *c, int32_t nTotal, int32_t nTouched, const int32_t *nodes, const int32_t *up,
                                const int32_t *child0, const int32_t *child1, const double *dist, const uint8_t *isTip,
                                const int32_t *lower, const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp)
{
    if (!c || nTouched < 0 || (nTouched && (!nodes || !up || !child0 || !child1 || !dist || !isTip || !lower || !upRight || !upLeft || !totUp)))
        return MAPLE_ERR_ARG;
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    HIPCK(c, hipSetDevice(c->device));
    if (c->ahead) c->ahead->join();                                       // (a speculative traversal reads the per-node records this call rewrites)
    const int32_t nOld = (int32_t)c->h_tree_up.size(), root = c->dtree.root;
    if (nTotal < nOld) return fail(c, MAPLE_ERR_ARG, "a patch cannot remove nodes (%d < %d)", nTotal, nOld);
    const int32_t nl = (int32_t)c->h_n_ent.size();
    std::vector<uint8_t> seenNew((size_t)(nTotal - nOld), 0);
    for (int i = 0; i < nTouched; i++) {
        const int v = nodes[i];
        if (v < 0 || v >= nTotal) return fail(c, MAPLE_ERR_ARG, "nodes[%d] = %d is not a node", i, v);
        if (v >= nOld) seenNew[v - nOld] = 1;
        if (up[i] < -1 || up[i] >= nTotal || child0[i] < -1 || child0[i] >= nTotal || child1[i] < -1 || child1[i] >= nTotal
            || (child0[i] < 0) != (child1[i] < 0))
            return fail(c, MAPLE_ERR_ARG, "node %d: relative out of range, or only one child", v);
        if ((up[i] < 0) != (v == root)) return fail(c, MAPLE_ERR_ARG, "node %d: the root cannot change in a patch", v);
        for (int32_t id : {lower[i], upRight[i], upLeft[i], totUp[i]})
            if (id < -1 || id >= nl) return fail(c, MAPLE_ERR_ARG, "node %d: %d is not a list id", v, id);
    }
    for (size_t k = 0; k < seenNew.size(); k++)
        if (!seenNew[k]) return fail(c, MAPLE_ERR_ARG, "new node %d is not in the patch", nOld + (int)k);
    {   // relatives must point back at each other in the tree AS PATCHED -- checked before anything changes, so that a
        // malformed patch leaves the library's copy as it was (the traversals trust these columns)
        // (position of a node in the patch, -1 if it is not in it: a sorted copy of the handful of touched nodes, not a tree-sized
        // table -- at 1 000 000 tips that table was 8 MB written per patch)
        std::vector<std::pair<int32_t, int32_t>> where((size_t)nTouched);
        for (int i = 0; i < nTouched; i++) where[i] = {nodes[i], i};
        std::sort(where.begin(), where.end());
        auto at = [&](int v) -> int {
            auto it = std::lower_bound(where.begin(), where.end(), std::make_pair((int32_t)v, (int32_t)-1));
            return (it != where.end() && it->first == v) ? it->second : -1;
        };
        auto upOf = [&](int v) { const int a = at(v); return a >= 0 ? up[a] : c->h_tree_up[v]; };
        auto c0Of = [&](int v) { const int a = at(v); return a >= 0 ? child0[a] : c->h_tree_c0[v]; };
        auto c1Of = [&](int v) { const int a = at(v); return a >= 0 ? child1[a] : c->h_tree_c1[v]; };
        for (int i = 0; i < nTouched; i++) {
            const int v = nodes[i];
            for (int32_t ch : {child0[i], child1[i]})
                if (ch >= 0 && (ch == v || upOf(ch) != v)) return fail(c, MAPLE_ERR_ARG, "node %d: child %d does not point back to it", v, ch);
            if (child0[i] >= 0 && child0[i] == child1[i]) return fail(c, MAPLE_ERR_ARG, "node %d: the same child twice", v);
            if (up[i] >= 0 && c0Of(up[i]) != v && c1Of(up[i]) != v)
                return fail(c, MAPLE_ERR_ARG, "node %d is not a child of its parent %d", v, up[i]);
        }
    }
    // ---- the host copy
    for (auto *vec : {&c->h_tree_up, &c->h_tree_c0, &c->h_tree_c1, &c->h_tree_lower, &c->h_tree_upRight, &c->h_tree_upLeft,
                      &c->h_tree_totUp, &c->h_tree_mut})
        vec->resize((size_t)nTotal, -1);
    c->h_tree_dist.resize((size_t)nTotal, 0.0);
    c->h_tree_tip.resize((size_t)nTotal, 0);
    for (int i = 0; i < nTouched; i++) {
        const int v = nodes[i];
        c->h_tree_up[v] = up[i]; c->h_tree_c0[v] = child0[i]; c->h_tree_c1[v] = child1[i];
        c->h_tree_dist[v] = dist[i]; c->h_tree_tip[v] = isTip[i];
        c->h_tree_lower[v] = lower[i]; c->h_tree_upRight[v] = upRight[i]; c->h_tree_upLeft[v] = upLeft[i]; c->h_tree_totUp[v] = totUp[i];
    }
    c->dtree.n = nTotal;
    c->tree_stale = true;
    c->scan_valid = false;
    c->cand_root_end = -1;
    c->h_clade.clear();
    // ---- the node records of the SPR search (search_dev.h): the touched nodes and their relatives are rewritten in place, so
    // that a small batch of searches -- the re-search of a proposed move before it is applied, M:9470-9484 -- can run on the
    // patched tree at once (frontier tier, no tree-sized table); everything tree-sized (depth-first orders, score columns)
    // waits for the rebuild
    bool inFlight = false;                                                // (a staged launch is queued: awaited once, before returning)
    auto settle_patch = [&](int rc) -> int {
        if (inFlight && hipStreamSynchronize(c->stream) != hipSuccess && rc == MAPLE_OK) return fail(c, MAPLE_ERR_HIP, "maple_tree_patch: hipStreamSynchronize failed");
        return rc;
    };
    if (c->nodes_current && !c->tree_has_mut && c->dtree.nd) {
        const size_t capNodes = (c->t_nodes.cap - 64) / sizeof(NodeRec);
        if ((size_t)nTotal > capNodes) c->nodes_current = false;
        else {
            c->h_nodes.resize((size_t)nTotal);
            std::vector<int32_t> redo;
            for (int i = 0; i < nTouched; i++) {
                const int v = nodes[i];
                redo.push_back(v);
                for (int32_t w : {c->h_tree_up[v], c->h_tree_c0[v], c->h_tree_c1[v]}) if (w >= 0) redo.push_back(w);
            }
            std::sort(redo.begin(), redo.end());
            redo.erase(std::unique(redo.begin(), redo.end()), redo.end());
            NodeRec *dn = const_cast<NodeRec *>(c->dtree.nd);
            std::vector<NodeRec> recs;
            recs.reserve(redo.size());
            for (int32_t v : redo) {
                NodeRec &r = c->h_nodes[v];
                const int32_t keepRank = v < nOld ? r.preRank : 0;
                memset(&r, 0, sizeof r);
                r.up = c->h_tree_up[v]; r.c0 = c->h_tree_c0[v]; r.c1 = c->h_tree_c1[v];
                r.lower = c->h_tree_lower[v]; r.upRight = c->h_tree_upRight[v]; r.upLeft = c->h_tree_upLeft[v]; r.totUp = c->h_tree_totUp[v];
                r.mutId = -1; r.dist = c->h_tree_dist[v]; r.isTip = c->h_tree_tip[v];
                r.upIsRoot = (r.up >= 0 && c->h_tree_up[r.up] < 0) ? 1 : 0;
                r.whichChild = (r.up >= 0 && c->h_tree_c1[r.up] == v) ? 1 : 0;
                r.preRank = keepRank;                                          // (stale: only the tree-sized tables use it)
                recs.push_back(r);
            }
            if (!redo.empty()) {
                TRY(stage_begin(c, redo.size() * (sizeof(NodeRec) + 8) + 256));
                STAGE(dIdx, c, redo.data(), redo.size());
                STAGE(dRec, c, recs.data(), recs.size());
                TRY(stage_flush(c));
                k_patch_nodes<<<1, 256, 0, c->stream>>>((int)redo.size(), dIdx, dRec, dn);
                HIPCK(c, hipGetLastError());
                inFlight = true;
            }
        }
    } else c->nodes_current = false;
    PlaceMeta &M = *c->place;
    if (!M.valid) return settle_patch(MAPLE_OK);                          // nothing of the placement search to keep up to date
    // ---- the placement search's columns
    M.scanStale = true;
    M.frameOf.resize((size_t)nTotal, -1);
    M.h_candIdx.resize((size_t)nTotal, -1);
    M.h_leafIdx.resize((size_t)nTotal, -1);
    for (bool again = true; again;) {                                     // a new node lives in its parent's reference frame
        again = false;
        for (int i = 0; i < nTouched; i++) {
            const int v = nodes[i];
            if (M.frameOf[v] >= 0) continue;
            const int u = c->h_tree_up[v];
            if (u >= 0 && M.frameOf[u] >= 0) { M.frameOf[v] = M.frameOf[u]; again = true; }
        }
    }
    for (int i = 0; i < nTouched; i++)
        if (M.frameOf[nodes[i]] < 0) return settle_patch(fail(c, MAPLE_ERR_ARG, "new node %d is not attached to the tree", nodes[i]));
    std::vector<PatchPoke> pokes;                                         // (single words of the columns: written together at the end)
    auto poke = [&](DevBuf<int32_t> &b, size_t at, int32_t value) -> int {
        if (at >= b.cap) { M.valid = false; return MAPLE_OK; }            // out of room: the next search rebuilds everything
        pokes.push_back(PatchPoke{b.p + at, value, 0});
        return MAPLE_OK;
    };
    auto flush_pokes = [&]() -> int {
        if (pokes.empty()) return MAPLE_OK;
        if (inFlight) HIPCK(c, hipStreamSynchronize(c->stream));         // (the staging arenas alternate: the first one's copy is done)
        TRY(stage_begin(c, pokes.size() * sizeof(PatchPoke) + 256));
        STAGE(dPk, c, pokes.data(), pokes.size());
        TRY(stage_flush(c));
        k_patch_pokes<<<1, 256, 0, c->stream>>>((int)pokes.size(), dPk);
        HIPCK(c, hipGetLastError());
        inFlight = true;
        return MAPLE_OK;
    };
    // (score rows made ahead, maple_placement_ahead: the columns whose list changes here, and the new ones, are scored again for
    // the samples still waiting before the next of them is searched)
    PlaceAhead *const ah = (c->ahead && c->ahead->active) ? c->ahead : nullptr;
    if (ah && ah->spec.row == ah->next)                                   // (what the traversal made ahead for the next sample must not have visited)
        for (int i = 0; i < nTouched; i++) { ah->spec.touched.push_back(nodes[i]); if (nodes[i] == root) ah->spec.rootTouched = true; }
    auto dirty_col = [&](int col) { if (ah) { if ((int64_t)col >= ah->ld - 1) ah->active = false; else ah->dirtyCols.push_back(col); } };
    auto dirty_leaf = [&](int lc) { if (ah) { if ((int64_t)lc >= ah->ldL) ah->active = false; else ah->dirtyLeaves.push_back(lc); } };
    for (int i = 0; i < nTouched && M.valid; i++) {
        const int v = nodes[i];
        if (v == root && lower[i] != -1) { M.rootVect = -1; if (ah) ah->rootDirty = true; }   // (recomputed by the next search if the root's list changed)
        const bool cand = v != root && up[i] >= 0 && dist[i] > M.effNon0 && totUp[i] >= 0;        // M:8049
        int col = M.h_candIdx[v];
        if (col >= 0 && !cand) M.h_candIdx[v] = -1;                       // (the column stays and is scored for nothing)
        else if (col >= 0) { if (M.h_candList[col] != totUp[i]) { M.h_candList[col] = totUp[i]; TRY(poke(M.d_candList, col, totUp[i])); dirty_col(col); } }
        else if (cand) {
            col = (int)M.cand.size();                                     // the column of the root vector moves up by one
            M.cand.push_back(v);
            M.h_candIdx[v] = col;
            const int32_t rootFrame = M.h_candFrame.back();
            M.h_candList.back() = totUp[i]; M.h_candFrame.back() = M.frameOf[v];
            M.h_candList.push_back(-1); M.h_candFrame.push_back(rootFrame);
            TRY(poke(M.d_candList, col, totUp[i]));
            TRY(poke(M.d_candFrame, col, M.frameOf[v]));
            TRY(poke(M.d_candFrame, col + 1, rootFrame));
            if ((size_t)col + 1 >= M.d_candList.cap) M.valid = false;
            dirty_col(col);
        }
        const bool leaf = child0[i] < 0;
        int lc = M.h_leafIdx[v];
        if (lc >= 0 && !leaf) M.h_leafIdx[v] = -1;
        else if (leaf) {
            if (lower[i] < 0) { (void)flush_pokes(); return settle_patch(fail(c, MAPLE_ERR_STATE, "leaf %d has no lower genome list", v)); }
            if (lc >= 0) { if (M.h_leafList[lc] != lower[i]) { M.h_leafList[lc] = lower[i]; TRY(poke(M.d_leafList, lc, lower[i])); dirty_leaf(lc); } }
            else {
                lc = (int)M.leaves.size();
                M.leaves.push_back(v);
                M.h_leafIdx[v] = lc;
                M.h_leafList.push_back(lower[i]); M.h_leafFrame.push_back(M.frameOf[v]);
                TRY(poke(M.d_leafList, lc, lower[i]));
                TRY(poke(M.d_leafFrame, lc, M.frameOf[v]));
                dirty_leaf(lc);
            }
        }
    }
    if (M.valid) {                                                        // (the host traversal's per-node records follow, and their device copy)
        M.h_pn.resize((size_t)nTotal, PlaceMeta::PNode{-1, -1, -1, -1});
        for (int i = 0; i < nTouched && M.valid; i++) {
            const int v = nodes[i];
            const PlaceMeta::PNode r{M.h_candIdx[v], M.h_leafIdx[v], c->h_tree_c0[v], c->h_tree_c1[v]};
            M.h_pn[v] = r;
            if (M.d_pn.cap) {
                if ((size_t)4 * v + 3 >= M.d_pn.cap) { M.d_pn.release(); if (ah) ah->active = false; }   // (no room: no expansion until the tables are made again)
                else { TRY(poke(M.d_pn, (size_t)4 * v, r.candCol)); TRY(poke(M.d_pn, (size_t)4 * v + 1, r.leafCol)); TRY(poke(M.d_pn, (size_t)4 * v + 2, r.c0)); TRY(poke(M.d_pn, (size_t)4 * v + 3, r.c1)); }
            }
        }
    }
    { const int rc_ = flush_pokes(); if (rc_) return settle_patch(rc_); }
    if (ah && !M.valid) ah->active = false;                              // (the columns will be numbered anew)
    return settle_patch(MAPLE_OK);
}

#ifndef MAPLE_WIDE_BUDGET_DEFAULT
#define MAPLE_WIDE_BUDGET_DEFAULT 256
#endif
// ---- the removed lists of a batch of whole-tree searches in every MAT reference frame of one nesting level, on the device --
// R[k * nF + f] = list id of query k's removed list expressed in frame f (-1: not yet).  One item per (query, frame of the
// level): the list in the parent frame goes down through the mutations of the frame's node (passGenomeListThroughBranch,
// M:7119 / 7342).  Sizes, scratch offsets, arena offsets and the rows of the list table are all produced here (two prefix
// sums per level); the host only learns the totals.
__global__ __launch_bounds__(MAPLE_BLOCK) void k_fan_cap(long long nItems, int nF, int a, int nL, const int32_t *R,
                                                         const int32_t *frameParent, const int32_t *frameMut, ArenaView av,
                                                         MutView mv, long long *cap)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nItems; t += (long long)gridDim.x * blockDim.x) {
        const long long k = t / nL;
        const int f = a + (int)(t - k * nL);
        long long cp = 0;
        if (R[k * nF + f] < 0) {
            const int src = R[k * nF + frameParent[f]];
            if (src >= 0) cp = (long long)av.n_ent[src] + 2ll * mv.cnt[frameMut[f]];
        }
        cap[t] = cp;
    }
}
__global__ __launch_bounds__(MAPLE_BLOCK) void k_fan_pass(int lRef, ArenaView av, MutView mv, long long nItems, int nF, int a, int nL,
                                                          const int32_t *R, const int32_t *frameParent, const int32_t *frameMut,
                                                          const long long *cap, const long long *woff, uint2 *words, double *aux,
                                                          long long *ne, long long *na)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nItems; t += (long long)gridDim.x * blockDim.x) {
        long long e = 0, x = 0;
        if (cap[t] > 0) {
            const long long k = t / nL;
            const int f = a + (int)(t - k * nL);
            Writer w;
            w.init(words + woff[t], aux + 5 * woff[t]);
            const int id = frameMut[f];
            e = pass_walk(lRef, list_ref(av, R[k * nF + frameParent[f]]), mv.mut3 + 3 * mv.off[id], mv.cnt[id], false, w);
            x = w.na;
        }
        ne[t] = e; na[t] = x;
    }
}
// one wavefront per item: scratch -> arena, the list's row of the list table, its id into R
__global__ __launch_bounds__(MAPLE_BLOCK) void k_fan_commit(long long nItems, int nF, int a, int nL, int32_t *R, const long long *woff,
                                                            const long long *ne, const long long *na, const long long *de,
                                                            const long long *da, long long baseE, long long baseA, int32_t firstId,
                                                            const uint2 *sw, const double *sa, uint2 *words, double *aux,
                                                            int64_t *t_ent_off, int64_t *t_aux_off, int32_t *t_n_ent, int32_t *t_n_aux)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long t = wave; t < nItems; t += nwaves) {
        const long long e = ne[t], x = na[t];
        if (lane == 0) {
            t_ent_off[firstId + t] = baseE + de[t]; t_aux_off[firstId + t] = baseA + da[t];
            t_n_ent[firstId + t] = (int32_t)e; t_n_aux[firstId + t] = (int32_t)x;
            if (e > 0) { const long long k = t / nL; R[k * nF + a + (int)(t - k * nL)] = firstId + (int32_t)t; }
        }
        const uint2 *s1 = sw + woff[t];
        uint2 *d1 = words + baseE + de[t];
        for (long long i = lane; i < e; i += 64) d1[i] = s1[i];
        const double *s2 = sa + 5 * woff[t];
        double *d2 = aux + baseA + da[t];
        for (long long i = lane; i < x; i += 64) d2[i] = s2[i];
    }
}

static int fan_out_level(maple_ctx *c, int m, int nF, int a, int b, int32_t *dR, const int32_t *dFrameParent, const int32_t *dFrameMut,
                         DevBuf<long long> *buf /* [6] */, DevBuf<uint8_t> &tmp, double *bytesRead)
{
    const int nL = b - a;
    const long long nItems = (long long)m * nL;
    if (nItems <= 0) return MAPLE_OK;
    if (nItems > 0x7fffffffLL) return fail(c, MAPLE_ERR_ARG, "too many (query, frame) items in one level");
    for (int i = 0; i < 6; i++) HIPCK(c, buf[i].reserve((size_t)nItems + 1));
    long long *cap = buf[0].p, *woff = buf[1].p, *ne = buf[2].p, *na = buf[3].p, *de = buf[4].p, *da = buf[5].p;
    const int grid = (int)std::min<long long>((nItems + MAPLE_BLOCK - 1) / MAPLE_BLOCK, 256 * 8);
    hipLaunchKernelGGL(k_fan_cap, dim3(grid), dim3(MAPLE_BLOCK), 0, c->stream, nItems, nF, a, nL, dR, dFrameParent, dFrameMut, view(c),
                       mview(c), cap);
    size_t tb = 0;
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cap, woff, (int)nItems, c->stream));
    HIPCK(c, tmp.reserve(tb + 256));
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cap, woff, (int)nItems, c->stream));
    long long lastOff = 0, lastCap = 0;
    HIPCK(c, hipMemcpyAsync(&lastOff, woff + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&lastCap, cap + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const long long tot = lastOff + lastCap;
    if (tot == 0) return MAPLE_OK;
    HIPCK(c, c->s_words.reserve((size_t)tot));
    HIPCK(c, c->s_aux.reserve((size_t)(5 * tot)));
    hipLaunchKernelGGL(k_fan_pass, dim3(grid), dim3(MAPLE_BLOCK), 0, c->stream, c->lRef, view(c), mview(c), nItems, nF, a, nL, dR,
                       dFrameParent, dFrameMut, cap, woff, c->s_words.p, c->s_aux.p, ne, na);
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, ne, de, (int)nItems, c->stream));
    HIPCK(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, na, da, (int)nItems, c->stream));
    long long tail[4] = {0, 0, 0, 0};
    HIPCK(c, hipMemcpyAsync(&tail[0], de + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&tail[1], ne + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&tail[2], da + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&tail[3], na + nItems - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const long long totE = tail[0] + tail[1], totA = tail[2] + tail[3];
    const int64_t first = (int64_t)c->h_n_ent.size();
    if (c->used_ent + totE > c->cap_ent || c->used_aux + totA > c->cap_aux)
        return fail(c, MAPLE_ERR_NOMEM, "arena full while expressing %d removed lists in %d reference frames", m, nL);
    if (first + nItems > c->cap_lists) return fail(c, MAPLE_ERR_NOMEM, "list table full (%lld)", (long long)c->cap_lists);
    const int gridW = (int)std::min<long long>((nItems + MAPLE_BLOCK / 64 - 1) / (MAPLE_BLOCK / 64), 256 * 8);
    hipLaunchKernelGGL(k_fan_commit, dim3(gridW), dim3(MAPLE_BLOCK), 0, c->stream, nItems, nF, a, nL, dR, woff, ne, na, de, da,
                       (long long)c->used_ent, (long long)c->used_aux, (int32_t)first, c->s_words.p, c->s_aux.p, c->d_words, c->d_aux,
                       c->d_ent_off, c->d_aux_off, c->d_n_ent, c->d_n_aux);
    HIPCK(c, hipGetLastError());
    // the host's copy of the new rows
    c->h_ent_off.resize((size_t)(first + nItems)); c->h_aux_off.resize((size_t)(first + nItems));
    c->h_n_ent.resize((size_t)(first + nItems)); c->h_n_aux.resize((size_t)(first + nItems));
    HIPCK(c, hipMemcpyAsync(c->h_ent_off.data() + first, c->d_ent_off + first, (size_t)nItems * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(c->h_aux_off.data() + first, c->d_aux_off + first, (size_t)nItems * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(c->h_n_ent.data() + first, c->d_n_ent + first, (size_t)nItems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(c->h_n_aux.data() + first, c->d_n_aux + first, (size_t)nItems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->used_ent += totE;
    c->used_aux += totA;
    *bytesRead += 8.0 * (double)totE + 8.0 * (double)totA;
    return MAPLE_OK;
}

// The tree in the searches' own depth-first order (SScan, search_dev.h) -- clade sizes, depths and the per-node facts the
// cached-regime descent tests -- plus what rows with a bitmap of their finite scores need (FiniteRows).  Per uploaded tree and
// effectivelyNon0BLen.
static int build_scan_tables(maple_ctx *c, const SearchParams &P)
{

    // the tree in the searches' own depth-first order (SScan, search_dev.h): clade sizes, depths and the per-node facts
    // the cached-regime descent tests
    const int32_t nT = c->dtree.n;
    std::vector<int32_t> byRank(nT, -1);
    for (int i = 0; i < nT; i++) byRank[c->h_nodes[i].preRank] = i;
    std::vector<SScan> sc((size_t)nT);
    std::vector<int32_t> size(nT, 1), depth(nT, 0);
    // Node slots the root does not reach (a tree read from a file keeps the slots of collapsed nodes, their `up` still
    // naming a live node) rank behind every reachable node and belong to no clade: counted into their stale parent's
    // clade they made the scan of that parent -- and of every ancestor -- run past the clade's end.
    // (what the root reaches, and how deep: the upload's own walk of the tree has both -- PlaceMeta::order, h_depth)
    std::vector<uint8_t> reach(nT, 0);
    int32_t maxDepth = 0;
    if (!c->tree_stale && c->place && (int)c->h_depth.size() == nT && !c->place->order.empty() && c->place->order[0] == c->dtree.root) {
        for (const int32_t v : c->place->order) { reach[v] = 1; depth[v] = c->h_depth[v]; maxDepth = std::max(maxDepth, depth[v]); }
    } else {
        std::vector<int32_t> stk{c->dtree.root};
        while (!stk.empty()) {
            const int v = stk.back();
            stk.pop_back();
            reach[v] = 1;
            if (c->h_tree_c0[v] >= 0) { stk.push_back(c->h_tree_c0[v]); stk.push_back(c->h_tree_c1[v]); }
        }
        for (int r = 0; r < nT; r++) {                                  // parents precede their clades in rank order
            const int v = byRank[r];
            const int u = c->h_tree_up[v];
            if (reach[v] && u >= 0 && v != c->dtree.root && c->h_nodes[u].preRank < r) depth[v] = depth[u] + 1;
            maxDepth = std::max(maxDepth, depth[v]);
        }
    }
    for (int r = nT - 1; r >= 0; r--) {
        const int v = byRank[r];
        const int u = c->h_tree_up[v];
        if (reach[v] && u >= 0 && v != c->dtree.root && c->h_nodes[u].preRank < r) size[u] += size[v];
    }
    for (int r = 0; r < nT; r++) {
        const int v = byRank[r];
        const NodeRec &nr = c->h_nodes[v];
        uint32_t fl = 0;
        if (nr.up >= 0 && (nr.dist > P.effNon0 || nr.upIsRoot)) fl |= SS_SCORED;
        if (nr.totUp >= 0) fl |= SS_TOTUP;
        if (nr.c0 >= 0) fl |= SS_INNER;
        if (nr.up >= 0 && (nr.whichChild ? c->h_nodes[nr.up].upLeft : c->h_nodes[nr.up].upRight) >= 0) fl |= SS_ENTER;
        sc[r] = SScan{v, size[v], depth[v], ((uint32_t)nr.frameOf << 4) | fl};
    }
    std::vector<int32_t> prank((size_t)nT, 0);
    for (int r = 0; r < nT; r++) { const int u = c->h_tree_up[byRank[r]]; prank[r] = u >= 0 ? c->h_nodes[u].preRank : 0; }
    // for rows that come with a bitmap of their finite scores (FiniteRows, search_dev.h): candidates before each rank, and
    // what a clade adds to the count of candidate placements when it is walked with every score -inf
    std::vector<int32_t> candBefore((size_t)nT + 1, 0), cladeVisits((size_t)nT, 0);
    for (int r = 0; r < nT; r++)
        candBefore[r + 1] = candBefore[r] + ((sc[r].ff & SS_TOTUP) ? 1 : 0);   // (the order of the dense kernel's candidates)
    for (int r = nT - 1; r >= 1; r--) {
        const int v = byRank[r];
        if (!reach[v]) continue;
        const uint32_t fl = sc[r].ff & 15u;
        if (!(fl & SS_ENTER)) continue;                                 // never pushed: neither it nor its clade is visited
        const bool scored = fl & SS_SCORED, counts = scored && (fl & SS_TOTUP), dropped = scored && !(fl & SS_TOTUP);
        const int add = (counts ? 1 : 0) + ((!dropped && (fl & SS_INNER)) ? cladeVisits[r] : 0);
        cladeVisits[prank[r]] += add;
    }
    TRY(h2d(c, c->t_cand_before, candBefore.data(), candBefore.size()));
    TRY(h2d(c, c->t_clade_visits, cladeVisits.data(), cladeVisits.size()));
    TRY(h2d(c, c->t_scan, sc.data(), sc.size()));
    TRY(h2d(c, c->t_scan_parent, prank.data(), prank.size()));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->tree_max_depth = maxDepth;
    c->scan_eff = P.effNon0;
    c->scan_valid = true;
    return MAPLE_OK;
}

// finite scores before each word of every row's bitmap (FiniteRows, search_dev.h): one wavefront per row
__global__ __launch_bounds__(64) void k_finite_prefix(int nRows, int nWords, const unsigned long long *mask, int32_t *prefix)
{
    const int lane = threadIdx.x;
    for (int row = blockIdx.x; row < nRows; row += gridDim.x) {
        const unsigned long long *m = mask + (size_t)row * nWords;
        int32_t *p = prefix + (size_t)row * (nWords + 1);
        int run = 0;
        for (int base = 0; base <= nWords; base += 64) {
            const int w = base + lane;
            const int cnt = w < nWords ? __popcll(m[w]) : 0;
            int incl = cnt;
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
            if (w <= nWords) p[w] = run + incl - cnt;
            run += __shfl(incl, 63, 64);
        }
    }
}

// Trees with MAT local references: lists up the chain of their frames (passGenomeListThroughBranch, one enclosing frame per
// round, batched) until every one is written in the ROOT's frame.  ids / fr: list ids and their frames, both updated in place.
static int lists_to_root_frame(maple_ctx *c, std::vector<int32_t> &ids, std::vector<int32_t> fr)
{
    const PlaceMeta &Fm = *c->place;
    std::vector<int32_t> who, src, ml, out;
    std::vector<uint8_t> dir;
    for (;;) {
        who.clear(); src.clear(); ml.clear();
        for (size_t k = 0; k < ids.size(); k++)
            if (fr[k] != 0 && ids[k] >= 0) { who.push_back((int32_t)k); src.push_back(ids[k]); ml.push_back(c->h_tree_mut[Fm.frameNode[fr[k]]]); }
        if (who.empty()) return MAPLE_OK;
        dir.assign(who.size(), 1);
        out.resize(who.size());
        TRY(maple_pass_branch_batch(c, (int32_t)who.size(), src.data(), ml.data(), dir.data(), out.data()));
        for (size_t i = 0; i < who.size(); i++) { ids[who[i]] = out[i]; fr[who[i]] = Fm.frameParent[fr[who[i]]]; }
    }
}
// ... the candidates' copies: made once per tree, kept in the arena until the tree changes or a release of the caller's takes them
static int ensure_cand_root(maple_ctx *c)
{
    if (c->cand_root_end >= 0) return MAPLE_OK;
    // (the copies of a tree that has been patched or uploaded again since: n_scored lists nobody reads any more.  They go when they are
    // still the last lists of the arena -- the usual case in a loop of searches and patches; lists the caller appended behind them pin
    // them until the caller's own release)
    // (only the genome lists go: the release mark carries the CURRENT count of mutation lists, so that MAT mutation lists the caller
    // uploaded -- or released -- since the copies were made are left as they are)
    if (c->cand_root_mark >= 0 && c->cand_root_top == (int64_t)c->h_n_ent.size())
        TRY(maple_arena_release(c, c->cand_root_mark | ((int64_t)c->h_mut_cnt.size() << 40)));
    c->cand_root_mark = c->cand_root_top = -1;
    const int64_t begin = (int64_t)c->h_n_ent.size();                  // (the list count alone, not an arena mark of both kinds)
    std::vector<int32_t> candRoot(c->h_cand_ids);
    TRY(lists_to_root_frame(c, candRoot, c->h_cand_frame));
    TRY(h2d(c, c->s_cand_root, candRoot.data(), candRoot.size()));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->cand_root_end = (int64_t)c->h_n_ent.size();
    c->cand_root_mark = begin; c->cand_root_top = c->cand_root_end;
    return MAPLE_OK;
}

extern "C" int maple_spr_search_batch(maple_ctx *c, int32_t n, const int32_t *nodes, const maple_search_params *sp,
                                      int32_t ws_entries_per_lane, int32_t *bestNode, double *bestScore, double *blen3,
                                      int32_t *placement, double *improvement, double *currentLK, int32_t *nAppend,
                                      int32_t *status, int32_t *outRprList)
{
    if (!c || n < 0 || !nodes || !sp || !bestNode || !bestScore || !blen3 || !placement || !improvement || !currentLK
        || !nAppend || !status)
        return MAPLE_ERR_ARG;
    if (n == 0) return MAPLE_OK;
    HIPCK(c, hipSetDevice(c->device));
    TRY(need_model(c));
    if (!c->tree_set) return fail(c, MAPLE_ERR_STATE, "maple_tree_upload has not been called");
    for (int i = 0; i < n; i++)
        if (nodes[i] < 0 || nodes[i] >= c->dtree.n) return fail(c, MAPLE_ERR_ARG, "nodes[%d] = %d is not a node", i, nodes[i]);
    // The tree was patched since its tables were built.  A small batch (the re-search of proposed moves before they are
    // applied) runs on the patched node records alone, through the frontier tier with no hand-over to the dense tier -- unless
    // one of its searches is a whole-tree search by construction (a zero-length branch without an error model); everything
    // else rebuilds the tables first.
    bool patchedOnly = false;
    c->last_search_frontier_only = false;
    if (c->tree_stale) {
        patchedOnly = c->nodes_current && !c->tree_has_mut && n <= 64 && sp->searchTier == 0 && c->trace_query < 0 && !outRprList;
        for (int i = 0; i < n && patchedOnly; i++)
            if (sp->wideSearchBudget >= 0 && !c->dm.usingErrorRate && c->h_tree_dist[nodes[i]] == 0.0) patchedOnly = false;
        if (!patchedOnly) TRY(tree_rebuild_from_host(c));
    }
    const bool dbgT = c->tuning.verbose != 0;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration_cast<std::chrono::microseconds>(b - a).count() * 1e-3;
    };
    const auto tStart = tnow();
    SearchParams P;
    P.strict = sp->strictTopologyStopRules; P.allowedFails = sp->allowedFailsTopology;
    P.thrLKtopology = sp->thresholdLogLKtopology; P.thrPlacement = sp->thresholdTopologyPlacement;
    P.thrOptTopo = sp->thresholdLogLKoptimizationTopology; P.thrConsec = sp->thresholdLogLKconsecutivePlacement;
    P.effNon0 = sp->effectivelyNon0BLen;
    HIPCK(c, c->s_search_out.reserve((size_t)n * sizeof(SearchOut)));
    HIPCK(c, c->s_counter.reserve(8));
    std::vector<SearchOut> &ho = c->h_search_out;
    ho.assign((size_t)n, SearchOut{});
    std::vector<uint8_t> hintedNow;                                    // (searches sent to the dense tier on the strength of h_over_hint)
    std::vector<int32_t> todo(nodes, nodes + n), slot(n);
    for (int i = 0; i < n; i++) slot[i] = i;
    // output pool for bestRemovedPartials: a list re-expressed in another frame stays close to its original size
    long long poolCapW = 0, poolCapA = 0;
    uint2 *poolW = nullptr;
    double *poolA = nullptr;
    if (outRprList) {
        for (int i = 0; i < n; i++) {
            int lid = c->h_tree_lower[nodes[i]];
            long long ne = lid >= 0 ? c->h_n_ent[lid] : 0, na = lid >= 0 ? c->h_n_aux[lid] : 0;
            poolCapW += 3 * ne + 64; poolCapA += 3 * na + 5 * ne + 64;
        }
        // its own buffers: the batch operators' scratch is reused by the per-frame passes of the wide searches
        HIPCK(c, c->s_pool_w.reserve((size_t)poolCapW));
        HIPCK(c, c->s_pool_a.reserve((size_t)poolCapA));
        poolW = c->s_pool_w.p; poolA = c->s_pool_a.p;
    }
    HIPCK(c, hipMemsetAsync(c->s_counter.p, 0, 8 * sizeof(int32_t), c->stream));
    unsigned long long *poolUsed = (unsigned long long *)(c->s_counter.p + 2);
    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: result records cleared\n", tms(tStart, tnow()));
    // per-lane list workspace: a search near the root of a tree with long lists (rate variation: many O vectors) merges
    // lists of several hundred entries a few hundred times before it is handed over or done
    const int capW0 = ws_entries_per_lane > 0 ? ws_entries_per_lane : std::max(16384, 64 * c->tree_max_ent);
    // Runs the searches `todo` (results into ho[slot[]]).  Queries whose per-lane workspace overflowed (status -3) are
    // re-run with 8x the workspace, twice at most.  cacheS (optional) = row-major (|todo| x T.n) cached scores.
    bool heavyQueries = false;
    // Lane searches assisted by their wavefront (k_spr_search, wave_dev.h): without an
    // error model -- with one, every search from a zero-length branch also runs to the budget here (no routing hint), the lane
    // tier is then bound by its throughput, not by its longest search, and 24 lanes walking in lockstep do better (100 000
    // tips, full model: 460 ms against 541).
    bool assistOK = !c->dm.usingErrorRate;
    const bool assistFew = !c->dm.usingErrorRate;      // few searching lanes per wavefront, every request served by all 64 lanes
    // (the frontier tier, frontier.hip / frontier_upd.hip; searchTier 1 keeps every search in the one-lane kernels)
    const bool useFrontier = sp->searchTier == 0 && c->trace_query < 0;
    const std::vector<int32_t> *rowOverride = nullptr;                 // rows of the score table the next cached launch reads
    FiniteRows finRows{nullptr, nullptr, 0};                           // ... and, where the rows come with one, the bitmap of their finite scores
    const int finWords = (c->n_scored + 63) / 64;                      // (words per row: one per tile of 64 candidates of the dense kernel)
    auto fin_reserve = [&](size_t rows) -> int {
        HIPCK(c, c->s_fin_mask.reserve_exact(std::max(rows * (size_t)finWords, c->s_fin_mask.cap)));
        HIPCK(c, c->s_fin_prefix.reserve_exact(std::max(rows * (size_t)(finWords + 1), c->s_fin_prefix.cap)));
        return MAPLE_OK;
    };
    auto fin_prefix = [&](hipStream_t st, size_t row0, size_t rows) -> int {   // (after the dense launch that wrote those rows' bitmaps)
        if (!rows) return MAPLE_OK;
        k_finite_prefix<<<(int)std::min<size_t>(rows, 4096), 64, 0, st>>>((int)rows, finWords, c->s_fin_mask.p + row0 * finWords,
                                                                          c->s_fin_prefix.p + row0 * (finWords + 1));
        HIPCK(c, hipGetLastError());
        return MAPLE_OK;
    };
    std::function<int()> afterLaunch;                                  // called once, right after the next search kernel is queued
    auto run_queries = [&](std::vector<int32_t> todo, std::vector<int32_t> slot, const double *cacheS, int budgetNow,
                           const int32_t *rTable, int nF) -> int {
        // the few cached (whole-tree) searches get room up front; more when the budgeted pass already ran out of it
        int capW = cacheS ? (heavyQueries ? 8 : 4) * capW0 : capW0;
        std::vector<int32_t> rows(todo.size());                        // row of each query in the cache / frame tables
        for (size_t k = 0; k < rows.size(); k++) rows[k] = rowOverride ? (*rowOverride)[k] : (int32_t)k;
        if ((int)c->h_depth.size() >= c->dtree.n) {
            // Lanes pull searches from a counter, so a launch ends one search after the last one is pulled: the expensive
            // searches go first.  The expensive ones are those near the root (long lists: an updating step there merges
            // several hundred entries; measured up to 100 ms of updating steps in one search of the 100 000-tip tree
            // against 4 ms on average) -- nodes in order of depth.
            std::vector<int32_t> ord(todo.size());
            for (size_t k = 0; k < ord.size(); k++) ord[k] = (int32_t)k;
            if (cacheS) {
                // ... for whole-tree searches, those that take a large clade out of the tree: what its removal changes reaches far,
                // the search updates lists for hundreds of steps (measured: the 11 searches of the 100 000-tip tree that used to
                // come back for more workspace sit at depths 17-22, with removed lists of ordinary length)
                if ((int)c->h_clade.size() != c->dtree.n) {
                    const int nT = c->dtree.n;
                    c->h_clade.assign((size_t)nT, 1);
                    std::vector<int32_t> order, stk{c->dtree.root};
                    while (!stk.empty()) {
                        const int v = stk.back();
                        stk.pop_back();
                        order.push_back(v);
                        if (c->h_tree_c0[v] >= 0) { stk.push_back(c->h_tree_c0[v]); stk.push_back(c->h_tree_c1[v]); }
                    }
                    for (size_t k = order.size(); k-- > 1;) c->h_clade[c->h_tree_up[order[k]]] += c->h_clade[order[k]];
                }
                std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return c->h_clade[todo[a]] > c->h_clade[todo[b]]; });
            } else
            std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return c->h_depth[todo[a]] < c->h_depth[todo[b]]; });
            std::vector<int32_t> t2(todo.size()), s2(todo.size()), r2(todo.size());
            for (size_t k = 0; k < ord.size(); k++) { t2[k] = todo[ord[k]]; s2[k] = slot[ord[k]]; r2[k] = rows[ord[k]]; }
            todo.swap(t2); slot.swap(s2); rows.swap(r2);
        }
        for (int attempt = 0; attempt < 3 && !todo.empty(); attempt++, capW *= 8) {
            const int m = (int)todo.size();
            WsLayout L;
            L.capW = capW;
            L.capA = 5 * L.capW;                                        // O-vector-heavy lists (rate variation) carry up to 4-5 aux doubles per entry
            L.capH = L.capW / 8 + 256;
            L.capS = 1024 * (attempt + 1);
            L.capB = (cacheS ? 4096 : 1024) * (attempt + 1);             // whole-tree (cached) searches short-list far more branches
            L.capAis = 8192 * (attempt + 1);
            LaneBytes LB = lane_bytes(L);
            // lanes: one query per lane while they last; at most 4 wavefronts per SIMD (the kernel's occupancy) and a
            // workspace footprint bounded to ~96 GB of the 288 GB
            long long wsBudget = 96ll << 30;
            {   // ... and to 70 % of what is free on the device right now (plus what this buffer already holds)
                size_t freeB = 0, totalB = 0;
                if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                    const long long avail = (long long)((double)(freeB + c->s_search_ws.cap) * 0.7);
                    if (avail < wsBudget) wsBudget = avail;
                }
            }
            long long maxLanes = wsBudget / (long long)LB.total;
            if (maxLanes > 4096 * 64) maxLanes = 4096 * 64;
            // lanes pull searches from a counter: more wavefronts than the GPU holds at once (4 per SIMD) only cost workspace
            if (cacheS && maxLanes > 8192) maxLanes = 8192;
            if (maxLanes < 64) maxLanes = 64;
            const int lanesWanted = (int)(m < maxLanes ? m : maxLanes);
            // searching lanes per wavefront (measured at 20k queries: the long searches of a non-strict round like 2 lanes,
            // 42 vs 46 ms with 1; the short ones of a strict round like 1, 31 vs 37 ms with 2; more is always worse)
            // (at 200k queries with the budget of a 100 000-tip tree: 2 / 4 / 8 / 13 / 24 lanes -> 825 / 645 / 512 / 424 / 417 ms)
            const int lanesDiv = P.strict ? 32768 : (lanesWanted > 65536 ? 8192 : 16384);
            int activeLanes = (lanesWanted + lanesDiv - 1) / lanesDiv;
            if (activeLanes < 1) activeLanes = 1;
            if (activeLanes > 64) activeLanes = 64;
            // wave-assisted lane searches (below): the wavefront serves its lanes' score requests one after the other, so few
            // searching lanes per wavefront (100 000 tips, budget 2 132: 2 / 4 / 6 / 8 / 16 / 24 lanes -> 298 / 281 / 279 / 289 /
            // 394 / 345 ms; without the assistance 375)
            if (!cacheS && assistOK && assistFew && activeLanes > 4) activeLanes = 4;
            int nWaves = (lanesWanted + activeLanes - 1) / activeLanes;
            if (nWaves > 8192) nWaves = 8192;
            const int lanes = nWaves * activeLanes;
            const auto tWs0 = tnow();
            {   // grow-only and at least doubling (a 20 GB hipMalloc costs ~0.7 s): batches of slowly growing size must not
                // reallocate every time
                size_t need = (size_t)lanes * LB.total;
                if (need > c->s_search_ws.cap) {
                    const size_t top = (size_t)maxLanes * LB.total;
                    need = std::min(std::max(need, 2 * c->s_search_ws.cap), std::max(top, need));
                }
                HIPCK(c, c->s_search_ws.reserve_exact(need));
            }
            if (dbgT) fprintf(stderr, "[maple]   workspace %d lanes x %zu B: reserve %.1f ms\n", lanes, (size_t)LB.total, tms(tWs0, tnow()));
            HIPCK(c, hipMemsetAsync(c->s_counter.p, 0, sizeof(int32_t), c->stream));
            TRY(h2d(c, c->s_i32[0], todo.data(), (size_t)m));
            if (cacheS) TRY(h2d(c, c->s_i32[1], rows.data(), (size_t)m));
            HIPCK(c, c->s_search_out.reserve((size_t)m * sizeof(SearchOut)));
            SearchOut *dout = (SearchOut *)c->s_search_out.p;
            // cached (whole-tree) searches descend by scanning the tree in their own depth-first order; the per-depth slots
            // of every searching lane live in LDS (deeper trees fall back to popping one node at a time)
            DevTree Tk = c->dtree;
            Tk.candBefore = c->t_cand_before.p;
            Tk.cladeVisits = c->t_clade_visits.p;
            size_t dynLds = 0;
            int launchLanes = activeLanes, launchWaves = nWaves;
            if (cacheS && c->scan_valid && !c->tuning.noCladeScan && (size_t)(c->tree_max_depth + 2) * 16 <= (48u << 10)) {
                Tk.scan = c->t_scan.p;
                Tk.scanParent = c->t_scan_parent.p;

                Tk.scanDepthCap = c->tree_max_depth + 2;
                dynLds = ((size_t)Tk.scanDepthCap * 16 + 15) & ~(size_t)15;
                launchLanes = 1;                                         // one search per wavefront, 64 lanes per clade scan
                launchWaves = lanes;                                     // (the workspace is sized for `lanes` searches at a time)
            } else {
                Tk.scan = nullptr; Tk.scanParent = nullptr; Tk.scanDepthCap = 0;
                if (!cacheS && assistOK) dynLds = sizeof(WaveLds);
            }
            // searches that update lists for hundreds of steps outgrow the per-lane list room; they carry on in chunks (one lane's
            // worth each) of a pool the launch shares instead of coming back for a second launch with 8x the room
            long long ovfChunks = 0;
            if (cacheS && attempt == 0) {
                ovfChunks = std::min<long long>(1024, std::max<long long>(64, m / 16));
                HIPCK(c, c->s_search_ws_big.reserve_exact((size_t)ovfChunks * ((size_t)L.capW * sizeof(uint2) + (size_t)L.capA * sizeof(double))));
                HIPCK(c, hipMemsetAsync(c->s_counter.p + 6, 0, 2 * sizeof(int32_t), c->stream));
            }
            const int coopMaxHost = 8;                                      // (see k_spr_search: requests served one by one)
            hipEvent_t e0, e1;
            TRY(ev_pair(c, &e0, &e1, cacheS ? MAPLE_K_SPR_REPLAY : MAPLE_K_SPR_SEARCH, (double)m, 0.0));
            const size_t slotEv = c->ev_used / 2 - 1;                       // (this launch's timing record: filled in below)
            HIPCK(c, hipEventRecord(e0, c->stream));
#define MAPLE_SPR_LAUNCH_ARGS <<<launchWaves, 64, dynLds, c->stream>>>(c->d_model, view(c), mview(c), Tk, P, m, c->s_i32[0].p,          \
                                                                         L, LB, c->s_search_ws.p, c->s_counter.p, dout, poolW,     \
                                                                         poolA, poolUsed, poolCapW, poolCapA,                      \
                                                                         attempt == 0 ? c->trace_query : -1, c->s_trace_i.p,       \
                                                                         c->s_trace_d.p, 4096, c->s_trace_i.p ? c->s_trace_i.p + 4 * 4096 : nullptr, \
                                                                         launchLanes, cacheS, budgetNow, rTable, nF,               \
                                                                         cacheS ? c->s_i32[1].p : nullptr,                          \
                                                                         assistOK ? 1 + coopMaxHost : 0, (unsigned long long *)(c->s_counter.p + 6),          \
                                                                         ovfChunks ? c->s_search_ws_big.p : nullptr, ovfChunks, cacheS ? finRows : FiniteRows{nullptr, nullptr, 0})
            if (!cacheS && assistOK) DISPATCH3(c, k_spr_search_assisted, MAPLE_SPR_LAUNCH_ARGS);
            else DISPATCH3(c, k_spr_search, MAPLE_SPR_LAUNCH_ARGS);
#undef MAPLE_SPR_LAUNCH_ARGS
            HIPCK(c, hipGetLastError());
            HIPCK(c, hipEventRecord(e1, c->stream));
            if (afterLaunch) {                                             // (work for the side stream, queued behind this launch)
                std::function<int()> f;
                f.swap(afterLaunch);
                TRY(f());
            }
            std::vector<SearchOut> part(m);
            HIPCK(c, hipMemcpyAsync(part.data(), dout, (size_t)m * sizeof(SearchOut), hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
            std::vector<int32_t> todo2, slot2, rows2;
            {   // what this launch did, for maple_timing_read_kind: candidate placements it scored itself (lane searches that
                // finished: each reads a candidate list and writes a score, SURVEY 8d, plus its removed list once) or replayed
                // from the score table (8 bytes each)
                const double meanCand = c->n_scored ? c->scored_bytes_total / c->n_scored : 0.0;
                double units = 0.0, bytes = 0.0;
                for (int k = 0; k < m; k++) {
                    if (part[k].status != 0 && part[k].status != -1) continue;
                    units += part[k].nAppend;
                    const int32_t l = c->h_tree_lower[todo[k]];
                    const double qb = l >= 0 ? 8.0 * c->h_n_ent[l] + 8.0 * c->h_n_aux[l] : 0.0;
                    bytes += cacheS ? 8.0 * part[k].nAppend + qb : meanCand * part[k].nAppend + qb;
                }
                c->ev_units[slotEv] = units; c->ev_bytes[slotEv] = bytes;
            }
            for (int k = 0; k < m; k++) {
                ho[slot[k]] = part[k];
                if (part[k].status == -3 && attempt < 2) {
                    todo2.push_back(todo[k]); slot2.push_back(slot[k]); rows2.push_back(rows[k]);
                    if (dbgT) fprintf(stderr, "[maple]   node %d ran out of workspace (capacity kind %d): position %d of the launch, depth %d, lower list %d entries\n", todo[k], part[k].nAppend, k, c->h_depth[todo[k]], c->h_tree_lower[todo[k]] >= 0 ? c->h_n_ent[c->h_tree_lower[todo[k]]] : -1);
                }
            }
            if (dbgT) fprintf(stderr, "[maple] search launch: %d queries, %zu retried with more workspace\n", m, todo2.size());
            todo.swap(todo2);
            slot.swap(slot2);
            rows.swap(rows2);
            // (the budget stays: a retried search that turns out to be wide still goes to the batch path)
        }
        return MAPLE_OK;
    };
    // Wide searches (the non-strict rounds let ~1 query in 5 walk the whole tree): in the cached regime the score of a
    // branch is a pure function of (query, branch), so those queries are scored against every branch by the batch
    // kernel (k_append_queries) and the state machine then only replays the traversal over the cached scores.
    // With MAT local references the removed list is first expressed in every reference frame (below).
    // A search that scores more branches than the budget is handed to the dense path, which costs it one appendProbNode per
    // branch of the tree -- so the budget that pays grows with the tree: 1/64 of the scored branches, measured best at 10 000
    // tips (256: 91 ms per round; 128: 100, 384: 97) and at 100 000 (2 048: 1.08 s; 256: 2.69, 1 024: 1.13, 4 096: 1.20)
    int wideBudget = sp->wideSearchBudget == 0 ? std::max(MAPLE_WIDE_BUDGET_DEFAULT, std::min(8192, c->n_scored / 64))
                                               : sp->wideSearchBudget;
    if (patchedOnly) wideBudget = -1;                                   // (no tree-sized table is current)
    // (a long search costs the wave-assisted lane tier a tenth of what it cost one lane: twice the budget pays -- 100 000 tips:
    // 2 132 / 3 072 / 4 096 / 6 144 -> 693 / 688 / 668 / 692 ms per round; 10 000 tips: 256 / 384 / 512 -> 75 / 72 / 72)
    // (with an error model too, since the frontier tier: 100 000 tips, budget 2 132 / 3 000 / 4 264 / 8 528 -> 532 / 447 / 420 / 441 ms
    // per round -- the searches between 2 000 and 4 000 items are the ones with the longest removed lists, which the dense kernel
    // walks slowest: 28 116 rows take it 260 ms, 25 785 rows 114)
    // (1 000 000 tips, 8 192 / 16 384: 2.32 / 2.73 s per 131 072 searches -- the pools of the longer searches are reallocated on
    // the way: with an error model the doubling stops at 8 192)
    if (sp->wideSearchBudget == 0) wideBudget = c->dm.usingErrorRate ? std::min(2 * wideBudget, std::max(wideBudget, 8192)) : 2 * wideBudget;
    const bool hybrid = wideBudget > 0;
    if (c->over_hint_budget != wideBudget || c->over_hint_eff0 != sp->effectivelyNon0BLen) {   // (hints taken under another budget or
        c->h_over_hint.clear();                                                                // threshold say nothing about this one)
        c->over_hint_budget = wideBudget; c->over_hint_eff0 = sp->effectivelyNon0BLen;
    }
    if (hybrid && !(c->scan_valid && c->scan_eff == P.effNon0) && !c->tuning.noCladeScan) {
        TRY(build_scan_tables(c, P));
        if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: scan tables of the tree built (first search since it changed)\n", tms(tStart, tnow()));
    }
    // rows of the score table come with the bitmap of their finite scores (FiniteRows) when the tables that go with it exist
    bool useFin = hybrid && c->scan_valid && !c->tree_has_mut;         // (trees with local references: only for the rows of the searches
                                                                       // known beforehand, below)
    // Without an error model the whole-tree searches are known before anything runs: they are the ones that start from a
    // zero-length branch (the routing hint in the kernel gives those 16 placements and sends them on).  Their dense scoring
    // needs nothing from the lane searches, so it is launched first, on a side stream, and shares the GPU with them -- the
    // lane launch is latency-bound and spends its second half on a thinning tail.  Rows of the score table: the predicted
    // searches in order, then up to `preSpare` searches that run over their budget unannounced.
    std::vector<int32_t> preIdx, preRowOf;
    int preSpare = 0;
    const int nTpre = c->dtree.n;
    // Trees with MAT local references: appendProbNode does not depend on the frame its two lists are written in (the same sites
    // need work, with the same nucleotides, lengths and rates, in the same order), so the rows of these searches are made in the
    // ROOT's frame: every candidate list and every removed list re-expressed there once per call (passGenomeListThroughBranch up
    // the chain of frames), the witness filter and the pair walks as on a plain tree.
    const bool matPre = c->tree_has_mut;
    int64_t preMark = -1;                                              // (the re-expressed lists live until the results are in)
    std::vector<int32_t> preFrameParent, preFrameNode;
    auto pre_rows_max = [&]() -> size_t {                             // rows the score table may have: half of what is free, 96 GiB at most
        size_t freeB = 0, totalB = 0;
        size_t budgetB = (size_t)4ull << 30;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess)
            budgetB = std::max(budgetB, std::min((freeB + c->s_cache.cap * sizeof(double)) / 2, (size_t)96ull << 30));
        return budgetB / ((size_t)nTpre * sizeof(double));
    };
    // the searches `preIdx` get their rows of the score table on the side stream, next to the pass that follows, and are whole-tree
    // searches inside the tier (witnessOK: removedBLen = 0 and no error model -- the witness filter instead of the dense kernel)
    auto pre_rows = [&](const bool witnessOK) -> int {
        const size_t rowsMax = pre_rows_max();
        if (preIdx.size() < 64 || preIdx.size() > rowsMax) preIdx.clear();
        else {
            preSpare = witnessOK ? (int)std::min<size_t>(4096, rowsMax - preIdx.size()) : 0;
            const int mZ = (int)preIdx.size();
            HIPCK(c, c->s_cache.reserve_exact((size_t)(mZ + preSpare) * nTpre));
            TRY(fin_reserve((size_t)mZ + preSpare));
            if (!c->stream2) {
                HIPCK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                HIPCK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIPCK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            std::vector<int32_t> ql(mZ);
            std::vector<uint8_t> qt(mZ);
            std::vector<double> qb(mZ);
            preRowOf.assign(n, -1);
            double qBytes = 0.0;
            for (int k = 0; k < mZ; k++) {
                const int node = nodes[preIdx[k]];
                ql[k] = c->h_tree_lower[node]; qt[k] = c->h_tree_tip[node]; qb[k] = c->h_tree_dist[node];
                preRowOf[preIdx[k]] = k;
                qBytes += 8.0 * c->h_n_ent[ql[k]] + 8.0 * c->h_n_aux[ql[k]];
                // (the witness filter writes -inf by omission ON THE PROOF that the pair is searched with removedBLen = 0 and no
                // error model -- witness.hip: a row with another length must never get here)
                if (witnessOK && qb[k] != 0.0) return fail(c, MAPLE_ERR_FATAL, "a search with removedBLen %g among the searches of the witness filter", qb[k]);
            }
            if (matPre) {
                const PlaceMeta &Fm = *c->place;
                // (the candidates' copies are made once per tree; the removed lists' copies live for this call)
                TRY(ensure_cand_root(c));
                TRY(maple_arena_mark(c, &preMark));
                std::vector<int32_t> qf(mZ);
                for (int k = 0; k < mZ; k++) qf[k] = Fm.frameOf[nodes[preIdx[k]]];
                TRY(lists_to_root_frame(c, ql, qf));
                if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: their removed lists in the root's frame\n", tms(tStart, tnow()));
                preFrameParent = Fm.frameParent;
                preFrameNode = Fm.frameNode;
                useFin = true;
            }
            HIPCK(c, c->z_ql.reserve(mZ)); HIPCK(c, c->z_qt.reserve(mZ)); HIPCK(c, c->z_qb.reserve(mZ));
            HIPCK(c, hipEventRecord(c->ev_fork, c->stream));               // (everything the tree tables wait for)
            HIPCK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            HIPCK(c, hipMemcpyAsync(c->z_ql.p, ql.data(), (size_t)mZ * sizeof(int32_t), hipMemcpyHostToDevice, c->stream2));
            HIPCK(c, hipMemcpyAsync(c->z_qt.p, qt.data(), (size_t)mZ, hipMemcpyHostToDevice, c->stream2));
            HIPCK(c, hipMemcpyAsync(c->z_qb.p, qb.data(), (size_t)mZ * sizeof(double), hipMemcpyHostToDevice, c->stream2));
            HIPCK(c, hipStreamSynchronize(c->stream2));                     // (the three vectors are locals)
            // queued BEHIND the lane launch: a workgroup of the dense kernel wants most of a compute unit's LDS, so it starts
            // where the lane searches have thinned out -- launched first it would hold them off instead (measured: no overlap)
            afterLaunch = [c, mZ, nTpre, qBytes, fin_prefix, useFin, finWords, dbgT, matPre, witnessOK]() -> int {
                // (every one of these searches has removedBLen = 0 and there is no error model: only the pairs the witness
                // filter cannot rule out are walked -- witness.hip)
                if (witnessOK && useFin && !c->tuning.denseWideScoring) {
                    long long pairs = 0;
                    TRY(witness_score(c, c->stream2, mZ, c->z_ql.p, c->z_qt.p, c->z_qb.p, c->n_scored, matPre ? c->s_cand_root.p : c->t_i32[8].p,
                                      matPre ? c->t_cand_rank.p : c->t_scored_col.p,
                                      c->s_cache.p, nTpre, c->s_fin_mask.p, finWords,
                                      c->n_scored ? c->scored_bytes_total / c->n_scored : 0.0, qBytes, &pairs));
                    if (dbgT) fprintf(stderr, "[maple] witness filter: %lld of %lld (search, branch) pairs walked\n", pairs, (long long)mZ * c->n_scored);
                    TRY(fin_prefix(c->stream2, 0, (size_t)mZ));
                    HIPCK(c, hipEventRecord(c->ev_join, c->stream2));
                    return MAPLE_OK;
                }
                // (a tree with local references: candidates and removed lists both in the root's frame, as for the witness filter)
                TRY(launch_append_queries(c, c->stream2, mZ, c->z_ql.p, c->n_scored, matPre ? c->s_cand_root.p : c->t_i32[8].p, 0, 0.0, c->s_cache.p, nTpre,
                                          matPre ? c->t_cand_rank.p : c->t_scored_col.p, c->z_qt.p, c->z_qb.p, MAPLE_K_SPR_SCORE,
                                          (double)mZ * c->scored_bytes_total + qBytes, nullptr, nullptr, nullptr, 0, 1, useFin ? c->s_fin_mask.p : nullptr));
                if (useFin) TRY(fin_prefix(c->stream2, 0, (size_t)mZ));
                HIPCK(c, hipEventRecord(c->ev_join, c->stream2));
                return MAPLE_OK;
            };
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %d searches %s to be scored on the side stream\n", tms(tStart, tnow()), mZ,
                              witnessOK ? "from zero-length branches" : "that ran over the budget the last time");
        }
        return MAPLE_OK;
    };
    if (hybrid && !c->dm.usingErrorRate && wideBudget > 16 && (!matPre || (c->scan_valid && c->place && !c->tuning.noCladeScan && !c->tuning.denseWideScoring && !c->tuning.wideOutsideFrontier))) {
        {   // a node on a zero-length branch is searched at all only if its current placement is bad enough (M:9674): the
            // kernel's own test, on the same appendProbNode, for all of them at once
            std::vector<int32_t> zi, pl, cl;
            std::vector<uint8_t> tp;
            for (int i = 0; i < n; i++) {
                const int v = nodes[i], u = c->h_tree_up[v];
                if (c->h_tree_dist[v] != 0.0 || u < 0) continue;
                if (matPre && c->h_tree_mut[v] >= 0) continue;          // (a reference node itself: the frame-by-frame path, below)
                const int32_t vu = c->h_tree_c0[u] == v ? c->h_tree_upRight[u] : c->h_tree_upLeft[u];
                if (vu < 0 || c->h_tree_lower[v] < 0) continue;
                zi.push_back(i); pl.push_back(vu); cl.push_back(c->h_tree_lower[v]); tp.push_back(c->h_tree_tip[v]);
            }
            if (zi.size() >= 64) {
                std::vector<double> bl(zi.size(), 0.0), cur(zi.size());
                const int rc = maple_append_batch(c, (int32_t)zi.size(), pl.data(), cl.data(), tp.data(), bl.data(), cur.data());
                if (rc != MAPLE_OK) return rc;
                for (size_t k = 0; k < zi.size(); k++) if (cur[k] < P.thrPlacement) preIdx.push_back(zi[k]);
            }
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: current placements of %zu nodes on zero-length branches scored\n", tms(tStart, tnow()), zi.size());
        }
        TRY(pre_rows(true));
    } else if (hybrid && c->dm.usingErrorRate && useFrontier && !c->tuning.noOverHint && (int)c->h_over_hint.size() >= c->dtree.n && wideBudget > 16
               && c->scan_valid && c->place && !c->tuning.noCladeScan && !c->tuning.wideOutsideFrontier) {
        // With an error model nothing about a node says that its search will be long -- except that it was, the last time the node was
        // searched on this tree (h_over_hint, below).  Those searches get their rows now, on the side stream next to the budgeted
        // pass, and stay in the tier as whole-tree searches like the zero-length ones above: no budget's worth of items expanded for
        // nothing, no second pass over the tier for them.  (As many as the table takes; the rest leave the pass at once.)
        const size_t rowsMax = pre_rows_max();
        for (int i = 0; i < n && preIdx.size() < rowsMax; i++)
            if (c->h_over_hint[nodes[i]] && !(matPre && c->h_tree_mut[nodes[i]] >= 0)) preIdx.push_back(i);   // (a reference node itself: as above)
        TRY(pre_rows(false));
    }
    // Frontier tier (frontier.hip): every search of the batch expanded level by level, one lane per (search, branch) item,
    // then replayed exactly -- for trees without MAT local references.  What it hands back (a search that would edit its
    // removed list in place, touches the root while still updating lists, or overflows a pool) runs one lane per search.
    if (useFrontier) {
        if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: prologue done\n", tms(tStart, tnow()));
        if (afterLaunch) { std::function<int()> f; f.swap(afterLaunch); TRY(f()); }   // (the side-stream scoring starts alongside)
        if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: side-stream scoring queued\n", tms(tStart, tnow()));
        FrontierStats fs;
        // An item of the frontier tier costs about what half a (search, branch) pair costs the dense tier on full walks (1.1e9
        // items/s against 2.3e9 pairs/s), so a search only pays for a row of the whole tree once it has expanded half a tree's
        // worth of items; the searches from zero-length branches (whole-tree searches without an error model) never start here.
        // (With an error model there are no searches known to be whole-tree ones beforehand, a fifth of all searches is long, and
        // half a tree's worth of items each does not fit any pool -- 5e8 items at 100 000 tips and counting: those searches
        // leave at the lane tiers' budget.)
        const int frontierBudget = !hybrid ? 0 : (c->dm.usingErrorRate ? wideBudget
                                                                       : std::max(wideBudget, sp->wideSearchBudget == 0 ? c->n_scored / 2 : 0));
        // the searches scored on the side stream stay in the tier: their updating steps run with everybody else's, their clades
        // in the cached regime are scanned over the rows (k_fr_replay_wide)
        FrontierWide fw{nullptr, nullptr, FiniteRows{nullptr, nullptr, 0}, nullptr, 0};
        if (!preIdx.empty() && !c->tuning.wideOutsideFrontier) {
            fw.rowOf = preRowOf.data(); fw.cacheS = c->s_cache.p; fw.rowsReady = c->ev_join;
            if (useFin) fw.fin = FiniteRows{c->s_fin_mask.p, c->s_fin_prefix.p, finWords};
            if (matPre) {                                               // the frames' nesting, for the clade scans' short lists
                TRY(h2d(c, c->s_frame_parent, preFrameParent.data(), preFrameParent.size()));
                TRY(h2d(c, c->s_frame_node, preFrameNode.data(), preFrameNode.size()));
                fw.frameParent = c->s_frame_parent.p; fw.frameNode = c->s_frame_node.p; fw.nFrames = (int)preFrameParent.size();
            }
        }
        // With an error model nothing says beforehand which searches are long (no routing hint), and a long one expands a whole
        // budget of items here before it leaves for the dense tier -- half of this pass's items at 100 000 tips.  What does say it:
        // the node's search ran over the budget the last time it was searched on this tree (rounds repeat over the same nodes;
        // maple_tree_patch keeps node ids).  Such a search leaves at once.  Only a hint: the dense tier is exact for any search.
        std::vector<uint8_t> overHint;
        if (hybrid && c->dm.usingErrorRate && !c->tuning.noOverHint && (int)c->h_over_hint.size() >= c->dtree.n) {
            overHint.resize((size_t)n);
            size_t nHint = 0;
            hintedNow.assign((size_t)n, 0);
            for (int i = 0; i < n; i++) {
                hintedNow[i] = c->h_over_hint[todo[i]];
                overHint[i] = hintedNow[i] && !(!preRowOf.empty() && preRowOf[i] >= 0);   // (with a row: a whole-tree search inside this pass)
                nHint += overHint[i];
            }
            if (!nHint) overHint.clear();
            else if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %zu searches that ran over the budget last time go to the dense tier at once\n", tms(tStart, tnow()), nHint);
        }
        TRY(frontier_search(c, P, n, todo.data(), frontierBudget, (hybrid && wideBudget > MAPLE_ZERO_DIST_BUDGET) ? MAPLE_ZERO_DIST_BUDGET : (1 << 30),
                            ho.data(), poolW, poolA, poolUsed, poolCapW, poolCapA, &fs, 0, fw.rowOf ? &fw : nullptr,
                            overHint.empty() ? nullptr : overHint.data()));
        if (hybrid && c->dm.usingErrorRate) {
            // what this pass saw: over the budget -> the hint is set; a hinted search's hint is looked at again when its result is in
            if ((int)c->h_over_hint.size() < c->dtree.n) c->h_over_hint.resize((size_t)c->dtree.n, 0);
            for (int i = 0; i < n; i++)
                if (ho[i].status == -5 && ho[i].nAppend >= 0 && (overHint.empty() || !overHint[i])) c->h_over_hint[todo[i]] = 1;   // (nAppend -1: left for shorten(), k_fr_begin -- not over the budget)
        }
        std::vector<int32_t> todoFb, slotFb;
        for (int i = 0; i < n; i++)
            if (ho[i].status == FR_STATUS_FALLBACK) { todoFb.push_back(todo[i]); slotFb.push_back(i); }
        if (dbgT)
            fprintf(stderr, "[maple] t=%.1f ms: frontier tier done: %d levels, %lld updating + %lld cached items, %lld temporary lists "
                            "(%lld words, %lld aux), %lld refined records, %zu searches handed back%s\n", tms(tStart, tnow()), fs.levels,
                    fs.itemsUpdating, fs.itemsCached, fs.tempLists, fs.tempWords, fs.tempAux, fs.records, todoFb.size(),
                    fs.overflow ? " (a pool overflowed)" : "");
        c->last_search_frontier_only = todoFb.empty() && !hybrid;
        if (!todoFb.empty()) TRY(run_queries(todoFb, slotFb, nullptr, hybrid ? wideBudget : 0, nullptr, 0));
    } else
        TRY(run_queries(todo, slot, nullptr, hybrid ? wideBudget : 0, nullptr, 0));
    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: budgeted pass done\n", tms(tStart, tnow()));
    if (hybrid) {
        std::vector<int32_t> wide;
        for (int i = 0; i < n; i++)
            if (ho[i].status == -5) { wide.push_back(i); if (ho[i].bestNode == -2) heavyQueries = true; }
        if (!preIdx.empty()) {
            // replay what was scored on the side, together with as many unannounced ones as the spare rows take
            const int mZ = (int)preIdx.size();
            std::vector<int32_t> qn, sl, rowsNow, rest, ql2;
            std::vector<uint8_t> qt2;
            std::vector<double> qb2;
            for (int32_t i : wide) {
                const int node = nodes[i];
                // (a tree with local references: what the tier did not finish over these rows goes frame by frame, below -- the
                // one-wavefront-per-search replay wants the removed list in every frame)
                if (matPre) { rest.push_back(i); continue; }
                if (preRowOf[i] >= 0) { qn.push_back(node); sl.push_back(i); rowsNow.push_back(preRowOf[i]); }
                else if ((int)ql2.size() < preSpare) {
                    qn.push_back(node); sl.push_back(i); rowsNow.push_back(mZ + (int)ql2.size());
                    ql2.push_back(c->h_tree_lower[node]); qt2.push_back(c->h_tree_tip[node]); qb2.push_back(c->h_tree_dist[node]);
                } else rest.push_back(i);
            }
            if (!ql2.empty()) {
                const int m2 = (int)ql2.size();
                TRY(h2d(c, c->s_i32[6], ql2.data(), (size_t)m2));
                TRY(h2d(c, c->s_u8[3], qt2.data(), (size_t)m2));
                TRY(h2d(c, c->s_f64[3], qb2.data(), (size_t)m2));
                double qBytes = 0.0;
                for (int k = 0; k < m2; k++) qBytes += 8.0 * c->h_n_ent[ql2[k]] + 8.0 * c->h_n_aux[ql2[k]];
                TRY(launch_append_queries(c, c->stream, m2, c->s_i32[6].p, c->n_scored, c->t_i32[8].p, 0, 0.0,
                                          c->s_cache.p + (size_t)mZ * nTpre, nTpre, c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p,
                                          MAPLE_K_SPR_SCORE, (double)m2 * c->scored_bytes_total + qBytes, nullptr, nullptr, nullptr, 0, 1,
                                          useFin ? c->s_fin_mask.p + (size_t)mZ * finWords : nullptr));
                if (useFin) TRY(fin_prefix(c->stream, (size_t)mZ, (size_t)m2));
            }
            if (afterLaunch) { std::function<int()> f; f.swap(afterLaunch); TRY(f()); }   // (no lane launch took it)
            HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
            if (!qn.empty()) {
                rowOverride = &rowsNow;
                if (useFin) finRows = FiniteRows{c->s_fin_mask.p, c->s_fin_prefix.p, finWords};
                const int rc = run_queries(qn, sl, c->s_cache.p, 0, nullptr, 0);
                finRows = FiniteRows{nullptr, nullptr, 0};
                rowOverride = nullptr;
                if (rc != MAPLE_OK) return rc;
            }
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %zu pre-scored and %zu other wide searches replayed, %zu left\n", tms(tStart, tnow()), qn.size() - ql2.size(), ql2.size(), rest.size());
            wide.swap(rest);
        }
        const PlaceMeta &F = *c->place;
        const int nT = c->dtree.n, nF = c->tree_has_mut ? F.nF : 1;
        const size_t rowBytes = (size_t)nT * sizeof(double);
        size_t cacheBudget = (size_t)4ull << 30;                      // (query x node) score table: 4 GiB, more on big trees
        {   // a row of a 1 000 000-tip tree is 16 MB and every launch over the table wants thousands of searches (one per
            // wavefront): up to half of what is free, 96 GiB at most
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                const size_t half = (freeB + c->s_cache.cap * sizeof(double)) / 2;
                cacheBudget = std::max(cacheBudget, std::min(half, (size_t)96ull << 30));
            }
        }
        size_t chunk = cacheBudget / rowBytes;
        if (chunk < 1) chunk = 1;
        // Trees with local references: the rows of the searches that ran over their budget are made in the ROOT's frame as well
        // (appendProbNode does not depend on the frame, see above) and replayed inside the frontier tier; only what the tier hands
        // back goes the frame-by-frame way.
        const bool rootRowsOK = nF > 1 && useFrontier && c->scan_valid && c->place && !c->tuning.noCladeScan && !c->tuning.wideOutsideFrontier
                                && wide.size() >= 64;
        std::vector<int32_t> frameWay;                                  // (searches of a tree with references left to the old path)
        for (int wpass = 0; wpass < 2; wpass++) {
        if (wpass == 1) { if (frameWay.empty()) break; wide.swap(frameWay); frameWay.clear(); }
        const bool rootRows = rootRowsOK && wpass == 0;
        size_t w0 = 0;
        while (w0 < wide.size()) {
            int m = (int)std::min(chunk, wide.size() - w0);
            if (nF > 1 && !rootRows) {
                // the removed list goes into EVERY reference frame: bound the batch by what the arena can take
                const int64_t freeEnt = (c->cap_ent - c->used_ent) / 3, freeAux = (c->cap_aux - c->used_aux) / 3;
                int64_t needEnt = 0, needAux = 0;
                int k = 0;
                for (; k < m; k++) {
                    const int32_t l = c->h_tree_lower[nodes[wide[w0 + k]]];
                    needEnt += (int64_t)nF * (c->h_n_ent[l] + 24);
                    needAux += (int64_t)nF * (c->h_n_aux[l] + 8);
                    if (needEnt > freeEnt || needAux > freeAux || (int64_t)(k + 1) * nF > (8ll << 20)) break;
                }
                if (k < m && k < 2048) {
                    // Too many frames for this arena (a 100 000-tip tree has ~2 000): batches this small would turn the
                    // replay into a chain of one-lane launches.  The remaining wide searches run lane-only instead.
                    std::vector<int32_t> rest, restSlot;
                    for (size_t w = w0; w < wide.size(); w++) { rest.push_back(nodes[wide[w]]); restSlot.push_back(wide[w]); }
                    TRY(run_queries(rest, restSlot, nullptr, 0, nullptr, 0));
                    break;
                }
                m = k;
            }
            std::vector<int32_t> qn(m), ql(m), sl(m);
            std::vector<uint8_t> qt(m);
            std::vector<double> qb(m);
            for (int k = 0; k < m; k++) {
                const int node = nodes[wide[w0 + k]];
                qn[k] = node; sl[k] = wide[w0 + k];
                ql[k] = c->h_tree_lower[node];                         // the removed subtree's lower list (M:6838)
                qt[k] = c->h_tree_tip[node];                           // isRemovedTip (M:6846)
                qb[k] = c->h_tree_dist[node];                          // removedBLen = dist[node] (M:9644)
            }
            if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: wide chunk of %d searches starts\n", tms(tStart, tnow()), m);
            {
                size_t need = (size_t)m * nT;
                if (need > c->s_cache.cap) need = std::min(std::max(need, 2 * c->s_cache.cap), std::max(chunk * (size_t)nT, need));
                HIPCK(c, c->s_cache.reserve_exact(need));
            }
            if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: score table reserved\n", tms(tStart, tnow())); }
            TRY(h2d(c, c->s_u8[3], qt.data(), (size_t)m));
            TRY(h2d(c, c->s_f64[3], qb.data(), (size_t)m));
            if (nF == 1 || rootRows) {
                int64_t chunkMark = -1;
                if (rootRows) {                                        // candidates (once per tree) and this chunk's removed lists in the root's frame
                    TRY(ensure_cand_root(c));
                    TRY(maple_arena_mark(c, &chunkMark));
                    std::vector<int32_t> qf(m);
                    for (int k = 0; k < m; k++) qf[k] = F.frameOf[qn[k]];
                    TRY(lists_to_root_frame(c, ql, qf));
                    useFin = true;
                }
                TRY(h2d(c, c->s_i32[6], ql.data(), (size_t)m));
                double qBytes = 0.0;                                   // each query list once per launch
                for (int k = 0; k < m; k++) qBytes += 8.0 * c->h_n_ent[ql[k]] + 8.0 * c->h_n_aux[ql[k]];
                TRY(fin_reserve((size_t)m));
                TRY(launch_append_queries(c, c->stream, m, c->s_i32[6].p, c->n_scored, rootRows ? c->s_cand_root.p : c->t_i32[8].p, 0, 0.0, c->s_cache.p, nT,
                                          rootRows ? c->t_cand_rank.p : c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p, MAPLE_K_SPR_SCORE,
                                          (double)m * c->scored_bytes_total + qBytes, nullptr, nullptr, nullptr, 0, 1, useFin ? c->s_fin_mask.p : nullptr));
                if (useFin) TRY(fin_prefix(c->stream, 0, (size_t)m));
                if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: scored\n", tms(tStart, tnow())); }
                if (useFin) finRows = FiniteRows{c->s_fin_mask.p, c->s_fin_prefix.p, finWords};
                if (useFrontier && useFin && !c->tuning.wideOutsideFrontier && m >= 64) {
                    // the searches that ran over their budget: back through the frontier tier as whole-tree searches -- their
                    // updating steps batched, their clades scanned over the rows just made (k_fr_replay_wide); what the tier
                    // hands back is replayed one wavefront per search as before, on its own row
                    std::vector<int32_t> rowId(m);
                    for (int k = 0; k < m; k++) rowId[k] = k;
                    FrontierWide fw2{rowId.data(), c->s_cache.p, finRows, nullptr, 1};
                    if (rootRows) {
                        TRY(h2d(c, c->s_frame_parent, F.frameParent.data(), F.frameParent.size()));
                        TRY(h2d(c, c->s_frame_node, F.frameNode.data(), F.frameNode.size()));
                        fw2.frameParent = c->s_frame_parent.p; fw2.frameNode = c->s_frame_node.p; fw2.nFrames = (int)F.frameParent.size();
                    }
                    std::vector<SearchOut> part(m);
                    FrontierStats fs2;
                    const int rcF = frontier_search(c, P, m, qn.data(), 1 << 30, 0, part.data(), poolW, poolA, poolUsed, poolCapW, poolCapA, &fs2, 0, &fw2);
                    if (rcF != MAPLE_OK) { finRows = FiniteRows{nullptr, nullptr, 0}; return rcF; }
                    std::vector<int32_t> qn2, sl2, rows2;
                    for (int k = 0; k < m; k++) {
                        if (part[k].status == FR_STATUS_FALLBACK || part[k].status == -5) { qn2.push_back(qn[k]); sl2.push_back(sl[k]); rows2.push_back(k); }
                        else ho[sl[k]] = part[k];
                    }
                    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: %d searches over budget replayed inside the frontier tier (%d levels), %zu handed back\n",
                                      tms(tStart, tnow()), m, fs2.levels, qn2.size());
                    qn.swap(qn2); sl.swap(sl2);
                    if (rootRows) {                                     // (the one-wavefront-per-search replay wants the list in every frame)
                        for (int32_t i : sl) frameWay.push_back(i);
                        qn.clear();
                    }
                    if (!qn.empty()) rowOverride = &rows2;
                    const int rcW = qn.empty() ? MAPLE_OK : run_queries(qn, sl, c->s_cache.p, 0, nullptr, 0);
                    rowOverride = nullptr;
                    finRows = FiniteRows{nullptr, nullptr, 0};
                    TRY(rcW);
                } else if (rootRows) {                                  // (a tail chunk too small for the tier: the frame-by-frame way)
                    for (int32_t i : sl) frameWay.push_back(i);
                    finRows = FiniteRows{nullptr, nullptr, 0};
                } else {
                const int rcW = run_queries(qn, sl, c->s_cache.p, 0, nullptr, 0);
                finRows = FiniteRows{nullptr, nullptr, 0};
                TRY(rcW);
                }
                if (chunkMark >= 0) TRY(maple_arena_release(c, chunkMark));
                if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: replayed\n", tms(tStart, tnow()));
            } else {
                // the removed list in every MAT reference frame, along the paths the traversal itself takes
                // (passGenomeListThroughBranch up the chain of enclosing frames, M:6844-6847 / 7392, then down into every
                // other frame from the nearest frame already known, M:7119 / 7342)
                int64_t mark = 0;
                TRY(maple_arena_mark(c, &mark));
                std::vector<int32_t> R((size_t)m * nF, -1), cur(m), src, ml, out;
                std::vector<uint8_t> dir;
                for (int k = 0; k < m; k++) { cur[k] = F.frameOf[qn[k]]; R[(size_t)k * nF + cur[k]] = ql[k]; }
                for (;;) {                                             // up, one enclosing frame per round
                    src.clear(); ml.clear();
                    std::vector<int> who;
                    for (int k = 0; k < m; k++)
                        if (cur[k] != 0) { who.push_back(k); src.push_back(R[(size_t)k * nF + cur[k]]); ml.push_back(c->h_tree_mut[F.frameNode[cur[k]]]); }
                    if (who.empty()) break;
                    dir.assign(who.size(), 1);
                    out.resize(who.size());
                    TRY(maple_pass_branch_batch(c, (int32_t)who.size(), src.data(), ml.data(), dir.data(), out.data()));
                    for (size_t i = 0; i < who.size(); i++) {
                        const int k = who[i];
                        cur[k] = F.frameParent[cur[k]];
                        R[(size_t)k * nF + cur[k]] = out[i];
                    }
                }
                // down, one nesting level per round: on the device (k_fan_*), all (query, frame) items of the level at once
                double qBytes = 0.0;                                   // every frame's copy of the query that is read
                for (size_t k = 0; k < R.size(); k++) if (R[k] >= 0) qBytes += 8.0 * c->h_n_ent[R[k]] + 8.0 * c->h_n_aux[R[k]];
                TRY(h2d(c, c->s_i32[6], R.data(), R.size()));
                {
                    std::vector<int32_t> fm((size_t)nF, 0);
                    for (int f = 1; f < nF; f++) fm[f] = c->h_tree_mut[F.frameNode[f]];
                    TRY(h2d(c, c->s_i32[4], F.frameParent.data(), (size_t)nF));
                    TRY(h2d(c, c->s_i32[5], fm.data(), (size_t)nF));
                    HIPCK(c, hipStreamSynchronize(c->stream));
                }
                int a = 1;
                for (size_t l = 0; l < F.levelStart.size(); l++) {
                    const int b = F.levelStart[l];
                    TRY(fan_out_level(c, m, nF, a, b, c->s_i32[6].p, c->s_i32[4].p, c->s_i32[5].p, c->s_fan, c->s_fan_tmp, &qBytes));
                    a = b;
                }
                if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: removed lists in all %d frames\n", tms(tStart, tnow()), nF); }
                if (c->n_frame_chunks > 0 && m >= 32)
                    TRY(launch_append_queries(c, c->stream, m, c->s_i32[6].p, c->n_scored, c->t_i32[8].p, 0, 0.0, c->s_cache.p, nT,
                                              c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p, MAPLE_K_SPR_SCORE,
                                              (double)m * c->scored_bytes_total + qBytes, nullptr, nullptr, c->t_frame_chunks.p,
                                              c->n_frame_chunks, nF));
                else
                TRY(launch_place_score(c, m, nF, c->s_i32[6].p, c->n_scored, c->t_i32[8].p, c->t_scored_frame.p, 0, 0.0,
                                       c->s_cache.p, nT, c->t_scored_col.p, c->s_u8[3].p, c->s_f64[3].p, MAPLE_K_SPR_SCORE,
                                       (double)m * c->scored_bytes_total + qBytes));
                if (dbgT) { HIPCK(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "[maple] t=%.1f ms: scored\n", tms(tStart, tnow())); }
                TRY(run_queries(qn, sl, c->s_cache.p, 0, c->s_i32[6].p, nF));
                if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: replayed\n", tms(tStart, tnow()));
                TRY(maple_arena_release(c, mark));
            }
            w0 += (size_t)m;
        }
        }   // (wpass)
    }
#ifdef MAPLE_SPR_PROFILE
    {
        for (int pass = 0; pass < 2; pass++) {
        long long ts = 0, tr = 0, tf = 0, ns = 0, nl = 0, mxs = 0, mxr = 0, mxf = 0, mxl = 0, cnt = 0, visits = 0;
        for (int i = 0; i < n; i++) {
            if (pass == 1 && !(ho[i].status == 0 && ho[i].nAppend <= wideBudget)) continue;   // pass 1: finished by the lane tier
            cnt++; visits += ho[i].nAppend;
            ts += ho[i].tStep; tr += ho[i].tReplay; tf += ho[i].tRefine; ns += ho[i].nSteps; nl += ho[i].nShortList;
            mxs = std::max<long long>(mxs, ho[i].tStep); mxr = std::max<long long>(mxr, ho[i].tReplay);
            mxf = std::max<long long>(mxf, ho[i].tRefine); mxl = std::max<long long>(mxl, ho[i].nShortList);
        }
        {
            std::vector<double> tot;
            for (int i = 0; i < n; i++) {
                const bool laneTier = ho[i].status == 0 && ho[i].nAppend <= wideBudget;
                if (ho[i].status != 0 || (pass == 1) != laneTier) continue;
                tot.push_back((ho[i].tStep + ho[i].tReplay + ho[i].tRefine) * 1e-5);
            }
            if (c->tuning.verbose > 1) {
                std::vector<int> idx;
                for (int i = 0; i < n; i++) if (ho[i].status == 0 && (ho[i].nAppend <= wideBudget) == (pass == 1)) idx.push_back(i);
                std::sort(idx.begin(), idx.end(), [&](int a, int b) {
                    return ho[a].tStep + ho[a].tReplay + ho[a].tRefine > ho[b].tStep + ho[b].tReplay + ho[b].tRefine; });
                for (size_t k = 0; k < std::min<size_t>(12, idx.size()); k++) {
                    const int i = idx[k];
                    const int32_t l = c->h_tree_lower[nodes[i]];
                    fprintf(stderr, "[maple]   slow search: node %d depth %d, %d placements, %d updating steps %.1f ms, visits %.1f ms, %d refinements %.1f ms, removed list %d entries\n",
                            nodes[i], c->h_depth[nodes[i]], ho[i].nAppend, ho[i].nSteps, ho[i].tStep * 1e-5, ho[i].tReplay * 1e-5,
                            ho[i].nShortList, ho[i].tRefine * 1e-5, l >= 0 ? c->h_n_ent[l] : -1);
                }
            }
            std::sort(tot.begin(), tot.end());
            if (!tot.empty())
                fprintf(stderr, "[maple] per-search time, %s (%zu): median %.2f ms, p90 %.2f, p99 %.2f, p99.9 %.2f, max %.2f\n",
                        pass ? "lane tier" : "dense tier", tot.size(), tot[tot.size() / 2], tot[tot.size() * 9 / 10],
                        tot[tot.size() * 99 / 100], tot[tot.size() * 999 / 1000], tot.back());
        }
        {
            long long tw = 0, tl = 0;
            for (int i = 0; i < n; i++) {
                if (pass == 1 && !(ho[i].status == 0 && ho[i].nAppend <= wideBudget)) continue;
                if (ho[i].rprWoff < 0) { tw += ho[i].rprN; tl += ho[i].rprNA; }
            }
            fprintf(stderr, "[maple]   of which inside append_walk %.1f ms, list lookup before it %.1f ms\n", tw * 1e-3, tl * 1e-3);
        }
        fprintf(stderr, "[maple] profile over %lld searches (%s; last launch each; %lld placements): updating steps %.1f ms total (max %.2f), "
                        "other visits %.1f (max %.2f), refine %.1f (max %.2f); %lld updating steps, %lld short-listed branches (max %lld)\n",
                cnt, pass ? "lane tier only" : "all", visits, ts * 1e-5, mxs * 1e-5, tr * 1e-5, mxr * 1e-5, tf * 1e-5, mxf * 1e-5, ns, nl, mxl);
        }
    }
#endif
    if (preMark >= 0) TRY(maple_arena_release(c, preMark));             // (the root-frame copies of this call)
    // (a hinted search that turned out short -- the tree changed around it -- is tried within the budget again next time)
    for (size_t i = 0; i < hintedNow.size(); i++)
        if (hintedNow[i] && ho[i].status == 0 && ho[i].nAppend <= wideBudget / 4) c->h_over_hint[nodes[i]] = 0;
    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: results in host memory\n", tms(tStart, tnow()));
    for (int i = 0; i < n; i++) {
        bestNode[i] = ho[i].bestNode; bestScore[i] = ho[i].bestScore;
        blen3[3 * i] = ho[i].blen[0]; blen3[3 * i + 1] = ho[i].blen[1]; blen3[3 * i + 2] = ho[i].blen[2];
        placement[i] = ho[i].placement; improvement[i] = ho[i].improvement; currentLK[i] = ho[i].currentLK;
        nAppend[i] = ho[i].nAppend; status[i] = ho[i].status;
    }
    if (outRprList) {                                                  // bestRemovedPartials become arena lists
        std::vector<int64_t> woff(n, 0), aoff(n, 0);
        std::vector<int32_t> ne(n, -1), na(n, 0);
        for (int i = 0; i < n; i++)
            if (ho[i].status == 0 && ho[i].rprWoff >= 0) { woff[i] = ho[i].rprWoff; aoff[i] = ho[i].rprAoff; ne[i] = ho[i].rprN; na[i] = ho[i].rprNA; }
        HIPCK(c, c->s_i32[2].reserve(n));
        HIPCK(c, c->s_i32[3].reserve(n));
        HIPCK(c, hipMemcpyAsync(c->s_i32[2].p, ne.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->s_i32[3].p, na.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        TRY(stage_begin(c, (size_t)n * 48 + 256));
        STAGE(dwo, c, woff.data(), n); STAGE(dao, c, aoff.data(), n);
        TRY(stage_flush(c));
        TRY(commit_lists(c, n, dwo, dao, c->s_i32[2].p, c->s_i32[3].p, outRprList, poolW, poolA));
    }
    if (dbgT) fprintf(stderr, "[maple] t=%.1f ms: outputs written\n", tms(tStart, tnow()));
    return MAPLE_OK;
}

extern "C" int maple_spr_search_visited(maple_ctx *c, int64_t cap, int32_t *query, int32_t *node, int64_t *n)
{
    if (!c || cap < 0 || !n || (cap && (!query || !node))) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    if (!c->last_search_frontier_only)
        return fail(c, MAPLE_ERR_STATE, "the last maple_spr_search_batch did not run wholly in the frontier tier (use wideSearchBudget < 0)");
    long long nn = 0;
    const int rc = frontier_export(c, cap, query, node, &nn);
    *n = nn;
    return rc;
}

#ifdef MAPLE_DEBUG_ABI
extern "C" int maple_debug_frontier_levels(maple_ctx *c, int32_t cap, int64_t *itemsUpdating, int64_t *itemsCached, float *msUpdating,
                                           float *msCached, int32_t *n, int64_t *waveItemsSmall, int64_t *waveItemsBig)
{
    if (!c || cap < 0 || !itemsUpdating || !itemsCached || !msUpdating || !msCached || !n) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int nn = 0;
    const int rc = frontier_level_profile(c, cap, (long long *)itemsUpdating, (long long *)itemsCached, msUpdating, msCached, &nn,
                                          (long long *)waveItemsSmall, (long long *)waveItemsBig);
    *n = nn;
    return rc;
}
#endif

