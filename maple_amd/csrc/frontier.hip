// maple_amd/csrc/frontier.hip -- the FRONTIER TIER of the SPR regraft search (findBestParentTopology, M:6817-7724, and the
// worker body of startTopologyUpdatesParallel, M:9615-9711), for trees without MAT local references.
//
// The reference's search is a depth-first walk with order-dependent pruning: the running best (bestLKdiff) and the
// failedPasses counter decide which branches are descended into (M:7090-7103, 7311-7323).  What an item of its stack
// COMPUTES, however -- the lists merged along the path while needsUpdating holds, the placement score
// appendProbNode(midTot, removed list) -- depends only on the path from the pruned node to that item, never on the order in
// which the stack is emptied.  So the search is split in two:
//
//  1. EXPANSION (order-free, data-parallel over every item of every search of the batch, level by level).  Each item is
//     scored by one lane and pushes its children under a PERMISSIVE form of the reference's rules that needs nothing but the
//     path: the running best is replaced by pathBest = max(current placement cost, scores of the item's ancestors on its
//     path) <= the running best the reference would hold when it reaches the item (ancestors are visited before it), and
//     failedPasses is reset whenever a score beats pathBest (a superset of the reference's resets).  Both rules are
//     monotone in the running best, so the set of expanded items is a SUPERSET of the items the reference visits, and
//     every expanded item carries exactly the score, lists and flags the reference would compute for it.
//  2. REPLAY (exact, one lane per search, integer and compare work only).  The reference's LIFO walk is run over the item
//     tree of step 1 with the real running best and failedPasses: same visiting order, same short list (M:7071, 7293), same
//     candidate count.  Short-listed branches of all searches are then refined in one batch (evaluatePlacement,
//     M:6790-6806; one lane per record), and a last pass per search applies the reference's final selection (M:7635) and the
//     worker's accept rule and vetoes (M:9681-9700).
//
// Results are those of the one-lane-per-search kernel (k_spr_search, search_dev.h) bit for bit; what changes is the shape
// of the work: ~10^7 independent (search, branch) items per launch sequence instead of 10^5 chains of dependent list walks.
//
// The reference shortens the removed list in place at every improvement (M:7087).  Without local references the removed
// list is the pruned node's own lower list, which is stored shortened, so that is a no-op; a search whose removed list
// WOULD change is handed back (status FR_FALLBACK), as are searches that touch the root while still updating lists
// (rootVector, M:6916-6960 / 7406-7432), run out of scratch, or exceed the pools.  Searches that expand more than `budget`
// items are whole-tree searches and go to the dense tier (status -5), as before.
#include "ctx_host.h"
#include "frontier_dev.h"

#include <chrono>

using namespace frt;

namespace {

// the arena's list table as one record per list (FPools::arec), from its four arrays: at the start of every call, for every list
// the arena holds (1 000 000 tips: 10 M lists, 0.4 GB moved: ~0.1 ms)
__global__ __launch_bounds__(FR_BLOCK) void k_fr_arena_recs(ArenaViewS av, int n, LRec *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = lrec_make(av.ent_off[i], av.n_ent[i], av.aux_off[i], av.n_aux[i]);
}

// ---- the worker's prologue (M:9626-9674) and the seeding of nodesToVisit (M:6855-6914) ----------------------------------
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(FR_BLOCK) void k_fr_begin(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, SearchParams P, int n,
                                                       const int32_t *nodes, FPools fp, SearchOut *out, int budget, int zeroBudget,
                                                       const int32_t *rowOf, int forceWide, const uint8_t *overHint)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const int node = nodes[q];
        SearchOut &o = out[q];
        o.bestNode = -1; o.placement = -1; o.status = 0; o.nAppend = 0;
        o.bestScore = 0.0; o.improvement = 0.0; o.currentLK = 0.0;
        o.blen[0] = o.blen[1] = o.blen[2] = 0.0;
        o.rprWoff = o.rprAoff = -1; o.rprN = o.rprNA = 0;
        o.nShortList = o.nSteps = 0; o.tStep = o.tReplay = o.tRefine = 0;
        FSearch &S = fp.S[q];
        S.node = node; S.parent = -1; S.sibling = -1; S.hRpr0 = -1; S.seed0 = S.seed1 = FR_NONE; S.nItems = 0; S.state = FS_FINAL;
        S.slHead = FR_NONE; S.nApp = 0; S.recBase = 0; S.recCount = 0; S.isRemovedTip = 0; S.removedBLen = 0.0; S.curLK = 0.0;
        const NodeRec rn = T.nd[node];
        const int parent = rn.up;
        if (parent < 0) { o.status = 1; continue; }                          // the root cannot be re-placed (M:9626)
        const NodeRec rp = T.nd[parent];
        const int childIdx = (rp.c0 == node) ? 0 : 1;
        const int vectUp = ftree(childIdx == 0 ? rp.upRight : rp.upLeft);
        if (!fvalid(vectUp) || rn.lower < 0) { o.status = -1; continue; }
        const FList lu = flist(av, fp, vectUp), ll = flist(av, fp, ftree(rn.lower));
        const long long laneId = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        double curLK;
        if (fp.mat && rn.mutId >= 0 && fp.mv.cnt[rn.mutId] > 0) {           // the parent's upper list in the node's own frame, M:9637-9638
            const int cnt = fp.mv.cnt[rn.mutId];
            FScr scr{nullptr, nullptr};
            if (!fscratch(fp, laneId, lu.n + 2 * cnt, scr)) { S.state = FS_FALLBACK; continue; }
            Writer wr;
            wr.init(scr.w, scr.a);
            pass_walk(c.m.lRef, fref(lu), fp.mv.mut3 + 3 * fp.mv.off[rn.mutId], cnt, false, wr);
            curLK = append_walk(c, ListRef{scr.w, scr.a}, fref(ll), rn.isTip != 0, rn.dist);
        } else
        curLK = append_walk(c, fref(lu), fref(ll), rn.isTip != 0, rn.dist);   // M:9646
        o.currentLK = curLK;
        if (!(curLK < P.thrPlacement || rn.dist != 0.0)) { o.status = 2; continue; }        // M:9674
        // (the caller has seen this node's search run over the budget on this tree before: it goes to the dense tier without
        // expanding a budget's worth of items first -- same result either way, the dense tier is exact for any search)
        if (overHint && overHint[q]) { S.state = FS_OVER; o.status = -5; continue; }
        S.parent = parent; S.sibling = childIdx == 0 ? rp.c1 : rp.c0;
        S.isRemovedTip = rn.isTip; S.removedBLen = rn.dist; S.curLK = curLK;
        const NodeRec rs = T.nd[S.sibling];
        // the removed list as the search starts with it (M:6838-6846): in the frame of the pruned node's parent, and once more in
        // the sibling's (bestRemovedPartials).  A re-expressed list that shorten() (M:7087) would change: the one-lane kernel.
        int rpr = ftree(rn.lower);
        const int mergeLevel = shorten_would_merge(c, fref(ll), ll.n);      // M:7087 would edit the removed list (see fpass_removed)
        const bool wouldMerge = mergeLevel != 0;
        S.rprMerge0 = mergeLevel;
        int hBest = rpr;
        if (fp.mat) {
            rpr = fpass_removed(c, fp, av, laneId, rpr, rn.mutId, true);
            hBest = fvalid(rpr) ? fpass_removed(c, fp, av, laneId, rpr, rs.mutId, false) : rpr;
            if (!fvalid(rpr) || !fvalid(hBest)) { S.state = FS_FALLBACK; continue; }
        }
        S.hRpr0 = hBest;
        // a search from a zero-length branch without an error model is a whole-tree search (see k_spr_search): dense tier
        bool wide = false;
        if (forceWide && rowOf && rowOf[q] >= 0) {
            if (wouldMerge) { S.state = FS_OVER; o.status = -5; o.nAppend = -1; continue; }   // (the one-wavefront-per-search kernel edits the list in place; nAppend -1: NOT a search over the budget -- no routing hint from it)
            wide = true;
        } else if (!U && budget > zeroBudget && rn.dist == 0.0) {
            wide = rowOf && rowOf[q] >= 0 && !wouldMerge;
            if (!wide) { S.state = FS_OVER; o.status = -5; o.nAppend = -1; continue; }
        } else if (U && rowOf && rowOf[q] >= 0) {
            // (error model: the caller gave this search a row because it ran over the budget the last time)
            if (wouldMerge) { S.state = FS_OVER; o.status = -5; o.nAppend = -1; continue; }
            wide = true;
        }
        S.state = wide ? FS_WIDE : FS_ACTIVE;
        if (rp.up < 0) {                                                    // the parent is the root (M:6916-6960): seeded by an item
            S.seed0 = fpush(fp, budget, q, true, S.sibling, 3, -1, 0.0, curLK, 0, S.hRpr0, curLK);   // of its own (k_fr_updating)
            continue;
        }
        const int pp = rp.up;
        const NodeRec rpp = T.nd[pp];
        const bool first = rpp.c0 == parent;
        const double d = rs.dist + rp.dist;
        int pv1 = ftree(rs.lower), rprUp = rpr, vUpUp = ftree(first ? rpp.upRight : rpp.upLeft), rprDown = rpr;
        if (fp.mat) {                                                       // M:6876-6901
            pv1 = fpass_store(fp, av, c.m.lRef, laneId, pv1, rs.mutId, true);
            if (rp.mutId >= 0) {
                pv1 = fvalid(pv1) ? fpass_store(fp, av, c.m.lRef, laneId, pv1, rp.mutId, true) : pv1;
                rprUp = fpass_removed(c, fp, av, laneId, rpr, rp.mutId, true);
            }
            vUpUp = fpass_store(fp, av, c.m.lRef, laneId, vUpUp, rp.mutId, false);
            if (rs.mutId >= 0) {
                vUpUp = fvalid(vUpUp) ? fpass_store(fp, av, c.m.lRef, laneId, vUpUp, rs.mutId, false) : vUpUp;
                rprDown = hBest;                                            // (the same pass: the removed list in the sibling's frame)
            }
            if (pv1 == -2 || vUpUp == -2 || !fvalid(rprUp) || !fvalid(rprDown)) { S.state = FS_FALLBACK; continue; }
        }
        S.seed0 = fpush(fp, budget, q, true, pp, first ? 1 : 2, pv1, d, curLK, 0, rprUp, curLK);
        S.seed1 = fpush(fp, budget, q, true, S.sibling, 0, vUpUp, d, curLK, 0, rprDown, curLK);
    }
}

// The two streams of the expansion keep their own books.  k_fr_snap_u closes a level of the list-updating items and opens the
// next (after k_fr_begin: the first), and PUBLISHES what the kernels before it on its stream pushed into the cached pool: only
// kernels that have finished -- items that are written in full.  k_fr_snap_c opens a launch of the cached-regime items: what the
// cached launches before it pushed (same stream: finished) and the roots published so far.
#define FR_LVL 8                       // per level / launch in FPools::lvl: loU hiU | loC hiC loR hiR (roots: absolute refs) | wavefront-wide items: small, 512
__global__ void k_fr_snap_u(FCtr *ctr, long long capU, long long capR, long long capPassR, unsigned long long *lvl, int maxLevels)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // (a pool that overflowed keeps counting what was asked of it: the items themselves end at its capacity)
        ctr->loU = ctr->hiU; ctr->hiU = min(ctr->usedU, (unsigned long long)capU);
        // (the pass entries first: a cached launch that sees the new root count must see at least the pass entries that go with it --
        // k_fr_snap_c on the other stream reads the two in the opposite order)
        atomicExch(&ctr->safePassR, min(ctr->nPassR, (unsigned long long)capPassR));
        __threadfence();
        atomicExch(&ctr->safeR, min(ctr->usedR, (unsigned long long)capR));
        ctr->bigUsed = 0;
        const unsigned long long heavySmall = ctr->permHeavy, heavyBig = ctr->permHeavy2;
        ctr->permDown = ctr->permUp = ctr->permDownB = ctr->permUpB = ctr->permHeavy = ctr->permHeavy2 = 0;
        if (lvl) {
            const int l = ctr->nLevels++;
            if (l < maxLevels) { lvl[FR_LVL * l] = ctr->loU; lvl[FR_LVL * l + 1] = ctr->hiU; lvl[FR_LVL * l + 6] = lvl[FR_LVL * l + 7] = 0; }
            if (l > 0 && l - 1 < maxLevels) { lvl[FR_LVL * (l - 1) + 6] = heavySmall; lvl[FR_LVL * (l - 1) + 7] = heavyBig; }   // (the level before: its wavefront-wide items)
        }
    }
}
__global__ void k_fr_snap_c(FCtr *ctr, long long capCC, long long capPass, long long capDeferred, unsigned long long *lvl, int maxLevels)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctr->loC = ctr->hiC; ctr->hiC = min(ctr->usedC, (unsigned long long)capCC);
        ctr->loR = ctr->hiR; ctr->hiR = atomicAdd(&ctr->safeR, 0ull);
        __threadfence();
        ctr->loP = ctr->hiP; ctr->hiP = min(ctr->nPass, (unsigned long long)capPass);
        ctr->loPR = ctr->hiPR; ctr->hiPR = atomicAdd(&ctr->safePassR, 0ull);
        ctr->bigUsedC = 0;
        ctr->loD = ctr->hiD; ctr->hiD = min(ctr->nDeferred, (unsigned long long)capDeferred);
        if (lvl) {
            const int k = ctr->nLevelsC++;
            if (k < maxLevels) {
                lvl[FR_LVL * k + 2] = ctr->loC; lvl[FR_LVL * k + 3] = ctr->hiC;
                lvl[FR_LVL * k + 4] = (unsigned long long)capCC + ctr->loR; lvl[FR_LVL * k + 5] = (unsigned long long)capCC + ctr->hiR;
            }
        }
    }
}

// ---- trees with MAT local references: the removed list of the level's items that were pushed across a reference branch --------
// The item's hRpr is still the list of the item that pushed it; here it goes through the branch (passGenomeListThroughBranch,
// M:7111-7118 / 7148-7155 on the way down, 7359-7366 / 7388-7395 on the way up).  One lane per such item (about one push in a
// hundred crosses a reference branch); a list that shorten() (M:7087) would change hands its search to the one-lane kernel.
// A launch with FEW such items lasts as long as its slowest list, and one lane takes 0.1-0.4 ms for one (0.35 ms per launch,
// 56 launches per round): those launches take a WAVEFRONT per item (wave_pass, frontier_dev.h: one lane per entry of the list).
// With many items the lanes win: 65 536 of them at 0.3 ms against 4 096 wavefronts at ~50 us (a round's large launches hold
// ~170 000 such items: 10-15 ms per launch by wavefronts, measured, against 4-5).
#define FR_PASS_WAVE_BELOW 8192
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(FR_BLOCK) void k_fr_pass_wave(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, FPools fp)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    // (the launch's two kinds of items: pushed by cached-regime items, and roots)
    const long long loA = (long long)fp.ctr->loP, nA = (long long)fp.ctr->hiP - loA, loB = (long long)fp.ctr->loPR, nB = (long long)fp.ctr->hiPR - loB;
    if (nA + nB > FR_PASS_WAVE_BELOW) return;                                // (k_fr_pass: one lane per item)
    for (long long i = wave; i < nA + nB; i += nWaves) {
        FItem &it = item_of(fp, i < nA ? fp.passList[loA + i] : fp.passListR[loB + (i - nA)]);
        FSearch &S = fp.S[it.q];
        if (!fs_live(S.state)) continue;
        const NodeRec r1 = T.nd[it.t1];
        // dir 0: came down the branch above t1; dir 1 / 2: came up the branch above t1's child 0 / 1
        const int mutId = it.dir == 0 ? r1.mutId : T.nd[it.dir == 1 ? r1.c0 : r1.c1].mutId;
        const int h = wave_pass(c, fp, av, it.hRpr, mutId, it.dir != 0, true);
        if (lane == 0) {
            if (!fvalid(h)) S.state = FS_FALLBACK;
            else {
                it.hRpr = h;
                const unsigned long long k = atomicAdd(&fp.ctr->nDeferred, 1ull);
                if ((long long)k < fp.capDeferred) fp.deferred[k] = (int32_t)(&it - fp.C);
                else { S.state = FS_FALLBACK; fp.ctr->overflow = 1; }
            }
        }
    }
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(FR_BLOCK) void k_fr_pass(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, FPools fp, long long slabBase)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const long long laneId = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long slab = slabBase + laneId;                                // (scratch slabs of its own: it runs next to k_fr_updating)
    // (the launch's two kinds of items: pushed by cached-regime items, and roots)
    const long long loA = (long long)fp.ctr->loP, nA = (long long)fp.ctr->hiP - loA, loB = (long long)fp.ctr->loPR, nB = (long long)fp.ctr->hiPR - loB;
    if (nA + nB <= FR_PASS_WAVE_BELOW) return;                               // (k_fr_pass_wave: a wavefront per item)
    for (long long i = laneId; i < nA + nB; i += (long long)gridDim.x * blockDim.x) {
        FItem &it = item_of(fp, i < nA ? fp.passList[loA + i] : fp.passListR[loB + (i - nA)]);
        FSearch &S = fp.S[it.q];
        if (!fs_live(S.state)) continue;
        const NodeRec r1 = T.nd[it.t1];
        // dir 0: came down the branch above t1; dir 1 / 2: came up the branch above t1's child 0 / 1
        const int mutId = it.dir == 0 ? r1.mutId : T.nd[it.dir == 1 ? r1.c0 : r1.c1].mutId;
        const int h = fpass_removed(c, fp, av, slab, it.hRpr, mutId, it.dir != 0);
        if (!fvalid(h)) { S.state = FS_FALLBACK; continue; }
        it.hRpr = h;
        const unsigned long long k = atomicAdd(&fp.ctr->nDeferred, 1ull);
        if ((long long)k < fp.capDeferred) fp.deferred[k] = (int32_t)(&it - fp.C);
        else { S.state = FS_FALLBACK; fp.ctr->overflow = 1; }
    }
}

// The one-lane updating items of the level, by direction: an item that moves down (M:6982-7160) and one that crawls up
// (M:7162-7434) share no code, and a wavefront that holds both runs the two paths one after the other -- a level lasts as long
// as its slowest wavefront.  Items moving down are listed from the front of `perm`, the others from its back.
__global__ __launch_bounds__(FR_BLOCK) void k_fr_sort_level(ArenaViewS av, DevTree T, FPools fp, int heavyMin, int bigMin)
{
    // (and by size: the items with the longest lists of the level -- the ones its slowest wavefront is made of -- are listed
    // apart, in perm2, and walked 16 to a wavefront: a wavefront runs the union of its lanes' paths for as many steps as its
    // longest list has, and a quarter of the lanes is a good deal less than the whole of that union)
    const long long lo = (long long)fp.ctr->loU, hi = (long long)fp.ctr->hiU;
    const long long n = hi - lo;
    const int lane = threadIdx.x & 63;
    heavyMin = fr_level_heavy_min(fp, heavyMin);
    auto bc = [](unsigned long long x) {
        return ((unsigned long long)(uint32_t)__shfl((int)(x >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)x, 0, 64);
    };
    for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += (long long)gridDim.x * blockDim.x) {
        const long long i = base + threadIdx.x;
        int kind = -1;                                                      // 0 down, 1 up, 2 / 3 the same with long lists, -1 a wavefront's item
        bool small = false;
        if (i < n) {
            const FItem &it = fp.U[lo + i];
            const int sz = fr_upd_size(av, T, fp, it);
            if (!(heavyMin > 0 && it.dir != 3 && sz >= heavyMin)) kind = (it.dir == 0 ? 0 : 1) + (sz >= bigMin ? 2 : 0);
            else {
                // An item no wavefront-wide walk takes -- next to a MAT reference branch (it re-expresses lists on the way), or with
                // a list beyond the 512-entry staging -- would be walked by lane 0 of a wavefront of the 512 class, of which there
                // is one per compute unit: 0.3-1.7 ms each, one after the other (a level of 32 000 items by wavefronts: 10.5 ms,
                // 8 of them for its 2 800 such items).  They stay one-lane items, 16 to a wavefront, next to each other.
                small = fr_wave_fits(av, T, fp, it, FR_WAVE_SMALL_IN, FR_WAVE_SMALL_CAPW);
                if (!small && !fr_wave_fits(av, T, fp, it, 512, 512)) kind = (it.dir == 0 ? 2 : 3);
            }
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        {   // the items that go a wavefront each (kind -1), by size class
            const bool hv = i < n && kind == -1;
            for (int cls = 0; cls < 2; cls++) {
                const bool mine = hv && (small == (cls == 0));
                const unsigned long long mh = __ballot(mine);
                if (!mh) continue;
                unsigned long long b0 = 0;
                if (lane == 0) b0 = atomicAdd(cls == 0 ? &fp.ctr->permHeavy : &fp.ctr->permHeavy2, (unsigned long long)__popcll(mh));
                b0 = bc(b0);
                if (mine) (cls == 0 ? fp.perm3 : fp.perm4)[(long long)(b0 + __popcll(mh & below))] = (int32_t)i;
            }
        }
        for (int k = 0; k < 4; k++) {
            const unsigned long long mk = __ballot(kind == k);
            if (!mk) continue;
            unsigned long long *ctr = k == 0 ? &fp.ctr->permDown : (k == 1 ? &fp.ctr->permUp : (k == 2 ? &fp.ctr->permDownB : &fp.ctr->permUpB));
            unsigned long long b0 = 0;
            if (lane == 0) b0 = atomicAdd(ctr, (unsigned long long)__popcll(mk));
            b0 = bc(b0);
            if (kind == k) {
                int32_t *pm = k < 2 ? fp.perm : fp.perm2;
                const long long at = (long long)(b0 + __popcll(mk & below));
                pm[(k & 1) ? n - 1 - at : at] = (int32_t)i;
            }
        }
    }
}

// ---- items in the cached regime (needsUpdating == False): appendProbNode(probVectTotUp[t1], removed list) ---------------
// One lane per item: the item, its search, the node record; the score (a one-lane walk of the branch's probVectTotUp against the
// removed list); the rule and the pushes.
// (Round 5, measured and not kept: the score by a GROUP of 16 lanes per item -- the two lists staged in LDS with loads that cover
// whole lines, the steps found on the merge path, the factors multiplied in walk order as wave_append does: bit-identical, no
// list word ever waited for, and SLOWER, 204-215 us per 64 items against 168: four wavefronts per SIMD then run out of issue
// slots -- a binary search per step and 16 lanes' worth of control per item cost ~7 x the instructions of the one-lane walk,
// which only ever waits.  docs/NOTES.md section 3T.)
#ifndef FR_CACHED_WAVES
#define FR_CACHED_WAVES 4             // wavefronts per SIMD the cached-regime kernel is compiled for (128 registers)
#endif
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(FR_BLOCK) __attribute__((amdgpu_waves_per_eu(FR_CACHED_WAVES, FR_CACHED_WAVES)))
void k_fr_cached(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, SearchParams P, FPools fp, int budget,
                 const int32_t *rowOf, FiniteRows fin)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x & 63;
    // the launch's items: what the cached launches before it pushed, then the roots published since
    // ... then the items of earlier launches whose removed list has been re-expressed since (deferred)
    const long long loC = (long long)fp.ctr->loC, nC = (long long)fp.ctr->hiC - loC, loR = fp.capCC + (long long)fp.ctr->loR,
                    nCR = nC + (long long)fp.ctr->hiR - (long long)fp.ctr->loR, loD = (long long)fp.ctr->loD,
                    hi = nCR + (long long)fp.ctr->hiD - loD;
    unsigned long long nSc = 0, bSc = 0;
    for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < hi; i0 += (long long)gridDim.x * blockDim.x) {
        const long long i = i0 + threadIdx.x;
#ifdef MAPLE_SPR_PROFILE
        const long long tp0 = wall_clock64();
#endif
        // ---- one lane per item: what to do with it (0 nothing more, 1 not scored: the rule and the pushes, 2 scored first)
        int mode = 0;
        FItem *itp = nullptr;
        const FSearch *Sp = nullptr;
        NodeRec r1{};
        FList lp{nullptr, nullptr, 0, 0}, lr{nullptr, nullptr, 0, 0};
        bool rt = false;
        double rbl = 0.0;
        if (i < hi) {
            FItem &it = fp.C[i < nC ? loC + i : (i < nCR ? loR + (i - nC) : (long long)fp.deferred[loD + (i - nCR)])];
            FSearch &S = fp.S[it.q];
            itp = &it; Sp = &S;
            const int st = S.state;
            if (i < nCR && (it.flags & FI_NEEDPASS)) { }                    // (pushed across a reference branch: the next launch, see FCtr::nDeferred)
            else if (!fs_live(st)) it.flags |= FI_DEAD;
            else if (st == FS_WIDE && it.dir == 0) {                        // (the clade below it: k_fr_replay_wide)
                // Most such clades hold no finite score at all for this search.  What the scan does with one of those only
                // depends on the state the walk arrives with: noted here, applied by the walk itself without a scan.
                int fl = FI_SEED;
                if (fin.mask && rowOf) {
                    const NodeRec r1s = T.nd[it.t1];
                    const SScan rec = T.scan[r1s.preRank];
                    const size_t row = (size_t)rowOf[it.q];
                    const unsigned long long *fm = fin.mask + row * fin.nWords;
                    const int32_t *fpx = fin.prefix + row * (fin.nWords + 1);
                    const int clo = T.candBefore[r1s.preRank], chi = T.candBefore[r1s.preRank + rec.size];
                    if (fin_count_before(fm, fpx, chi) == fin_count_before(fm, fpx, clo)) {
                        const bool firstScored = !(r1s.up == S.parent || r1s.up < 0) && (r1s.dist > P.effNon0 || r1s.upIsRoot);
                        const bool dropped = firstScored && !(rec.ff & SS_TOTUP);
                        fl |= FI_SEED_EMPTY;
                        it.hA = (firstScored && !dropped) ? 1 : 0;           // the clade's root counts as one placement
                        it.hB = T.cladeVisits[r1s.preRank];                  // placements below it when it is descended into
                        it.hMid = ((rec.ff & SS_INNER) && !dropped) ? 1 : 0;
                    }
                }
                it.flags |= (uint8_t)fl;
            } else {
                r1 = T.nd[it.t1];
                const int upT = r1.up;
                const bool scored = (it.dir == 0) ? (!(upT == S.parent || upT < 0) && (r1.dist > P.effNon0 || r1.upIsRoot))
                                                  : (upT >= 0 && (r1.dist > P.effNon0 || r1.upIsRoot));
                mode = 1;
                if (scored) {
                    if (r1.totUp < 0) { it.flags |= FI_DEAD; mode = 0; }
                    else {
                        lp = flist(av, fp, ftree(r1.totUp)); lr = flist(av, fp, it.hRpr);
                        rt = S.isRemovedTip != 0; rbl = S.removedBLen;
                        mode = 2;
                    }
                }
            }
        }
#ifdef MAPLE_SPR_PROFILE
        __builtin_amdgcn_s_waitcnt(0);
        const long long tp1 = wall_clock64();
#endif
        // ---- the score
        double scoreV = 0.0;
        if (mode == 2) scoreV = append_walk(c, fref(lp), fref(lr), rt, rbl);
        const int nS = __popcll(__ballot(mode == 2));
        (void)nS;
#ifdef MAPLE_SPR_PROFILE
        const long long tp2 = wall_clock64();
#endif
        // ---- one lane per item again: the rule (M:7090-7103 / 7311-7323 in its permissive form) and the pushes
        if (mode != 0) {
            FItem &it = *itp;
            const FSearch &S = *Sp;
            const int q = it.q, hRpr = it.hRpr, upT = r1.up;
            const double lastLK = it.lastLK;
            double midProb = lastLK;
            const bool scored = mode == 2;
            if (scored) {
                midProb = scoreV;
                it.flags |= FI_SCORED;
                nSc++; bSc += 8ull * (unsigned long long)(lp.n + lp.na) + 8ull;
            }
            it.midProb = midProb;
            const PRule pr = p_rule(P, scored, midProb, lastLK, it.failsP, it.pathBest);
            if (pr.go) {
                // (a relative in another MAT reference frame gets the removed list through the branch at the start of its level: k_fr_pass)
                const bool x0 = fp.mat && r1.c0Frame != r1.frameOf, x1 = fp.mat && r1.c1Frame != r1.frameOf, xu = fp.mat && r1.upFrame != r1.frameOf;
                if (it.dir == 0) {
                    if (r1.c0 >= 0) {
                        if (r1.upRight >= 0) it.child0 = fpush(fp, budget, q, false, r1.c0, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, x0, true);
                        if (r1.upLeft >= 0) it.child1 = fpush(fp, budget, q, false, r1.c1, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, x1, true);
                    }
                } else {
                    const int other = (it.dir == 1) ? r1.c1 : r1.c0;
                    const bool xo = (it.dir == 1) ? x1 : x0;
                    if (upT >= 0) {
                        if (((it.dir == 1) ? r1.upLeft : r1.upRight) >= 0) {
                            it.child0 = fpush(fp, budget, q, false, other, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, xo, true);
                            it.child1 = fpush(fp, budget, q, false, upT, (int)r1.whichChild + 1, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, xu, true);
                        }
                    } else
                        it.child0 = fpush(fp, budget, q, false, other, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, xo, true);
                }
            }
            (void)S;
        }
#ifdef MAPLE_SPR_PROFILE
        {
            __builtin_amdgcn_s_waitcnt(0);
            const long long tp3 = wall_clock64();
            int len = mode == 2 ? lp.n + lr.n : 0, tot = len;
            for (int off = 32; off > 0; off >>= 1) { len = max(len, __shfl_down(len, off, 64)); tot += __shfl_down(tot, off, 64); }
            if (lane == 0) {
                atomicAdd(&fp.ctr->dbgC[0], 1ull); atomicAdd(&fp.ctr->dbgC[1], (unsigned long long)(tp1 - tp0));
                atomicAdd(&fp.ctr->dbgC[2], (unsigned long long)(tp2 - tp1)); atomicAdd(&fp.ctr->dbgC[3], (unsigned long long)(tp3 - tp2));
                atomicAdd(&fp.ctr->dbgC[4], (unsigned long long)len); atomicAdd(&fp.ctr->dbgC[5], (unsigned long long)nS);
                atomicAdd(&fp.ctr->dbgC[6], (unsigned long long)tot);
            }
        }
#endif
    }
    // (what the launch scored, for the roofline of the bench line: one atomic per wavefront)
    for (int off = 32; off > 0; off >>= 1) {
        nSc += ((unsigned long long)(uint32_t)__shfl_down((int)(nSc >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)nSc, off, 64);
        bSc += ((unsigned long long)(uint32_t)__shfl_down((int)(bSc >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)bSc, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && nSc) { atomicAdd(&fp.ctr->scoredC, nSc); atomicAdd(&fp.ctr->bytesC, bSc); }
}

// ---- the reference's own walk over the expanded items: "while nodesToVisit", M:6964-7434, with the real running best ------
// ---- the visiting-order layout ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long fidx(const FPools &fp, int ref) { return ref >= 0 ? fp.capU + ref : -(long long)(ref + 2); }
__device__ __forceinline__ int fsize(const FPools &fp, int ref) { return ref == FR_NONE ? 0 : fp.lsize[fidx(fp, ref)]; }

// subtree sizes, one level at a time from the last to the first (a child is one level below its parent)
// (sizes, ranks and parents live in arrays of their own: the passes read one 32-byte sector of an item and 4-byte neighbours)
// (which = 0: level `level` of the list-updating items; 1: launch `level` of the cached-regime items -- its two ranges.  A child of an
// updating item is in the next updating level or in some cached launch; a child of a cached item is in a LATER cached launch:
// bottom-up = the cached launches from the last to the first, then the updating levels from the last to the first)
__device__ __forceinline__ void fr_layer(const FPools &fp, int level, int which, long long &lo1, long long &n1, long long &lo2, long long &n2)
{
    const unsigned long long *L = fp.lvl + (size_t)FR_LVL * level;
    if (which == 0) { lo1 = (long long)L[0]; n1 = (long long)L[1] - lo1; lo2 = 0; n2 = 0; }
    else { lo1 = (long long)L[2]; n1 = (long long)L[3] - lo1; lo2 = (long long)L[4]; n2 = (long long)L[5] - lo2; }
}
__global__ __launch_bounds__(FR_BLOCK) void k_fr_layout_sizes(FPools fp, int level, int which)
{
    long long lo1, n1, lo2, n2;
    fr_layer(fp, level, which, lo1, n1, lo2, n2);
    const long long n = n1 + n2;
    const bool isU = which == 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long at = i < n1 ? lo1 + i : lo2 + (i - n1);
        const FItem &it = isU ? fp.U[at] : fp.C[at];
        fp.lsize[isU ? at : fp.capU + at] = 1 + fsize(fp, it.child0) + fsize(fp, it.child1);
    }
}
__global__ __launch_bounds__(FR_BLOCK) void k_fr_layout_totals(int n, FPools fp)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const FSearch &S = fp.S[q];
        fp.tot[q] = S.state == FS_ACTIVE ? fsize(fp, S.seed0) + fsize(fp, S.seed1) : 0;
    }
}
// exclusive prefix sums of n counts by ONE workgroup: out[0 .. n], out[n] = the total
__global__ __launch_bounds__(1024) void k_fr_layout_scan(int n, const int32_t *in, long long *out)
{
    __shared__ long long part[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; i++) s += in[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; i++) { const long long v = part[i]; part[i] = run; run += v; }
        out[n] = run;
    }
    __syncthreads();
    long long run = part[t];
    for (int i = lo; i < hi; i++) { out[i] = run; run += in[i]; }
}
// the seeds' ranks (the one pushed last, seed1, is visited first)
__global__ __launch_bounds__(FR_BLOCK) void k_fr_layout_seeds(int n, FPools fp)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const FSearch &S = fp.S[q];
        if (S.state != FS_ACTIVE || fp.vbase[q + 1] > fp.capVisit) continue;
        long long p = fp.vbase[q];
        if (S.seed1 != FR_NONE) { const long long x = fidx(fp, S.seed1); fp.lpos[x] = (int32_t)p; fp.lpar[x] = -1; p += fp.lsize[x]; }
        if (S.seed0 != FR_NONE) { const long long x = fidx(fp, S.seed0); fp.lpos[x] = (int32_t)p; fp.lpar[x] = -1; }
    }
}
// ranks of the children and the item's own record, one level at a time from the first to the last
__global__ __launch_bounds__(FR_BLOCK) void k_fr_layout_place(FPools fp, int level, int which)
{
    long long lo1, n1, lo2, n2;
    fr_layer(fp, level, which, lo1, n1, lo2, n2);
    const long long n = n1 + n2;
    const bool isU = which == 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long at = i < n1 ? lo1 + i : lo2 + (i - n1);
        const long long own = isU ? at : fp.capU + at;
        const int p = fp.lpos[own];
        if (p < 0) continue;                                                // (its search is not laid out)
        const FItem &it = isU ? fp.U[at] : fp.C[at];
        FVisit v;
        v.midProb = it.midProb; v.lastLK = it.lastLK; v.ref = isU ? -((int)at + 2) : (int)at; v.size = fp.lsize[own];
        v.parent = fp.lpar[own]; v.flags = it.flags; v.dir = it.dir; v.failsOut = 0;
        fp.visit[p] = v;
        int q = p + 1;
        if (it.child1 != FR_NONE) { const long long x = fidx(fp, it.child1); fp.lpos[x] = q; fp.lpar[x] = p; q += fp.lsize[x]; }
        if (it.child0 != FR_NONE) { const long long x = fidx(fp, it.child0); fp.lpos[x] = q; fp.lpar[x] = p; }
    }
}

// what the exact walk of one search leaves behind: the short list, the candidate count, the records to refine
// (hShort: the removed lists the reference shortened in place on the way, M:7087 -- see k_fr_replay)
__device__ inline void fr_replay_done(const SearchParams &P, const FPools &fp, SearchOut *out, int q, FSearch &S, int slHead, int nApp, bool handBack,
                                      const int *hShort = nullptr, int nShort = 0, int reason = 0)
{
    // (the list the search starts from, bestRemovedPartials when nothing better is found, shortened: the one-lane kernel)
    for (int k = 0; k < nShort; k++) if (hShort[k] == S.hRpr0 && !handBack) { handBack = true; reason = 4; }
    if (handBack) { S.state = FS_FALLBACK; out[q].status = FR_STATUS_FALLBACK; atomicAdd(&fp.ctr->fbReason[reason & 7], 1); return; }
    S.slHead = slHead; S.nApp = nApp;
    // the short-listed branches that are refined (M:7465: within thresholdLogLKoptimizationTopology of the ORIGINAL cost)
    int cnt = 0;
    for (int r = slHead; r != FR_NONE; r = item_of(fp, r).next)
        if (item_of(fp, r).midProb >= S.curLK - P.thrOptTopo) cnt++;
    const unsigned long long base = atomicAdd(&fp.ctr->nRecs, (unsigned long long)cnt);
    if ((long long)(base + cnt) > fp.capRecs) {
        S.state = FS_FALLBACK; out[q].status = FR_STATUS_FALLBACK; fp.ctr->overflow = 1;
        for (long long k2 = (long long)base; k2 < fp.capRecs; k2++) { fp.recs[k2].q = q; fp.recs[k2].ref = FR_NONE; }   // (skipped by its state)
        return;
    }
    S.recBase = (int32_t)base; S.recCount = cnt;
    int k = 0;
    for (int r = slHead; r != FR_NONE; r = item_of(fp, r).next)
        if (item_of(fp, r).midProb >= S.curLK - P.thrOptTopo) {
            FRec &x = fp.recs[base + k++];
            x.q = q; x.ref = r; x.ok = 0; x.hRprS = -1;
            // (the short list holds the list OBJECT: shortened in place at any time of the walk, it is the shortened list by the
            // time the refinement reads it)
            const int hr = item_of(fp, r).hRpr;
            for (int j = 0; j < nShort; j++) if (hShort[j] == hr) x.hRprS = FR_SHORTEN;
        }
}

__global__ __launch_bounds__(FR_BLOCK) void k_fr_replay(SearchParams P, int n, FPools fp, SearchOut *out)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        FSearch &S = fp.S[q];
        if (S.state == FS_OVER) { out[q].status = -5; continue; }
        if (S.state == FS_FALLBACK) { out[q].status = FR_STATUS_FALLBACK; continue; }
        if (S.state != FS_ACTIVE) continue;
        double best = S.curLK;
        int nApp = 0, slHead = FR_NONE, slTail = FR_NONE;
        bool marked = false;
        int why = 0;                                                        // 1 a merge within tolerance, 2 a list re-expressed from a shortened one, 3 a fifth list, 4 the search's own first list, 5 no layout
        // The reference shortens a branch's removed list IN PLACE when the branch beats the running best on the way down (M:7087).
        // A list that shorten() would change is marked by the expansion (shorten_would_merge): level 2 -- tails merged within a
        // tolerance: another list to every reader -- hands the search to the one-lane kernel.  Level 1 -- the entries that go
        // away have exactly the tail of the one that stays -- changes no score; what it does change is emulated: the list is
        // noted here (with the rank of the event), the records that hold it are refined with its shortened form (fr_replay_done,
        // k_fr_refine), and a search in which a list is re-expressed FROM it after the event (an item that crosses a reference
        // branch, pushed by an item visited later than the event) goes to the one-lane kernel after all.
        int hShort[4], nShort = 0;
        long long rankShort[4];
        if (fp.visit && fp.vbase[q + 1] <= fp.capVisit) {
            // the same walk as a forward scan over the search's items in visiting order
            const long long end = fp.vbase[q + 1];
            long long i = fp.vbase[q];
            // (failedPasses by rank in an array of its own -- lpos, which the layout is done with: a store into the records would
            // throw their cache line out of L1 under the scan, and every record would come from L2 again)
            int32_t *const failsAt = fp.lpos;
            // The records are read EIGHT at a time (independent loads, one wait) and worked through in order while the walk goes to
            // the record that follows; a skip over a subtree starts the next block at the place it lands.  (One record per
            // iteration, asked for two iterations ahead: ~0.5 us per record for the longest search of a round -- a 128-byte line
            // holds four records and the next line was a miss every time: 7.8 ms.)
            bool stop = false;
            while (i < end && !stop) {
                constexpr int BLK = 8;
                FVisit w[BLK];
#pragma unroll
                for (int k = 0; k < BLK; k++) w[k] = fp.visit[i + k < end ? i + k : end - 1];
                const long long i0 = i;
#pragma unroll
                for (int k = 0; k < BLK; k++) {
                    if (stop || i != i0 + k || i >= end) continue;
                    const FVisit rec = w[k];
                    const int size = rec.size;
                    long long step = size;
                    if (!(rec.flags & FI_DEAD)) {
                        int fails = rec.parent < 0 ? 0 : failsAt[rec.parent];
                        const double mp = rec.midProb;
                        if (nShort && rec.parent >= 0) {                            // (after an event only: two item records per visit)
                            const int hr = item_of(fp, rec.ref).hRpr, hp = item_of(fp, fp.visit[rec.parent].ref).hRpr;
                            if (hr != hp)
                                for (int j = 0; j < nShort; j++) if (hShort[j] == hp && (long long)rec.parent >= rankShort[j]) marked = true;   // (>=: the event item shortens at its own visit, BEFORE it pushes its children, M:7087 / 7116-7118)
                            if (marked) { stop = true; why = 2; }
                        }
                        if (rec.flags & FI_SCORED) {
                            nApp++;
                            const bool list = (rec.dir == 0) ? (mp > best - P.thrOptTopo) : (mp >= best - P.thrOptTopo);   // M:7071 / 7293
                            if (list) {
                                const int ref = rec.ref;
                                item_of(fp, ref).next = FR_NONE;
                                if (slTail == FR_NONE) slHead = ref; else item_of(fp, slTail).next = ref;
                                slTail = ref;
                            }
                            if (mp > best) {
                                best = mp; fails = 0;
                                // (M:7087: the reference shortens the branch's removed list in place here; if that changes the list, the
                                // one-lane kernel takes the search)
                                if (rec.dir == 0) {
                                    const int hr = item_of(fp, rec.ref).hRpr, lvl = frpr_marked(fp, S, hr);
                                    if (lvl == 2) { marked = true; stop = true; why = 1; }
                                    if (lvl == 1) {
                                        bool known = false;
                                        for (int j = 0; j < nShort; j++) known |= hShort[j] == hr;
                                        if (!known) {
                                            if (nShort == 4) { marked = true; stop = true; why = 3; }
                                            else { hShort[nShort] = hr; rankShort[nShort] = i; nShort++; }
                                        }
                                    }
                                }
                            }
                            else if (mp < (rec.lastLK - P.thrConsec)) fails++;
                        }
                        const bool within = mp > (best - P.thrLKtopology);
                        const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);
                        if (go && size > 1) failsAt[i] = fails;                    // (read by the item's children only)
                        if (go) step = 1;
                    }
                    i += step;
                }
            }
        } else {
        int top = FR_NONE;
        auto push = [&](int ref, int fails) {
            if (ref == FR_NONE) return;
            FItem &x = item_of(fp, ref);
            x.next = top; x.failsA = (int16_t)fails; top = ref;
        };
        push(S.seed0, 0);
        push(S.seed1, 0);
        while (top != FR_NONE) {
            const int ref = top;
            FItem &it = item_of(fp, ref);
            top = it.next;
            if (it.flags & FI_DEAD) continue;
            int fails = it.failsA;
            const double mp = it.midProb;
            if (it.flags & FI_SCORED) {
                nApp++;
                const bool list = (it.dir == 0) ? (mp > best - P.thrOptTopo) : (mp >= best - P.thrOptTopo);   // M:7071 / 7293
                if (list) {
                    it.next = FR_NONE;
                    if (slTail == FR_NONE) slHead = ref; else item_of(fp, slTail).next = ref;
                    slTail = ref;
                }
                if (mp > best) {
                    best = mp; fails = 0;
                    if (it.dir == 0 && frpr_marked(fp, S, it.hRpr)) { marked = true; why = 5; break; }   // (M:7087, see above)
                }
                else if (mp < (it.lastLK - P.thrConsec)) fails++;
            }
            const bool within = mp > (best - P.thrLKtopology);
            const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);
            if (!go) continue;
            push(it.child0, fails);
            push(it.child1, fails);
        }
        }
        fr_replay_done(P, fp, out, q, S, slHead, nApp, marked, hShort, nShort, why);
    }
}

// ---- the same walk for a whole-tree search (FS_WIDE), one wavefront per search ------------------------------------------
// Lane 0 pops items like k_fr_replay; a seed (an item that arrived in the cached regime on the way down) hands the clade below
// it to all 64 lanes: wave_scan_clade (search_dev.h) applies the reference's rules to the clade in pop order over the search's
// row of the dense score table -- what the LIFO stack would do with the item and everything it pushes, a pushed clade being
// finished before anything older is popped.  The short list is kept in visiting order in `br` (items: hUp = item ref, hDown =
// WR_ITEM; scan entries: the node); the scan's entries that are refined become items of the cached pool so that k_fr_refine
// and k_fr_finish treat them like any other.
#define WR_ITEM (-77)
__global__ __launch_bounds__(64) void k_fr_replay_wide(DevTree T, SearchParams P, int nWide, const int32_t *wideQ, const int32_t *rowOf,
                                                       const double *cacheS, FiniteRows fin, FPools fp, SearchOut *out,
                                                       BestRec *brAll, int capB, int *counter)
{
    extern __shared__ double dynW[];
    double *slotLK = dynW;
    int *slotFails = (int *)(dynW + T.scanDepthCap);
    unsigned *slotOwner = (unsigned *)(slotFails + T.scanDepthCap);
    const int lane = threadIdx.x;
    BestRec *br = brAll + (size_t)blockIdx.x * capB;
    for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(counter, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= nWide) break;
        const int q = wideQ[k];
        FSearch &S = fp.S[q];
        if (S.state != FS_WIDE) continue;
        const size_t row = (size_t)rowOf[q];
        const double *cs = cacheS + row * (size_t)T.n;
        const unsigned long long *fm = fin.mask ? fin.mask + row * fin.nWords : nullptr;
        const int32_t *fpx = fin.mask ? fin.prefix + row * (fin.nWords + 1) : nullptr;
        int top = FR_NONE;
        auto push = [&](int ref, int fails) {
            if (ref == FR_NONE) return;
            FItem &x = item_of(fp, ref);
            x.next = top; x.failsA = (int16_t)fails; top = ref;
        };
        double best = S.curLK;
        int nApp = 0, nB = 0, bad = 0;
        if (lane == 0) { push(S.seed0, 0); push(S.seed1, 0); }
#ifdef MAPLE_SPR_PROFILE
        long long tWalk = 0, tScan = 0, nScan = 0, tw0 = wall_clock64();
#endif
        for (;;) {
            int seed = FR_NONE, seedFails = 0;
#ifdef MAPLE_SPR_PROFILE
            tw0 = wall_clock64();
#endif
            if (lane == 0) {
                while (top != FR_NONE) {
                    const int ref = top;
                    FItem &it = item_of(fp, ref);
                    top = it.next;
                    if (it.flags & FI_DEAD) continue;
                    int fails = it.failsA;
                    if (it.flags & FI_SEED) {
                        const double hMp = it.hA ? -INFINITY : it.lastLK;       // what the clade's root hands on
                        if ((it.flags & FI_SEED_EMPTY) && hMp == -INFINITY) {
                            // a clade without a finite score, entered with -inf: counted, not walked (wave_scan_clade's skipClade)
                            nApp += it.hA;
                            if (it.hA && hMp < (it.lastLK - P.thrConsec)) fails++;
                            if (!P.strict && fails <= P.allowedFails && it.hMid) nApp += it.hB;
                            continue;
                        }
                        seed = ref; seedFails = fails; break;
                    }
                    const double mp = it.midProb;
                    if (it.flags & FI_SCORED) {
                        nApp++;
                        const bool list = (it.dir == 0) ? (mp > best - P.thrOptTopo) : (mp >= best - P.thrOptTopo);   // M:7071 / 7293
                        if (list) {
                            if (nB >= capB) { bad = 1; break; }
                            br[nB++] = BestRec{it.t1, ref, WR_ITEM, -1, it.hRpr, mp, 0.0};
                        }
                        if (mp > best) {
                            best = mp; fails = 0;
                            if (it.dir == 0 && frpr_marked(fp, S, it.hRpr)) { bad = 1; break; }   // (M:7087: see k_fr_replay)
                        }
                        else if (mp < (it.lastLK - P.thrConsec)) fails++;
                    }
                    const bool within = mp > (best - P.thrLKtopology);
                    const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);
                    if (!go) continue;
                    push(it.child0, fails);
                    push(it.child1, fails);
                }
            }
            seed = __builtin_amdgcn_readfirstlane(seed);
            bad = __builtin_amdgcn_readfirstlane(bad);
#ifdef MAPLE_SPR_PROFILE
            tWalk += wall_clock64() - tw0; tw0 = wall_clock64(); nScan++;
#endif
            if (seed == FR_NONE || bad) break;
            seedFails = __builtin_amdgcn_readfirstlane(seedFails);
            const FItem &it = fp.C[seed];
            const NodeRec r1 = T.nd[it.t1];
            const bool firstScored = !(r1.up == S.parent || r1.up < 0) && (r1.dist > P.effNon0 || r1.upIsRoot);
            ScanState st;
            st.best = readfirst_f64(best);
            const double bestBefore = st.best;
            st.nB = __builtin_amdgcn_readfirstlane(nB);
            st.nApp = __builtin_amdgcn_readfirstlane(nApp);
            st.overflow = 0;
            st.shortenSeed = false;
            st.fShort[0] = st.fShort[1] = st.fShort[2] = st.fShort[3] = -1;
            __threadfence();                                                // (lane 0's short-list entries before the scan's)
            wave_scan_clade(T.scan, T.scanParent, cs, nullptr, r1.preRank, firstScored, r1.frameOf, it.hRpr, it.lastLK, seedFails, P, br, capB,
                            slotLK, slotFails, slotOwner, T.scanDepthCap, st, fm, fpx, T.candBefore, T.cladeVisits);
            if (st.overflow) { bad = 1; break; }
            // (a tree with local references: a branch of the clade that beats the running best would have its removed list -- the
            // seed's, re-expressed in the branch's frame, which only exists after the scan -- shortened in place, M:7087.  That is a
            // no-op unless shorten() would merge entries of THAT list: looked at when the list exists -- below for a branch in the
            // seed's own frame, in k_fr_wide_frames for the others (FR_FRAMES_EVENT) -- instead of handing back every such search:
            // one of them is 100 ms of one wavefront at 1 000 000 tips)
            (void)bestBefore;
            best = st.best; nB = st.nB; nApp = st.nApp;
            __threadfence();
#ifdef MAPLE_SPR_PROFILE
            tScan += wall_clock64() - tw0;
#endif
        }
        nB = __builtin_amdgcn_readfirstlane(nB);
        nApp = __builtin_amdgcn_readfirstlane(nApp);
        __builtin_amdgcn_wave_barrier();
        __threadfence();                                                    // (the scan's short-list entries, written by other lanes)
#ifdef MAPLE_SPR_PROFILE
        if (lane == 0) { out[q].tStep = tWalk; out[q].tReplay = tScan; out[q].nSteps = (int32_t)nScan; }
#endif
        if (lane == 0) {
            // the short-listed branches that are refined (M:7465), in visiting order; the scan's become cached-pool items
            int cnt = 0, cntScan = 0;
            int why = bad ? 6 : 0;
            if (fp.mat && !bad) {
                // the branches that beat the running best when they were visited (all of them are short-listed), in visiting order
                double rb = S.curLK;
                for (int i = 0; i < nB; i++) {
                    const BestRec b = br[i];
                    if (!(b.score > rb)) continue;
                    rb = b.score;
                    if (b.hDown != WR_ITEM && b.hMid == b.hDown && frpr_marked(fp, S, b.hRpr)) { bad = 1; why = 7; break; }
                }
            }
            for (int i = 0; i < nB && !bad; i++)
                if (br[i].score >= S.curLK - P.thrOptTopo) { cnt++; if (br[i].hDown != WR_ITEM) cntScan++; }
            unsigned long long base = 0ull, ibase = 0ull;
            if (!bad) {
                base = atomicAdd(&fp.ctr->nRecs, (unsigned long long)cnt);
                ibase = cntScan ? atomicAdd(&fp.ctr->usedC, (unsigned long long)cntScan) : 0ull;
                if ((long long)(base + cnt) > fp.capRecs || (long long)(ibase + cntScan) > fp.capCC) {
                    bad = 1; fp.ctr->overflow = 1;
                    for (long long k2 = (long long)base; k2 < min((long long)(base + cnt), fp.capRecs); k2++) { fp.recs[k2].q = q; fp.recs[k2].ref = FR_NONE; }
                }
            }
            if (bad) { S.state = FS_FALLBACK; atomicAdd(&fp.ctr->fbReason[why ? why : 6], 1); }   // (short list, depth slots or pools too small, a list shorten() would change: the one-lane kernel)
            else {
                S.recBase = (int32_t)base; S.recCount = cnt; S.nApp = nApp;
                int kk = 0, ks = 0;
                double rb = S.curLK;
                for (int i = 0; i < nB; i++) {
                    const BestRec b = br[i];
                    const bool event = b.score > rb;
                    if (event) rb = b.score;
                    if (!(b.score >= S.curLK - P.thrOptTopo)) continue;
                    int ref = b.hUp;
                    bool frames = false;
                    if (b.hDown != WR_ITEM) {
                        ref = (int)(ibase + ks++);
                        FItem &x = fp.C[ref];
                        x.q = q; x.t1 = b.t1; x.dir = 0; x.flags = FI_SCORED; x.failsP = 0; x.hPassed = -1; x.hRpr = b.hRpr;
                        x.distance = 0.0; x.lastLK = b.score; x.pathBest = b.score; x.midProb = b.score; x.recDist = 0.0;
                        x.child0 = x.child1 = FR_NONE; x.hA = x.hB = x.hMid = -1; x.next = FR_NONE; x.failsA = 0;
                        // (the scan's removed list is its seed's: a branch in a frame nested below the seed's gets it through the
                        // reference branches in between before it is refined -- k_fr_wide_frames)
                        if (fp.mat && b.hMid != b.hDown) { frames = true; x.hA = b.hDown; x.hB = b.hMid; }
                    }
                    FRec &x = fp.recs[base + kk++];
                    x.q = q; x.ref = ref; x.ok = 0; x.hRprS = frames ? (event ? FR_FRAMES_EVENT : FR_FRAMES) : -1;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- trees with MAT local references: the removed list of a short-listed branch the clade scan found in a frame nested below its
// seed's: down the reference branches from the seed's frame (x.hA) to the branch's (x.hB), outermost first (M:7111-7118)
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(FR_BLOCK) void k_fr_wide_frames(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, FPools fp,
                                                             const int32_t *frameParent, const int32_t *frameNode)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const long long laneId = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nRecs = min((long long)fp.ctr->nRecs, fp.capRecs);
    for (long long i = laneId; i < nRecs; i += (long long)gridDim.x * blockDim.x) {
        FRec &R = fp.recs[i];
        if (R.ref < 0 || (R.hRprS != FR_FRAMES && R.hRprS != FR_FRAMES_EVENT)) continue;   // (only the scan's entries: cached-pool items)
        FItem &x = fp.C[R.ref];
        FSearch &S = fp.S[R.q];
        if (!fs_live(S.state)) continue;
        int chain[24], n = 0;
        for (int g = x.hB; g != x.hA && g > 0 && n < 24; g = frameParent[g]) chain[n++] = g;
        int h = x.hRpr;
        bool ok = n < 24;
        for (int k = n - 1; k >= 0 && ok; k--) {
            h = fpass_removed(c, fp, av, laneId, h, T.nd[frameNode[chain[k]]].mutId, false);
            ok = fvalid(h);
        }
        if (!ok) { S.state = FS_FALLBACK; continue; }
        // (the branch beat the running best when the scan visited it: the reference shortened THIS list in place there, M:7087 --
        // nothing to emulate unless that would merge entries)
        if (R.hRprS == FR_FRAMES_EVENT && frpr_marked(fp, S, h)) { S.state = FS_FALLBACK; atomicAdd(&fp.ctr->fbReason[7], 1); continue; }
        x.hRpr = h; x.hA = x.hB = -1;
        R.hRprS = -1;
    }
}

// ---- refinement of one short-listed branch, M:7460-7639 (evaluatePlacement M:6790-6806 and the two compensation appends) --
template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(FR_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_fr_refine(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, FPools fp)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const long long laneId = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double *ais = fp.sais + laneId * 2ll * fp.capE;
    const long long nRecs = min((long long)fp.ctr->nRecs, fp.capRecs);
    for (long long i = laneId; i < nRecs; i += (long long)gridDim.x * blockDim.x) {
        FRec &R = fp.recs[i];
        const FItem &it = item_of(fp, R.ref);
        FSearch &S = fp.S[R.q];
        R.ok = 0;
        if (!fs_live(S.state)) continue;
        const int t1 = it.t1;
        const NodeRec r1 = T.nd[t1];
        int hUp, hDown, hMid;
        double distance;
        if (it.flags & FI_REC_UPD) { hUp = it.hA; hDown = it.hB; hMid = it.hMid; distance = it.recDist; }
        else {
            hUp = r1.up >= 0 ? ftree(r1.whichChild ? T.nd[r1.up].upLeft : T.nd[r1.up].upRight) : -1;
            if (fp.mat && fvalid(hUp)) {                                    // the parent's upper list in t1's own frame (M:7469-7470)
                hUp = fpass_store(fp, av, c.m.lRef, laneId, hUp, r1.mutId, false);
                if (hUp == -2) { S.state = FS_FALLBACK; continue; }
            }
            hDown = ftree(r1.lower); hMid = ftree(r1.totUp); distance = r1.dist;
        }
        if (!fvalid(hUp) || !fvalid(hDown) || !fvalid(hMid)) { R.ok = -1; continue; }
        int hRem = it.hRpr;
        if (R.hRprS == FR_SHORTEN) {                                        // the reference shortened this list in place on its way (M:7087)
            const FList l0 = flist(av, fp, hRem);
            FScr scrS;
            if (!fscratch(fp, laneId, l0.n, scrS)) { S.state = FS_FALLBACK; continue; }
            Writer ws;
            ws.init(scrS.w, scrS.a);
            shorten_walk(c, fref(l0), l0.n, ws);
            hRem = fstore(fp, ws);
            if (hRem < 0) { S.state = FS_FALLBACK; continue; }
            R.hRprS = hRem;
        }
        const FList lUp = flist(av, fp, hUp), lDown = flist(av, fp, hDown), lMid = flist(av, fp, hMid), lRem = flist(av, fp, hRem);
        const bool ft = r1.isTip != 0, rt = S.isRemovedTip != 0;
        const int need = max(lDown.n + lRem.n, max(lUp.n + lRem.n, lUp.n + lDown.n));
        FScr scr0;
        if (lMid.n + lRem.n > 2 * fp.capE || lUp.n + need > 2 * fp.capE || !fscratch(fp, laneId, need, scr0)) { S.state = FS_FALLBACK; continue; }
        uint2 *sw = scr0.w;
        double *sa = scr0.a;
        bool f;
        Writer w;
        const double app = blen_walk(c, fref(lMid), fref(lRem), rt, ais, 1, &f);
        w.init(sw, sa);
        int r = merge_walk(c, fref(lDown), distance / 2, ft, fref(lRem), app, rt, false, false, 0, 0, w, nullptr);
        if (r < 0) { R.ok = -1; continue; }                                 // the reference fails here (caught by the worker)
        const ListRef scr{sw, sa};
        double top = blen_walk(c, fref(lUp), scr, false, ais, 1, &f);
        w.init(sw, sa);
        r = merge_walk(c, fref(lUp), top, false, fref(lRem), app, rt, true, false, 0, 0, w, nullptr);
        if (r == -1) {
            top = m.defaultBLen * 0.1;
            w.init(sw, sa);
            r = merge_walk(c, fref(lUp), top, false, fref(lRem), app, rt, true, false, 0, 0, w, nullptr);
        }
        if (r < 0) { R.ok = -1; continue; }
        const double bottom = blen_walk(c, scr, fref(lDown), ft, ais, 1, &f);
        w.init(sw, sa);
        r = merge_walk(c, fref(lUp), top, false, fref(lDown), bottom, ft, true, false, 0, 0, w, nullptr);
        if (r < 0) { R.ok = -1; continue; }
        const double cost = append_walk(c, scr, fref(lRem), rt, app);
        const double initialCost = append_walk(c, fref(lUp), fref(lDown), ft, distance);
        const double newPartialCost = append_walk(c, fref(lUp), fref(lDown), ft, bottom + top);
        R.optimized = cost + newPartialCost - initialCost;
        R.top = top; R.bottom = bottom; R.app = app;
        R.ok = 1;
    }
}

// ---- final selection (M:7635), the worker's accept rule and vetoes (M:9681-9700), bestRemovedPartials --------------------
__global__ __launch_bounds__(FR_BLOCK) void k_fr_finish(ArenaViewS av, DevTree T, SearchParams P, int n, FPools fp, SearchOut *out,
                                                        uint2 *poolW, double *poolA, unsigned long long *poolUsed, long long poolCapW,
                                                        long long poolCapA)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        FSearch &S = fp.S[q];
        SearchOut &o = out[q];
        if (S.state == FS_FALLBACK) { o.status = FR_STATUS_FALLBACK; continue; }
        if (!fs_live(S.state)) continue;
        const int node = S.node;
        const NodeRec rp = T.nd[S.parent];
        double bestScore = S.curLK;
        int bestNode = S.sibling, hBestRpr = S.hRpr0, nApp = S.nApp;
        double bl0 = rp.up >= 0 ? rp.dist : 0.0, bl1 = T.nd[S.sibling].dist, bl2 = S.removedBLen;
        bool failed = false;
        for (int k = 0; k < S.recCount; k++) {
            const FRec &R = fp.recs[S.recBase + k];
            if (R.ok != 1) { failed = true; break; }
            nApp += 3;
            if (R.optimized >= bestScore) {
                const FItem &it = item_of(fp, R.ref);
                bestNode = it.t1; bestScore = R.optimized; bl0 = R.top; bl1 = R.bottom; bl2 = R.app; hBestRpr = R.hRprS >= 0 ? R.hRprS : it.hRpr;
            }
        }
        o.nAppend = nApp;
        o.nShortList = S.recCount;
        if (failed) { o.status = -1; continue; }                            // the reference raises here; its worker swallows it (M:9703)
        o.bestNode = bestNode; o.bestScore = bestScore;
        o.blen[0] = bl0; o.blen[1] = bl1; o.blen[2] = bl2;
        if (poolW) {
            const FList rp2 = flist(av, fp, hBestRpr);
            const long long ow = (long long)atomicAdd(&poolUsed[0], (unsigned long long)rp2.n);
            const long long oa = (long long)atomicAdd(&poolUsed[1], (unsigned long long)rp2.na);
            if (ow + rp2.n <= poolCapW && oa + rp2.na <= poolCapA) {
                for (int k = 0; k < rp2.n; k++) poolW[ow + k] = rp2.w[k];
                for (int k = 0; k < rp2.na; k++) poolA[oa + k] = rp2.aux[k];
                o.rprWoff = ow; o.rprAoff = oa; o.rprN = rp2.n; o.rprNA = rp2.na;
            } else o.status = -4;
        }
        const double curLK = S.curLK;
        if (bestScore + P.thrPlacement > curLK) {
            bool updated = true;
            int topNode = T.nd[node].up;
            if (bestNode == topNode) updated = false;
            while (T.nd[topNode].dist == 0.0 && T.nd[topNode].up >= 0) topNode = T.nd[topNode].up;
            if (bestNode == topNode && bl1 == 0.0) updated = false;
            if (bestNode == S.sibling) updated = false;
            if (T.nd[bestNode].up == S.sibling && bl0 == 0.0) updated = false;
            if (updated) { o.improvement = bestScore - curLK; o.placement = bestNode; }
        }
    }
}

// (query index, node) of every expanded item of the last call: a superset of what each search visited
__global__ __launch_bounds__(FR_BLOCK) void k_fr_export(FPools fp, long long nU, long long nC, long long nR, int32_t *outQ, int32_t *outNode)
{
    const long long n = nU + nC + nR;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const FItem &it = i < nU ? fp.U[i] : (i < nU + nC ? fp.C[i - nU] : fp.C[fp.capCC + (i - nU - nC)]);
        outQ[i] = it.q; outNode[i] = it.t1;
    }
}

struct FrontierScratch {
    DevBuf<uint8_t> itemsU, itemsC, srch, recs, ctr;
    DevBuf<uint2> tw, sw, bw;
    DevBuf<double> ta, sa, sais, ba;
    DevBuf<LRec> trec, arec;                       // one 16-byte record per temporary list / per list of the arena (FPools)
    DevBuf<int32_t> nodes, expQ, expNode;
    DevBuf<uint8_t> out, wideBr, tflag, overHint;
    DevBuf<int32_t> wideRow, wideQ, wideCtr, perm, perm2, perm3, perm4, tot, lsize, lpos, lpar, passList, passListR, deferred;
    DevBuf<uint2> bw2; DevBuf<double> ba2;   // k_fr_pass's own shared scratch (it runs next to the updating levels)
    DevBuf<uint8_t> visit;
    DevBuf<unsigned long long> lvl;
    DevBuf<long long> vbase;
    std::vector<unsigned long long> lastLvl;        // [levels][4]: the level ranges of the last call (frontier_level_profile)
    std::vector<size_t> lastSlotsU, lastSlotsC;    // ... and the timing slots of its level kernels
    long long lastU = -1, lastC = -1, lastR = 0;     // items of the last call (for frontier_export), -1: none
    long long needR = 0;
    long long needU = 0, needC = 0, needL = 0, needW = 0, needA = 0, needM = 0;   // what the last call asked of the pools, and its searches
    bool lastOverflow = false;         // ... and whether one of them ran over
    FPools lastPools{};
    // the two kernels of a level read disjoint items (they only meet in the pools' atomic counters): the cached-regime one runs
    // on a stream of its own next to the updating one -- a level of the latter lasts as long as its slowest item, on a few lanes
    int capE = 0;                                    // entries of a per-lane scratch slab
    hipStream_t side = nullptr, side2 = nullptr;     // (side: the cached-regime launches; side2: the k_fr_pass kernels next to them)
    hipEvent_t evFork = nullptr, evJoin = nullptr, evFork2 = nullptr, evJoin2 = nullptr;
};

}  // namespace

void frontier_scratch_free(maple_ctx *c)
{
    FrontierScratch *F = (FrontierScratch *)c->frontier;
    if (!F) return;
    F->itemsU.release(); F->itemsC.release(); F->srch.release(); F->recs.release(); F->ctr.release();
    F->tw.release(); F->sw.release(); F->ta.release(); F->sa.release(); F->sais.release(); F->bw.release(); F->ba.release();
    F->trec.release(); F->arec.release(); F->nodes.release(); F->out.release();
    F->expQ.release(); F->expNode.release();
    F->perm.release(); F->perm2.release(); F->perm3.release(); F->perm4.release(); F->tot.release(); F->lsize.release(); F->lpos.release(); F->lpar.release(); F->visit.release(); F->lvl.release(); F->vbase.release(); F->wideBr.release(); F->wideRow.release(); F->wideQ.release(); F->wideCtr.release(); F->passList.release(); F->passListR.release(); F->deferred.release(); F->overHint.release(); F->bw2.release(); F->ba2.release(); F->tflag.release();
    if (F->evFork) (void)hipEventDestroy(F->evFork);
    if (F->evJoin) (void)hipEventDestroy(F->evJoin);
    if (F->evFork2) (void)hipEventDestroy(F->evFork2);
    if (F->evJoin2) (void)hipEventDestroy(F->evJoin2);
    if (F->side) (void)hipStreamDestroy(F->side);
    if (F->side2) (void)hipStreamDestroy(F->side2);
    delete F;
    c->frontier = nullptr;
}


// Runs the searches nodes[0..m) through the frontier tier; out[k] as k_spr_search leaves it: status 0 / 1 / 2 / -1 final,
// -5 = over `budget` expanded items (dense tier), FR_STATUS_FALLBACK = hand to the one-lane-per-search kernel.
int frontier_search(maple_ctx *c, const SearchParams &P, int m, const int32_t *nodes, int budget, int zeroBudget, SearchOut *hostOut,
                    uint2 *poolW, double *poolA, unsigned long long *poolUsed, long long poolCapW, long long poolCapA,
                    FrontierStats *stats, long long itemsHint, const FrontierWide *wide, const uint8_t *overHint)
{
    if (!c->frontier) c->frontier = new FrontierScratch();
    FrontierScratch &F = *(FrontierScratch *)c->frontier;
    if (budget < 1) budget = 1 << 30;
    const auto tEnter = std::chrono::steady_clock::now();
    auto sinceEnter = [&] { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tEnter).count() * 1e-3; };
    const bool dbgTime = c->tuning.verbose != 0;
    // pools, sized for the batch and bounded by what the device has free
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = (size_t)8 << 30;
    const DevBufStats allocs0 = devbuf_stats();
    struct PoolCap { const char *name; const size_t *cap; size_t elem, before; };
    PoolCap poolCaps[] = {{"itemsU", &F.itemsU.cap, 1, 0}, {"itemsC", &F.itemsC.cap, 1, 0}, {"tw", &F.tw.cap, 8, 0}, {"ta", &F.ta.cap, 8, 0},
                          {"trec", &F.trec.cap, 16, 0}, {"sw", &F.sw.cap, 8, 0}, {"sa", &F.sa.cap, 8, 0}, {"sais", &F.sais.cap, 8, 0},
                          {"bw", &F.bw.cap, 8, 0}, {"bw2", &F.bw2.cap, 8, 0}, {"visit", &F.visit.cap, 1, 0}, {"recs", &F.recs.cap, 1, 0},
                          {"perm", &F.perm.cap, 4, 0}, {"passList", &F.passList.cap, 4, 0}};
    for (PoolCap &pc : poolCaps) pc.before = *pc.cap;
    const size_t held = F.itemsU.cap + F.itemsC.cap + F.tw.cap * sizeof(uint2) + F.ta.cap * sizeof(double)
                        + F.sw.cap * sizeof(uint2) + F.sa.cap * sizeof(double) + F.sais.cap * sizeof(double) + F.bw.cap * 48 + F.bw2.cap * 48;
    const double room = 0.5 * (double)(freeB + held);
    // (first guesses, for a context's first batch: afterwards the pools go by what the batch before used.  With an error model a fifth
    // of the searches runs to the budget -- a quarter of it per search on average, and three times the temporary lists: measured 1 500
    // cached items and 61 lists per search at 1 000 000 tips, where a first call that ran over took 2.9-5.3 s and the two calls
    // after it 2.3-3.1 s before the pools had grown)
    const bool longSearches = c->dm.usingErrorRate && !(wide && wide->forceWide);
    const long long perSearchC = longSearches ? std::min<long long>(budget, std::max<long long>(768, budget / 4)) : std::min<long long>(budget, 768);
    long long capC = std::max<long long>(1 << 16, std::max((long long)m * perSearchC, itemsHint));
    long long capU = std::max<long long>(1 << 14, (long long)m * std::min<long long>(budget, 16));
    // (roots: the cached-regime items the list-updating items push -- the upper part of the cached pool)
    long long capR = std::max<long long>(1 << 16, std::min(capC, (long long)m * std::min<long long>(budget, 32)));
    const long long meanEnt = std::max<long long>(16, c->h_n_ent.empty() ? 64 : c->used_ent / (long long)c->h_n_ent.size());   // entries per list in the arena
    long long capL = (longSearches ? 6 : 2) * capU;
    long long capW = 2 * capL * meanEnt, capA = capW;
    if (F.needM > 0 && !(wide && wide->forceWide)) {
        // what the last batch asked for, scaled to this one: with an error model the searches are several times as long as the
        // first guess, and a pool that runs over hands its searches back
        // (after an overflow the asks themselves are too low -- the searches that were handed back stopped asking: twice, not 1.25 x)
        const double f = (F.lastOverflow ? 2.0 : 1.25) * std::min(16.0, (double)m / (double)F.needM);   // (a far smaller batch is no measure: capped)
        capU = std::max(capU, (long long)(f * F.needU)); capC = std::max(capC, (long long)(f * F.needC)); capR = std::max(capR, (long long)(f * F.needR));
        capL = std::max(capL, (long long)(f * F.needL));
        // (the words' first guess goes with the arena's mean list length, which moves by an entry when the tree is uploaded again:
        // 47 -> 48 made two 5 GB pools "grow" by half -- 0.73 s of the first round on a changed tree.  Once a batch has said what it
        // used, that alone sizes them.)
        capW = std::max<long long>(1 << 20, (long long)(f * F.needW)); capA = std::max<long long>(1 << 20, (long long)(f * F.needA));
    }
    // per-lane scratch for lists of up to capE entries (two average lists merged, with room); the few longer ones -- near the
    // root -- take pieces of a shared region.  As many lanes as the GPU holds at once at this kernel's occupancy.
    // (in steps of 32 and never below what the slabs were cut for: they are 6 GB, and 282 -> 288 entries would allocate them again)
    const int capE = std::max(F.capE, std::max(256, std::min(1024, (6 * (int)meanEnt + 31) / 32 * 32)));
    F.capE = capE;
    long long scratchLanes = m <= 64 ? 16384 : 256ll * 4 * 4 * 64;         // 256 CUs x 4 SIMDs x 4 wavefronts (a handful of searches: what they can use)
    // slabs beyond the lanes' own: one per wavefront of the two wavefront-wide kernels (256 + 1 280 <= 2 048) and, on a tree with MAT
    // local references, one per lane of k_fr_pass (256 workgroups), which runs next to k_fr_updating
    const bool matTree = c->tree_has_mut;
    const long long extraSlabs = 2048 + (matTree ? 256ll * FR_BLOCK : 0);
    while (scratchLanes > 16384 && (double)(scratchLanes + extraSlabs) * capE * 64 > 0.25 * room) scratchLanes /= 2;
    const int gridUpd = (int)(scratchLanes / FR_BLOCK);
    const long long capBig = std::max<long long>(1 << 20, 64ll * 1024 * std::max(1, c->tree_max_ent));
    const long long capBig2 = matTree ? std::max<long long>(1 << 20, capBig / 4) : 0;
    {   // What a pool already holds it keeps (a request below its capacity costs nothing, and a pool that shrinks and grows from
        // call to call is freed and allocated again -- tens of GB: 1.3 s of a 1 000 000-tip step); what a pool would GROW by is cut
        // down proportionally if the total would not fit.  (The temporary lists' pools used to be set back to their first guess
        // whenever the total met the room -- always, at 1 000 000 tips: the list words overflowed on EVERY step and sent 15 000
        // of 131 072 searches to the one-lane kernel, 550 ms.)
        const long long curU = (long long)(F.itemsU.cap / sizeof(FItem)), curCR = (long long)(F.itemsC.cap / sizeof(FItem));
        const long long curL = (long long)std::min(F.trec.cap, F.tflag.cap);
        const long long curW = (long long)F.tw.cap, curA = (long long)F.ta.cap;
        const double perItem = (double)(sizeof(FItem) + sizeof(FVisit) + 12);
        const double fixed = (double)(scratchLanes + extraSlabs) * capE * 64 + (double)(capBig + capBig2) * 48;
        const double base = (double)(curCR + curU) * perItem + (double)(curW + curA) * 8 + (double)curL * 24 + fixed;
        const double gCR = std::max<double>(0.0, (double)(capC + capR - curCR)), gU = std::max<double>(0.0, (double)(capU - curU));
        const double gL = std::max<double>(0.0, (double)(capL - curL)), gW = std::max<double>(0.0, (double)(capW - curW)), gA = std::max<double>(0.0, (double)(capA - curA));
        const double growth = (gCR + gU) * perItem + (gW + gA) * 8 + gL * 24;
        double f = 1.0;
        if (base + growth > room && growth > 0) f = std::max(0.0, (room - base) / growth);
        const double wantCR = std::max<double>((double)curCR, (double)curCR + f * gCR);
        const double shareR = (double)capR / (double)(capC + capR);
        capR = std::max<long long>(1 << 16, (long long)(wantCR * shareR)); capC = std::max<long long>(1 << 16, (long long)wantCR - capR);
        capU = std::max<long long>(1 << 14, std::max(curU, (long long)(curU + f * gU)));
        capL = std::max<long long>(2 * (1 << 14), std::max(curL, (long long)(curL + f * gL)));
        capW = std::max<long long>(1 << 20, std::max(curW, (long long)(curW + f * gW)));
        capA = std::max<long long>(1 << 20, std::max(curA, (long long)(curA + f * gA)));
    }
    const long long capRecs = std::max<long long>(1 << 14, (long long)m * 16);
    auto grow = [](size_t want, size_t cap) { return want <= cap ? cap : std::max(want, cap + cap / 2); };
    HIPCK(c, F.itemsU.reserve_exact(grow((size_t)capU * sizeof(FItem), F.itemsU.cap)));
    HIPCK(c, F.itemsC.reserve_exact(grow((size_t)(capC + capR) * sizeof(FItem), F.itemsC.cap)));
    HIPCK(c, F.srch.reserve((size_t)m * sizeof(FSearch)));
    HIPCK(c, F.recs.reserve_exact(grow((size_t)capRecs * sizeof(FRec), F.recs.cap)));
    HIPCK(c, F.ctr.reserve(sizeof(FCtr)));
    HIPCK(c, F.tw.reserve_exact(grow((size_t)capW, F.tw.cap)));
    HIPCK(c, F.ta.reserve_exact(grow((size_t)capA, F.ta.cap)));
    HIPCK(c, F.trec.reserve_exact(grow((size_t)capL, F.trec.cap)));
    HIPCK(c, F.arec.reserve_exact(grow(c->h_n_ent.size(), F.arec.cap)));
    HIPCK(c, F.tflag.reserve_exact(grow((size_t)capL, F.tflag.cap)));
    HIPCK(c, F.sw.reserve_exact((size_t)(scratchLanes + extraSlabs) * capE));
    HIPCK(c, F.sa.reserve_exact((size_t)(scratchLanes + extraSlabs) * capE * 5));
    HIPCK(c, F.sais.reserve_exact((size_t)(scratchLanes + extraSlabs) * capE * 2));
    HIPCK(c, F.bw.reserve_exact((size_t)capBig));
    HIPCK(c, F.ba.reserve_exact((size_t)capBig * 5));
    if (capBig2) { HIPCK(c, F.bw2.reserve_exact((size_t)capBig2)); HIPCK(c, F.ba2.reserve_exact((size_t)capBig2 * 5)); }
    HIPCK(c, F.nodes.reserve((size_t)m));
    HIPCK(c, F.out.reserve((size_t)m * sizeof(SearchOut)));
    FPools fp;
    fp.U = (FItem *)F.itemsU.p; fp.C = (FItem *)F.itemsC.p;
    fp.capU = (long long)(F.itemsU.cap / sizeof(FItem)); fp.capC = (long long)(F.itemsC.cap / sizeof(FItem));
    // (a buffer kept from a larger call: both parts grow with it)
    fp.capCC = fp.capC - std::max(capR, (long long)((double)fp.capC * capR / (double)(capC + capR)));
    fp.tw = F.tw.p; fp.ta = F.ta.p; fp.trec = F.trec.p; fp.arec = F.arec.p; fp.nArec = (int32_t)c->h_n_ent.size();
    fp.capW = (long long)F.tw.cap; fp.capA = (long long)F.ta.cap;
    fp.tflag = F.tflag.p;
    fp.capL = (long long)std::min(F.trec.cap, F.tflag.cap);
    HIPCK(c, F.perm.reserve_exact(std::max(F.perm.cap, (size_t)fp.capU)));
    HIPCK(c, F.perm2.reserve_exact(std::max(F.perm2.cap, (size_t)fp.capU)));
    HIPCK(c, F.perm3.reserve_exact(std::max(F.perm3.cap, (size_t)fp.capU)));
    HIPCK(c, F.perm4.reserve_exact(std::max(F.perm4.cap, (size_t)fp.capU)));
    fp.perm = F.perm.p; fp.perm2 = F.perm2.p; fp.perm3 = F.perm3.p; fp.perm4 = F.perm4.p;
    fp.waveAllBelow = c->tuning.waveAllBelow > 0 ? c->tuning.waveAllBelow : (c->tuning.waveAllBelow < 0 ? 0 : 49152);
    // the visiting-order layout of the items (k_fr_layout_*): one 32-byte record per item, the levels' ranges, per-search bases
    fp.maxLevels = 4096;
    fp.capVisit = fp.capU + fp.capC;
    const bool layoutOK = m > 64 && fp.capVisit < (1ll << 31) - 64;       // (ranks are 32-bit; a handful of searches is not worth 90 launches)
    if (layoutOK) {
        HIPCK(c, F.visit.reserve_exact(std::max(F.visit.cap, (size_t)fp.capVisit * sizeof(FVisit))));
        HIPCK(c, F.lsize.reserve_exact(std::max(F.lsize.cap, (size_t)fp.capVisit)));
        HIPCK(c, F.lpos.reserve_exact(std::max(F.lpos.cap, (size_t)fp.capVisit)));
        HIPCK(c, F.lpar.reserve_exact(std::max(F.lpar.cap, (size_t)fp.capVisit)));
    }
    fp.lsize = F.lsize.p; fp.lpos = F.lpos.p; fp.lpar = F.lpar.p;
    HIPCK(c, F.lvl.reserve((size_t)fp.maxLevels * FR_LVL));
    HIPCK(c, F.tot.reserve((size_t)m));
    HIPCK(c, F.vbase.reserve((size_t)m + 1));
    fp.visit = layoutOK ? (FVisit *)F.visit.p : nullptr; fp.lvl = F.lvl.p; fp.tot = F.tot.p; fp.vbase = F.vbase.p;
    fp.sw = F.sw.p; fp.sa = F.sa.p; fp.sais = F.sais.p; fp.capE = capE;
    fp.bw = F.bw.p; fp.ba = F.ba.p; fp.capBig = (long long)F.bw.cap; fp.bigSide = 0;
    fp.ctr = (FCtr *)F.ctr.p; fp.S = (FSearch *)F.srch.p; fp.recs = (FRec *)F.recs.p;
    fp.capRecs = (long long)(F.recs.cap / sizeof(FRec));
    // trees with MAT local references: the lists a search carries go through the reference branches it crosses
    fp.mat = c->tree_has_mut ? 1 : 0;
    fp.mv = mview(c);
    fp.passList = nullptr; fp.capPass = 0; fp.passListR = nullptr; fp.capPassR = 0; fp.deferred = nullptr; fp.capDeferred = 0;
    if (fp.mat) {
        HIPCK(c, F.passList.reserve_exact(std::max(F.passList.cap, (size_t)std::max<long long>(1 << 16, fp.capC / 8))));
        fp.passList = F.passList.p; fp.capPass = (long long)F.passList.cap;
        HIPCK(c, F.passListR.reserve_exact(std::max(F.passListR.cap, (size_t)std::max<long long>(1 << 16, (fp.capC - fp.capCC) / 4))));
        fp.passListR = F.passListR.p; fp.capPassR = (long long)F.passListR.cap;
        HIPCK(c, F.deferred.reserve_exact(F.passList.cap + F.passListR.cap));
        fp.deferred = F.deferred.p; fp.capDeferred = (long long)F.deferred.cap;
    }
    if (dbgTime) fprintf(stderr, "[maple]   frontier +%.1f ms: pools reserved (%zu allocations, %.2f GB; free %.1f GB, room %.1f GB; per-lane scratch %lld x %d entries, shared %lld + %lld)\n", sinceEnter(),
                         devbuf_stats().allocs - allocs0.allocs, 1e-9 * (double)(devbuf_stats().bytes - allocs0.bytes), 1e-9 * (double)freeB, 1e-9 * room,
                         scratchLanes + extraSlabs, capE, capBig, capBig2);
    if (dbgTime)
        for (const PoolCap &pc : poolCaps)
            if (*pc.cap != pc.before) fprintf(stderr, "[maple]     %s: %.3f -> %.3f GB\n", pc.name, 1e-9 * (double)(pc.before * pc.elem), 1e-9 * (double)(*pc.cap * pc.elem));
    hipStream_t s = c->stream;
    const bool dbgSync = c->tuning.verbose > 2;                            // (MAPLE_DEBUG=3: every launch awaited and named)
    auto stage = [&](const char *what) -> int {
        if (!dbgSync) return MAPLE_OK;
        HIPCK(c, hipDeviceSynchronize());
        fprintf(stderr, "[maple] frontier: %s done\n", what);
        return MAPLE_OK;
    };
    HIPCK(c, hipMemsetAsync(fp.ctr, 0, sizeof(FCtr), s));
    HIPCK(c, hipMemcpyAsync(F.nodes.p, nodes, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (overHint) {
        HIPCK(c, F.overHint.reserve((size_t)m));
        HIPCK(c, hipMemcpyAsync(F.overHint.p, overHint, (size_t)m, hipMemcpyHostToDevice, s));
    }
    SearchOut *dout = (SearchOut *)F.out.p;
    DevTree T = c->dtree;
    const ArenaViewS av = view(c);
    const int gridN = std::max(1, std::min(2048, (m + FR_BLOCK - 1) / FR_BLOCK));
    hipEvent_t e0, e1, er0, er1;
    TRY(maple_internal_ev_pair(c, &e0, &e1, MAPLE_K_SPR_SEARCH, (double)m, 0.0));
    const size_t slotEv = c->ev_used / 2 - 1;
    HIPCK(c, hipEventRecord(e0, s));
    // whole-tree searches whose rows of the dense score table are being made on another stream (FS_WIDE)
    std::vector<int32_t> wideIdx;
    const bool useWide = wide && wide->rowOf && wide->cacheS && (wide->forceWide || budget > zeroBudget) && c->scan_valid && !c->tuning.noCladeScan
                         && (size_t)(c->tree_max_depth + 2) * 16 <= (48u << 10);
    DevTree Tw = T;                                                         // (with the tables of the clade scan, as k_spr_search gets them)
    if (useWide) {
        Tw.scan = c->t_scan.p; Tw.scanParent = c->t_scan_parent.p; Tw.scanDepthCap = c->tree_max_depth + 2;
        Tw.candBefore = c->t_cand_before.p; Tw.cladeVisits = c->t_clade_visits.p;
    }
    if (useWide) {
        for (int k = 0; k < m; k++) if (wide->rowOf[k] >= 0) wideIdx.push_back(k);
        // (the searches with the largest clades to scan first)
        if ((int)c->h_clade.size() == T.n)
            std::stable_sort(wideIdx.begin(), wideIdx.end(), [&](int a, int b) { return c->h_clade[nodes[a]] > c->h_clade[nodes[b]]; });
        HIPCK(c, F.wideRow.reserve((size_t)m));
        HIPCK(c, F.wideQ.reserve(std::max<size_t>(1, wideIdx.size())));
        HIPCK(c, F.wideCtr.reserve(4));
        HIPCK(c, hipMemcpyAsync(F.wideRow.p, wide->rowOf, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (!wideIdx.empty())
            HIPCK(c, hipMemcpyAsync(F.wideQ.p, wideIdx.data(), wideIdx.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIPCK(c, hipMemsetAsync(F.wideCtr.p, 0, 4 * sizeof(int32_t), s));
        HIPCK(c, hipStreamSynchronize(s));                                  // (wideIdx is a local)
    }
    const bool anyWide = useWide && !wideIdx.empty();
    if (fp.nArec > 0) {
        k_fr_arena_recs<<<std::min(4096, (fp.nArec + FR_BLOCK - 1) / FR_BLOCK), FR_BLOCK, 0, s>>>(av, fp.nArec, F.arec.p);
        HIPCK(c, hipGetLastError());
    }
    // (the seeding of a tree with local references writes lists through the lanes' scratch slabs: no more lanes than slabs)
    FR_DISPATCH3(c, k_fr_begin, <<<std::min(gridN, gridUpd), FR_BLOCK, 0, s>>>(c->d_model, av, T, P, m, F.nodes.p, fp, dout, budget, zeroBudget,
                                                            anyWide ? F.wideRow.p : nullptr, anyWide ? wide->forceWide : 0,
                                                            overHint ? F.overHint.p : nullptr));
    HIPCK(c, hipGetLastError());
    // level loop: the counters stay on the device; the host looks at them every few levels
    FCtr hc;
    std::memset(&hc, 0, sizeof hc);
    int levels = 0;
    // (many short-lived workgroups rather than a grid-stride loop over few long-lived ones: wavefront slots come free all the
    // time, and the dispatcher hands them to the higher-priority stream first)
    // (a launch of 16 384 workgroups costs 0.4-0.5 ms even when it finds ten thousand items: the grid goes with the batch -- a
    // round's largest launch holds ~21 items per search)
    const int gridCached = (int)std::min<long long>(16384, std::max<long long>(512, (long long)m / 12));
    // items whose two lists add up to this many entries are walked by a wavefront each (k_fr_updating_wave: 54 KB of LDS per
    // wavefront, two per compute unit)
    // (a handful of searches -- the re-search of a proposed move -- wait for every single item: all of them by wavefronts)
    const int heavyMin = m <= 64 ? 1 : std::max(256, FR_HEAVY_MULT * (int)meanEnt / 4), gridWave = 256, gridWaveSmall = 1280;
    const int bigMin = m <= 64 ? (1 << 30) : FR_BIG_MIN;                  // (lists of this many entries together: 16 items to a wavefront)
    std::vector<size_t> slotsC, slotsU;
    if (!F.side) {
        // (the cached-regime kernel's stream at the LOWEST priority: its kernel fills the GPU -- 128 registers x 4 wavefronts is a
        // SIMD's whole register file -- and the level's list-updating kernels on the context's stream, a chain of four short
        // launches that the level waits for, must not queue behind it for wavefront slots)
        {
            int prLow = 0, prHigh = 0;
            if (hipDeviceGetStreamPriorityRange(&prLow, &prHigh) != hipSuccess) prLow = 0;
            HIPCK(c, hipStreamCreateWithPriority(&F.side, hipStreamNonBlocking, prLow));
            HIPCK(c, hipStreamCreateWithPriority(&F.side2, hipStreamNonBlocking, prLow));
        }
        HIPCK(c, hipEventCreateWithFlags(&F.evFork2, hipEventDisableTiming));
        HIPCK(c, hipEventCreateWithFlags(&F.evJoin2, hipEventDisableTiming));
        HIPCK(c, hipEventCreateWithFlags(&F.evFork, hipEventDisableTiming));
        HIPCK(c, hipEventCreateWithFlags(&F.evJoin, hipEventDisableTiming));
    }
    const hipStream_t s2 = F.side, s3 = F.side2;
    // The two kinds of items run on two streams that never wait for each other inside the expansion: the list-updating levels
    // on the context's stream -- a chain of ~20 levels, each as long as its slowest item -- and the cached-regime launches on the
    // side stream, each taking whatever was complete when it started (k_fr_snap_c).  A cached launch is queued behind every
    // updating level's publication (two per level: the cached subtrees are deeper than the updating chain is long), the rest
    // follow when the updating items are done.  (Round 4 joined the streams after every level: sum over levels of
    // max(updating, cached) instead of the longer of the two sums.)
    FPools fpC = fp;                                                       // (k_fr_pass: its own shared scratch and counter)
    fpC.bw = F.bw2.p; fpC.ba = F.ba2.p; fpC.capBig = (long long)F.bw2.cap; fpC.bigSide = 1;
    int launchesC = 0;
    auto u_level = [&]() -> int {                                          // the kernels of the open level, between events of their own
        hipEvent_t a0, a1;
        TRY(maple_internal_ev_pair(c, &a0, &a1, MAPLE_K_FR_UPDATING, 0.0, 0.0));
        slotsU.push_back(c->ev_used / 2 - 1);
        HIPCK(c, hipEventRecord(a0, s));
        k_fr_sort_level<<<512, FR_BLOCK, 0, s>>>(av, T, fp, heavyMin, bigMin);
        TRY(stage("k_fr_sort_level"));
        // (the level's three kinds of list-updating items behind each other on this stream.  The 512-entry class on a stream of its
        // own was measured: 175 ms per round instead of 152 -- a third side stream shares a hardware queue with the cached-regime
        // kernel's and runs behind it.)
        TRY(fr_launch_updating(c, s, gridUpd, av, T, P, fp, budget, heavyMin));
        TRY(stage("k_fr_updating"));
        if (heavyMin > 0) {
            TRY(fr_launch_updating_wave_small(c, s, gridWaveSmall, av, T, P, fp, budget, heavyMin, scratchLanes + 256));
            TRY(fr_launch_updating_wave(c, s, gridWave, av, T, P, fp, budget, heavyMin, scratchLanes));
        }
        TRY(stage("k_fr_updating_wave"));
        HIPCK(c, hipEventRecord(a1, s));
        // the next level is opened, and what this one pushed into the cached pool is published
        k_fr_snap_u<<<1, 64, 0, s>>>(fp.ctr, fp.capU, fp.capC - fp.capCC, fp.capPassR, fp.lvl, fp.maxLevels);
        HIPCK(c, hipEventRecord(F.evFork, s));
        HIPCK(c, hipStreamWaitEvent(s2, F.evFork, 0));
        levels++;
        return MAPLE_OK;
    };
    bool passQueued = false;
    auto c_launch = [&]() -> int {
        hipEvent_t b0, b1;
        TRY(maple_internal_ev_pair(c, &b0, &b1, MAPLE_K_FR_CACHED, 0.0, 0.0));
        slotsC.push_back(c->ev_used / 2 - 1);
        if (fp.mat && passQueued) HIPCK(c, hipStreamWaitEvent(s2, F.evJoin2, 0));      // (the deferred items of the launch before)
        k_fr_snap_c<<<1, 64, 0, s2>>>(fp.ctr, fp.capCC, fp.capPass, fp.capDeferred, fp.lvl, fp.maxLevels);
        HIPCK(c, hipEventRecord(b0, s2));
        if (fp.mat) {
            // the removed lists of the launch's items that crossed a reference branch are re-expressed NEXT to the launch's
            // k_fr_cached, on a stream of their own: the launch scores every other item, these are the next launch's
            HIPCK(c, hipEventRecord(F.evFork2, s2));
            HIPCK(c, hipStreamWaitEvent(s3, F.evFork2, 0));
            FR_DISPATCH3(c, k_fr_pass_wave, <<<1024, FR_BLOCK, 0, s3>>>(c->d_model, av, T, fp));
            FR_DISPATCH3(c, k_fr_pass, <<<256, FR_BLOCK, 0, s3>>>(c->d_model, av, T, fpC, scratchLanes + 2048));
            HIPCK(c, hipEventRecord(F.evJoin2, s3));
            passQueued = true;
            TRY(stage("k_fr_pass"));
        }
        FR_DISPATCH3(c, k_fr_cached, <<<gridCached, FR_BLOCK, 0, s2>>>(c->d_model, av, anyWide ? Tw : T, P, fp, budget,
                                                                        anyWide ? F.wideRow.p : nullptr,
                                                                        anyWide ? wide->fin : FiniteRows{nullptr, nullptr, 0}));
        HIPCK(c, hipEventRecord(b1, s2));
        TRY(stage("k_fr_cached"));
        launchesC++;
        return MAPLE_OK;
    };
    // (an error inside the loop leaves nothing in flight on either stream behind it)
    auto bail = [&](int rc) { (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(s2); (void)hipStreamSynchronize(s3); return rc; };
    // the first level is opened (the seeds of k_fr_begin) and the cached stream let go
    k_fr_snap_u<<<1, 64, 0, s>>>(fp.ctr, fp.capU, fp.capC - fp.capCC, fp.capPassR, fp.lvl, fp.maxLevels);
    if (hipGetLastError() != hipSuccess || hipEventRecord(F.evFork, s) != hipSuccess || hipStreamWaitEvent(s2, F.evFork, 0) != hipSuccess
        || (anyWide && wide->rowsReady && hipStreamWaitEvent(s2, wide->rowsReady, 0) != hipSuccess))   // (k_fr_cached reads the rows' bitmaps)
        return bail(fail(c, MAPLE_ERR_HIP, "frontier level loop: HIP error"));
    // the host looks at the counters every few levels: a handful of searches (the re-search of a proposed move) is over after a few
    // levels and each look costs them less than the levels it saves; a whole round runs ~20 updating levels
    const int groupLevels = m <= 64 ? 2 : 8;
    for (;;) {
        for (int g = 0; g < groupLevels; g++) {
            { const int rc_ = u_level(); if (rc_) return bail(rc_); }
            for (int k2 = 0; k2 < 2; k2++) { const int rc_ = c_launch(); if (rc_) return bail(rc_); }
        }
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&hc, fp.ctr, sizeof(FCtr), hipMemcpyDeviceToHost, s) != hipSuccess
            || hipStreamSynchronize(s) != hipSuccess)
            return bail(fail(c, MAPLE_ERR_HIP, "frontier level loop: HIP error"));
        if (hc.hiU == hc.loU) break;                                       // (the level that was just opened is empty)
        if (levels > 100000) return bail(fail(c, MAPLE_ERR_FATAL, "frontier search did not terminate"));
    }
    // the cached-regime items that are left: launches until one has found nothing new behind its snapshot
    for (;;) {
        for (int g = 0; g < groupLevels; g++) { const int rc_ = c_launch(); if (rc_) return bail(rc_); }
        if (passQueued && hipStreamWaitEvent(s2, F.evJoin2, 0) != hipSuccess) return bail(fail(c, MAPLE_ERR_HIP, "frontier level loop: HIP error"));
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&hc, fp.ctr, sizeof(FCtr), hipMemcpyDeviceToHost, s2) != hipSuccess
            || hipStreamSynchronize(s2) != hipSuccess)
            return bail(fail(c, MAPLE_ERR_HIP, "frontier level loop: HIP error"));
        const unsigned long long haveC = std::min<unsigned long long>(hc.usedC, (unsigned long long)fp.capCC),
                                 haveR = std::min<unsigned long long>(hc.usedR, (unsigned long long)(fp.capC - fp.capCC));
        if (hc.hiC == haveC && hc.hiR == haveR && hc.hiD == std::min<unsigned long long>(hc.nDeferred, (unsigned long long)fp.capDeferred)) break;
        if (launchesC > 200000) return bail(fail(c, MAPLE_ERR_FATAL, "frontier search did not terminate"));
    }
    if (hipEventRecord(F.evJoin, s2) != hipSuccess || hipStreamWaitEvent(s, F.evJoin, 0) != hipSuccess)
        return bail(fail(c, MAPLE_ERR_HIP, "frontier level loop: HIP error"));
    if (dbgTime) fprintf(stderr, "[maple]   frontier +%.1f ms: expansion done (%d updating levels, %d cached launches)\n", sinceEnter(), levels, launchesC);
    size_t slotWide = (size_t)-1;
    // (replay, refinement, final selection: an error in here must not return with kernels still in flight on either stream --
    // the caller may reuse or free the pools -- so the stage is a lambda and its status goes through bail() like the loop's)
    auto finishStage = [&]() -> int {
    if (anyWide) {
        // the whole-tree searches: the same exact walk, a wavefront each, the clades in the cached regime scanned over the rows
        // of the dense score table -- next to the other searches' walk (k_fr_replay: one lane each, as long as its longest search)
        hipEvent_t w0, w1;
        TRY(maple_internal_ev_pair(c, &w0, &w1, MAPLE_K_FR_WIDE, 0.0, 0.0));
        slotWide = c->ev_used / 2 - 1;
        const int capB = 512, gridW = (int)std::min<size_t>(8192, wideIdx.size());
        HIPCK(c, F.wideBr.reserve_exact(std::max(F.wideBr.cap, (size_t)gridW * capB * sizeof(BestRec))));
        HIPCK(c, hipEventRecord(F.evFork, s));
        HIPCK(c, hipStreamWaitEvent(s2, F.evFork, 0));
        HIPCK(c, hipEventRecord(w0, s2));
        const size_t dyn = (size_t)Tw.scanDepthCap * (sizeof(double) + sizeof(int) + sizeof(unsigned));
        k_fr_replay_wide<<<gridW, 64, dyn, s2>>>(Tw, P, (int)wideIdx.size(), F.wideQ.p, F.wideRow.p, wide->cacheS, wide->fin, fp, dout,
                                                  (BestRec *)F.wideBr.p, capB, F.wideCtr.p);
        HIPCK(c, hipGetLastError());
        HIPCK(c, hipEventRecord(w1, s2));
        HIPCK(c, hipEventRecord(F.evJoin, s2));
    }
    TRY(maple_internal_ev_pair(c, &er0, &er1, MAPLE_K_FR_REPLAY, (double)m, 0.0));
    HIPCK(c, hipEventRecord(er0, s));
    const int gridLayout = (int)std::min<long long>(1024, std::max<long long>(64, (long long)m / 64));   // (~100 small launches: the grid goes with the batch)
    if (levels <= fp.maxLevels && launchesC <= fp.maxLevels && fp.visit) {
        // the items of every search in the order its walk visits them: sizes bottom-up, ranks top-down, one record each
        HIPCK(c, hipMemsetAsync(fp.lpos, 0xFF, (size_t)fp.capVisit * sizeof(int32_t), s));   // (-1: not laid out)
        for (int l = launchesC - 1; l >= 0; l--) k_fr_layout_sizes<<<gridLayout, FR_BLOCK, 0, s>>>(fp, l, 1);
        for (int l = levels - 1; l >= 0; l--) k_fr_layout_sizes<<<gridLayout, FR_BLOCK, 0, s>>>(fp, l, 0);
        k_fr_layout_totals<<<gridN, FR_BLOCK, 0, s>>>(m, fp);
        k_fr_layout_scan<<<1, 1024, 0, s>>>(m, fp.tot, fp.vbase);
        k_fr_layout_seeds<<<gridN, FR_BLOCK, 0, s>>>(m, fp);
        for (int l = 0; l < levels; l++) k_fr_layout_place<<<gridLayout, FR_BLOCK, 0, s>>>(fp, l, 0);
        for (int l = 0; l < launchesC; l++) k_fr_layout_place<<<gridLayout, FR_BLOCK, 0, s>>>(fp, l, 1);
        HIPCK(c, hipGetLastError());
        TRY(stage("k_fr_layout"));
    } else fp.visit = nullptr;
    k_fr_replay<<<gridN, FR_BLOCK, 0, s>>>(P, m, fp, dout);
    TRY(stage("k_fr_replay"));
    if (anyWide) HIPCK(c, hipStreamWaitEvent(s, F.evJoin, 0));
    if (anyWide && fp.mat && wide->nFrames > 0) {
        FR_DISPATCH3(c, k_fr_wide_frames, <<<std::min(gridUpd, 512), FR_BLOCK, 0, s>>>(c->d_model, av, T, fp, wide->frameParent, wide->frameNode));
        TRY(stage("k_fr_wide_frames"));
    }
    FR_DISPATCH3(c, k_fr_refine, <<<gridUpd, FR_BLOCK, 0, s>>>(c->d_model, av, T, fp));
    TRY(stage("k_fr_refine"));
    k_fr_finish<<<gridN, FR_BLOCK, 0, s>>>(av, T, P, m, fp, dout, poolW, poolA, poolUsed, poolCapW, poolCapA);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(er1, s));
    HIPCK(c, hipEventRecord(e1, s));
    HIPCK(c, hipMemcpyAsync(hostOut, dout, (size_t)m * sizeof(SearchOut), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipMemcpyAsync(&hc, fp.ctr, sizeof(FCtr), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipStreamSynchronize(s));
    return MAPLE_OK;
    };
    { const int rc_ = finishStage(); if (rc_) return bail(rc_); }
    if (dbgTime) fprintf(stderr, "[maple]   frontier +%.1f ms: replay, refinement, final selection done; results on the host\n", sinceEnter());
    if (dbgTime && (hc.fbReason[0] | hc.fbReason[1] | hc.fbReason[2] | hc.fbReason[3] | hc.fbReason[4] | hc.fbReason[5] | hc.fbReason[6] | hc.fbReason[7]))
        fprintf(stderr, "[maple]   exact walk handed searches to the one-lane kernel: %d a merge within tolerance, %d a list re-expressed from a shortened one, "
                        "%d a fifth shortened list, %d the search's own first list shortened, %d marked list without a layout, %d other; whole-tree searches: "
                        "%d short list / scan slots / marked item list, %d a scanned branch whose list shorten() would change\n",
                hc.fbReason[1], hc.fbReason[2], hc.fbReason[3], hc.fbReason[4], hc.fbReason[5], hc.fbReason[0], hc.fbReason[6], hc.fbReason[7]);
    if (dbgTime && hc.overflow)
        fprintf(stderr, "[maple]   frontier pools, asked / capacity: updating items %llu / %lld, cached items %llu / %lld, roots %llu / %lld, temporary lists %llu / %lld "
                        "(words %llu / %lld, aux %llu / %lld), pass entries %llu / %lld, of roots %llu / %lld, records %llu / %lld\n",
                hc.usedU, fp.capU, hc.usedC, fp.capCC, hc.usedR, fp.capC - fp.capCC, hc.nLists, fp.capL, hc.usedW, fp.capW, hc.usedA, fp.capA,
                hc.nPass, fp.capPass, hc.nPassR, fp.capPassR, hc.nRecs, fp.capRecs);
    {   // what this tier did, for maple_timing_read_kind: candidate placements of the searches it finished (SURVEY 8d bytes)
        const double meanCand = c->n_scored ? c->scored_bytes_total / c->n_scored : 0.0;
        double units = 0.0, bytes = 0.0;
        double unitsW = 0.0, bytesW = 0.0;
        for (int k = 0; k < m; k++) {
            if (hostOut[k].status != 0 && hostOut[k].status != -1) continue;
            if (anyWide && wide->rowOf[k] >= 0) {                           // replayed over its score row: 8 B per placement + its list
                const int32_t l = c->h_tree_lower[nodes[k]];
                unitsW += hostOut[k].nAppend;
                bytesW += 8.0 * hostOut[k].nAppend + (l >= 0 ? 8.0 * c->h_n_ent[l] + 8.0 * c->h_n_aux[l] : 0.0);
                continue;
            }
            units += hostOut[k].nAppend;
            const int32_t l = c->h_tree_lower[nodes[k]];
            bytes += meanCand * hostOut[k].nAppend + (l >= 0 ? 8.0 * c->h_n_ent[l] + 8.0 * c->h_n_aux[l] : 0.0);
        }
        c->ev_units[slotEv] = units; c->ev_bytes[slotEv] = bytes;
        if (slotWide != (size_t)-1) { c->ev_units[slotWide] = unitsW; c->ev_bytes[slotWide] = bytesW; }
        // the cached-regime kernel's own share: what its launches scored (counted on the device), booked on the first launch
        if (!slotsC.empty()) { c->ev_units[slotsC[0]] = (double)hc.scoredC; c->ev_bytes[slotsC[0]] = (double)hc.bytesC; }
        if (!slotsU.empty()) { c->ev_units[slotsU[0]] = (double)hc.itemsU; c->ev_bytes[slotsU[0]] = (double)hc.bytesU; }
    }
#ifdef MAPLE_SPR_PROFILE
    {
        const char *nm[8] = {"<64", "<96", "<128", "<192", "<256", "<384", "<512", ">=512"};
        for (int b = 0; b < 8; b++)
            if (hc.dbgCnt[b]) fprintf(stderr, "[maple] one-lane updating items, lists of %s entries: %llu items, mean %.3f ms, slowest %.3f ms\n", nm[b],
                                      hc.dbgCnt[b], hc.dbgT[b] * 1e-5 / hc.dbgCnt[b], hc.dbgMax[b] * 1e-5);
    }
    if (hc.dbgC[0])
        fprintf(stderr, "[maple] k_fr_cached, per wavefront-iteration (%llu of them, %.1f scored items each): %.2f us before the scores (item, search, node, "
                        "list table), %.2f us in the scores (%.1f entries per scored item in its two lists, the longest of an iteration %.1f), %.2f us "
                        "after them (rule, pushes)\n",
                hc.dbgC[0], (double)hc.dbgC[5] / hc.dbgC[0], hc.dbgC[1] * 1e-2 / hc.dbgC[0], hc.dbgC[2] * 1e-2 / hc.dbgC[0],
                (double)hc.dbgC[6] / std::max(1.0, (double)hc.dbgC[5]), (double)hc.dbgC[4] / hc.dbgC[0], hc.dbgC[3] * 1e-2 / hc.dbgC[0]);
    {
        const char *nm[4] = {"small class", "512 class", "small class, one lane (a list does not fit)", "512 class, one lane (a list does not fit)"};
        for (int b = 0; b < 4; b++)
            if (hc.dbgWCnt[b]) fprintf(stderr, "[maple] wavefront-wide updating items, %s: %llu items, mean %.3f ms, slowest %.3f ms\n", nm[b],
                                       hc.dbgWCnt[b], hc.dbgWT[b] * 1e-5 / hc.dbgWCnt[b], hc.dbgWMax[b] * 1e-5);
    }
    if (anyWide) {
        double tw = 0, ts = 0, mw = 0, ms = 0; long long ns = 0, mxn = 0;
        for (int k : wideIdx) {
            tw += hostOut[k].tStep * 1e-5; ts += hostOut[k].tReplay * 1e-5; ns += hostOut[k].nSteps;
            mw = std::max(mw, hostOut[k].tStep * 1e-5); ms = std::max(ms, hostOut[k].tReplay * 1e-5); mxn = std::max<long long>(mxn, hostOut[k].nSteps);
        }
        fprintf(stderr, "[maple] wide replay profile over %zu searches: item walk %.1f ms total (max %.2f), scans %.1f ms total (max %.2f), %lld scans (max %lld)\n",
                wideIdx.size(), tw, mw, ts, ms, ns, mxn);
    }
#endif
    // (a batch of whole-tree searches only says nothing about the next full one -- and neither does a handful of searches: the
    // re-searches of the apply phase, one to 32 at a time, used to leave their asks behind, the next ROUND scaled them up by
    // 200 000 / 1 and grew the pools to whatever fitted: 1.8 s of hipMalloc in the first search after the moves, round 6)
    if (!(anyWide && wide->forceWide) && (m >= 1024 || m >= F.needM)) {
        F.needU = (long long)hc.usedU; F.needC = (long long)hc.usedC; F.needR = (long long)hc.usedR; F.needL = (long long)hc.nLists; F.needW = (long long)hc.usedW;
        F.needA = (long long)hc.usedA; F.needM = m; F.lastOverflow = hc.overflow != 0;
    }
    F.lastU = std::min((long long)hc.usedU, fp.capU); F.lastC = std::min((long long)hc.usedC, fp.capCC);
    F.lastR = std::min((long long)hc.usedR, fp.capC - fp.capCC); F.lastPools = fp;
    {
        const int nl = std::min(std::max(levels, launchesC), fp.maxLevels);
        F.lastLvl.assign((size_t)nl * FR_LVL, 0ull);
        if (nl) HIPCK(c, hipMemcpy(F.lastLvl.data(), fp.lvl, (size_t)nl * FR_LVL * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        // (rows beyond what a stream ran hold an earlier call's numbers: zeroed)
        for (int l = 0; l < nl; l++) {
            unsigned long long *L = F.lastLvl.data() + (size_t)FR_LVL * l;
            if (l >= levels) L[0] = L[1] = L[6] = L[7] = 0;
            if (l >= launchesC) L[2] = L[3] = L[4] = L[5] = 0;
        }
        F.lastSlotsU = slotsU; F.lastSlotsC = slotsC;
    }
    if (hc.overflow) F.lastU = F.lastC = -1;                              // (a pool overflowed: the item lists are not complete)
    if (stats) {
        stats->levels = levels; stats->itemsUpdating = (long long)hc.usedU; stats->itemsCached = (long long)(hc.usedC + hc.usedR);
        stats->tempLists = (long long)hc.nLists; stats->tempWords = (long long)hc.usedW; stats->tempAux = (long long)hc.usedA;
        stats->records = (long long)hc.nRecs; stats->overflow = hc.overflow;
    }
    return MAPLE_OK;
}

// (query index, node) of every item the last frontier_search expanded: *n pairs (MAPLE_ERR_STATE if that call did not run or
// overflowed, MAPLE_ERR_ARG if they do not fit in cap)
int frontier_export(maple_ctx *c, long long cap, int32_t *q, int32_t *node, long long *n)
{
    FrontierScratch *F = (FrontierScratch *)c->frontier;
    if (!F || F->lastU < 0) return fail(c, MAPLE_ERR_STATE, "no complete frontier search to export");
    const long long tot = F->lastU + F->lastC + F->lastR;
    *n = tot;
    if (tot > cap) return fail(c, MAPLE_ERR_ARG, "%lld expanded items do not fit in %lld", tot, cap);
    if (tot == 0) return MAPLE_OK;
    HIPCK(c, F->expQ.reserve((size_t)tot));
    HIPCK(c, F->expNode.reserve((size_t)tot));
    k_fr_export<<<(int)std::min<long long>(1024, (tot + FR_BLOCK - 1) / FR_BLOCK), FR_BLOCK, 0, c->stream>>>(F->lastPools, F->lastU, F->lastC, F->lastR,
                                                                                                          F->expQ.p, F->expNode.p);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(q, F->expQ.p, (size_t)tot * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(node, F->expNode.p, (size_t)tot * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MAPLE_OK;
}

// per level of the last frontier_search: updating / cached items and the HIP-event time of the level's two kernels (ms)
int frontier_level_profile(maple_ctx *c, int cap, long long *itemsU, long long *itemsC, float *msU, float *msC, int *n, long long *waveSmall,
                           long long *waveBig)
{
    FrontierScratch *F = (FrontierScratch *)c->frontier;
    if (!F) return fail(c, MAPLE_ERR_STATE, "no frontier search to report on");
    const int nl = (int)(F->lastLvl.size() / FR_LVL);
    *n = nl;
    for (int l = 0; l < nl && l < cap; l++) {
        const unsigned long long *L = F->lastLvl.data() + (size_t)FR_LVL * l;
        itemsU[l] = (long long)(L[1] - L[0]);
        itemsC[l] = (long long)(L[3] - L[2]) + (long long)(L[5] - L[4]);
        if (waveSmall) waveSmall[l] = (long long)L[6];
        if (waveBig) waveBig[l] = (long long)L[7];
        msU[l] = msC[l] = 0.f;
        const size_t su = l < (int)F->lastSlotsU.size() ? F->lastSlotsU[l] : (size_t)-1 / 4, sc = l < (int)F->lastSlotsC.size() ? F->lastSlotsC[l] : (size_t)-1 / 4;
        if (2 * su + 1 < c->ev_used) { HIPCK(c, hipEventSynchronize(c->evs[2 * su + 1])); HIPCK(c, hipEventElapsedTime(&msU[l], c->evs[2 * su], c->evs[2 * su + 1])); }
        if (2 * sc + 1 < c->ev_used) { HIPCK(c, hipEventSynchronize(c->evs[2 * sc + 1])); HIPCK(c, hipEventElapsedTime(&msC[l], c->evs[2 * sc], c->evs[2 * sc + 1])); }
    }
    return MAPLE_OK;
}
