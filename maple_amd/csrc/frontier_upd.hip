// maple_amd/csrc/frontier_upd.hip -- frontier tier of the SPR search (see frontier.hip): the kernels of the items that arrive with
// needsUpdating == True (M:6982-7091, 7182-7304): lists merged along the path, by one lane per item (k_fr_updating) or, for the few
// with the longest lists, by a wavefront per item (k_fr_updating_wave).
#include "frontier_dev.h"
// (this translation unit's wavefront-wide walks serve the FEW items with the longest lists, one wavefront per compute unit:
// staging areas for lists of 512 entries, 110 KB of LDS)
#define MAPLE_WAVE_CAPW 512
#define MAPLE_WU_IN 512
#include "wave_dev.h"
#include "wave_update.h"

using namespace frt;

namespace {

// ---- items that arrived with needsUpdating == True (M:6982-7091, 7182-7304): lists merged along the path ----------------
// (dir 3: the seeding of a search whose pruned node hangs off the root, M:6916-6960 -- two rootVector calls)
// one such item by one lane (the one-lane list walks of genome_dev.h)
// MAT: the tree has MAT local references -- every list that leaves a frame goes through the branch between the frames
// (passGenomeListThroughBranch; a template switch: the plain tree's kernel carries none of it)
template <bool RV, bool U, bool SS, bool MAT>
__device__ __forceinline__ void fr_upd_item_lane(const Ctx<RV, U, SS> &c, const ArenaViewS &av, const DevTree &T, const SearchParams &P, const FPools &fp,
                                 const int budget, const long long laneId, const long long i, unsigned long long *algBytes = nullptr)
{
        FItem &it = fp.U[i];
        FSearch &S = fp.S[it.q];
        if (!fs_live(S.state)) { it.flags |= FI_DEAD; return; }
        const int q = it.q, t1 = it.t1;
        const NodeRec r1 = T.nd[t1];
        const int hPassed = it.hPassed, hRpr = it.hRpr;
        const double distance = it.distance, lastLK = it.lastLK;
        const bool rt = S.isRemovedTip != 0;
        const double rbl = S.removedBLen;
        bool upd = true;
        double midProb = lastLK;
        Writer wr;
        FScr scr{nullptr, nullptr};
        // merge two lists into scratch (wr / scr); 0 ok, -1 None, -2 fatal / no room (then the search is handed back)
        auto merge = [&](int h1, double b1, bool tp1, int h2, double b2, bool tp2, bool upDown) -> int {
            if (!fvalid(h1) || !fvalid(h2)) return -2;
            const FList l1 = flist(av, fp, h1), l2 = flist(av, fp, h2);
            if (!fscratch(fp, laneId, l1.n + l2.n, scr)) { S.state = FS_FALLBACK; return -2; }
            wr.init(scr.w, scr.a);
            const int r = merge_walk(c, fref(l1), b1, tp1, fref(l2), b2, tp2, upDown, false, 0, 0, wr, nullptr);
            // (the item's algorithmic bytes: the two lists a mergeVectors reads and the one it writes; the lists of the item's
            // areVectorsDifferent and appendProbNode are among them or of the same size)
            if (algBytes) *algBytes += 8ull * (unsigned long long)(l1.n + l1.na + l2.n + l2.na + (r > 0 ? wr.n + wr.na : 0));
            return r == -1 ? -1 : (r < 0 ? -2 : 0);
        };
        // rootVector(list, bLen, isFromTip) without local references (M:4916-4996): the walk, then shorten; a stored handle,
        // -2 when out of room
        auto rootVector = [&](int h, double bLen, bool fromTip) -> int {
            if (!fvalid(h)) return -2;
            const FList l = flist(av, fp, h);
            if (!fscratch(fp, laneId, l.n, scr)) return -2;
            wr.init(scr.w, scr.a);
            root_walk(c, fref(l), bLen, fromTip, wr);
            const int hr = fstore(fp, wr);
            if (hr < 0) return -2;
            const FList lr = flist(av, fp, hr);
            wr.init(scr.w, scr.a);
            shorten_walk(c, fref(lr), lr.n, wr);
            if (wr.n == lr.n) return hr;                                   // nothing merged: the list as it is
            return fstore(fp, wr);
        };
        // a stored list through the branch above a node (-2: no room); the removed list likewise
        auto passL = [&](int h, int mutId, bool up) -> int { return MAT ? fpass_store(fp, av, c.m.lRef, laneId, h, mutId, up) : h; };
        auto passR = [&](int h, int mutId, bool up) -> int { return MAT ? fpass_removed(c, fp, av, laneId, h, mutId, up) : h; };
        if (it.dir == 3) {                                                  // the pruned node's parent is the root; t1 = its sibling
            it.midProb = lastLK;
            it.flags |= FI_UPD_OUT;
            if (r1.c0 >= 0) {
                const int ch1 = r1.c0, ch2 = r1.c1;
                const NodeRec rc1 = T.nd[ch1], rc2 = T.nd[ch2];
                int v1 = rootVector(passL(ftree(rc2.lower), rc2.mutId, true), rc2.dist, rc2.isTip != 0);
                int v2 = v1 < 0 ? -2 : rootVector(passL(ftree(rc1.lower), rc1.mutId, true), rc1.dist, rc1.isTip != 0);
                int rp1 = hRpr, rp2 = hRpr;
                if (MAT && v1 >= 0 && v2 >= 0) {                             // M:6926-6948
                    if (rc1.mutId >= 0) { rp1 = passR(hRpr, rc1.mutId, false); v1 = passL(v1, rc1.mutId, false); }
                    if (rc2.mutId >= 0) { rp2 = passR(hRpr, rc2.mutId, false); v2 = passL(v2, rc2.mutId, false); }
                }
                if (v1 < 0 || v2 < 0 || !fvalid(rp1) || !fvalid(rp2)) { S.state = FS_FALLBACK; return; }
                it.child0 = fpush(fp, budget, q, true, ch1, 0, v1, rc1.dist, lastLK, 0, rp1, it.pathBest);
                it.child1 = fpush(fp, budget, q, true, ch2, 0, v2, rc2.dist, lastLK, 0, rp2, it.pathBest);
            }
            return;
        }
        if (it.dir == 0) {                                                  // moving from a parent to its child, M:6982-7160
            const int upT = r1.up;
            const bool scored = !(upT == S.parent || upT < 0) && (r1.dist > P.effNon0 || r1.upIsRoot);
            if (scored) {
                if (merge(hPassed, distance / 2, false, ftree(r1.lower), distance / 2, r1.isTip != 0, true) != 0) { it.flags |= FI_DEAD; return; }
                const ListRef mid{scr.w, scr.a};
                if (r1.totUp >= 0) { const FList tu = flist(av, fp, ftree(r1.totUp)); if (!differ_walk(c, mid, fref(tu))) upd = false; }
                const FList lr = flist(av, fp, hRpr);
                midProb = append_walk(c, mid, fref(lr), rt, rbl);
                it.flags |= FI_SCORED;
                if (upd && midProb >= it.pathBest - P.thrOptTopo) {          // may be short-listed (M:7071): keep the record's lists
                    const int hm = fstore(fp, wr);
                    if (hm < 0) { S.state = FS_FALLBACK; it.flags |= FI_DEAD; return; }
                    it.hA = hPassed; it.hB = ftree(r1.lower); it.hMid = hm; it.recDist = distance; it.flags |= FI_REC_UPD;
                }
            }
            it.midProb = midProb;
            if (upd) it.flags |= FI_UPD_OUT;
            const PRule pr = p_rule(P, scored, midProb, lastLK, it.failsP, it.pathBest);
            if (pr.go && r1.c0 >= 0) {
                for (int k = 0; k < 2; k++) {                               // child 0 uses vectUpRight, child 1 vectUpLeft
                    const int ch = k == 0 ? r1.c0 : r1.c1, other = k == 0 ? r1.c1 : r1.c0;
                    int ref = FR_NONE;
                    const bool cross = MAT && (k == 0 ? r1.c0Frame : r1.c1Frame) != r1.frameOf;
                    if (upd) {
                        const NodeRec ro = T.nd[other];
                        const int hOther = passL(ftree(ro.lower), ro.mutId, true);   // (the other child's list in t1's frame, M:7107-7109)
                        const int r = hOther == -2 ? -2 : merge(hPassed, distance, false, hOther, ro.dist, ro.isTip != 0, true);
                        if (r == -2) { S.state = FS_FALLBACK; break; }
                        if (r == 0) {
                            int hv = fstore(fp, wr), hr = hRpr;
                            if (hv >= 0 && cross) {                         // into the child's frame, M:7111-7121
                                const int mId = T.nd[ch].mutId;
                                hv = passL(hv, mId, false); hr = passR(hRpr, mId, false);
                            }
                            if (hv < 0 || !fvalid(hr)) { S.state = FS_FALLBACK; break; }
                            ref = fpush(fp, budget, q, true, ch, 0, hv, T.nd[ch].dist, midProb, pr.fails, hr, pr.pathBest);
                        }
                    } else if ((k == 0 ? r1.upRight : r1.upLeft) >= 0)
                        ref = fpush(fp, budget, q, false, ch, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, cross);
                    if (k == 0) it.child0 = ref; else it.child1 = ref;
                }
            }
        } else {                                                             // crawling up from a child to its parent t1, M:7162-7434
            const int other = (it.dir == 1) ? r1.c1 : r1.c0;
            const int upT = r1.up;
            const NodeRec ro = T.nd[other];
            int hBottom = -1;
            // (the parent's upper list and the other child's lower list in t1's frame, M:7186 / 7196)
            const int vectUp = upT >= 0 ? passL(ftree(r1.whichChild ? T.nd[upT].upLeft : T.nd[upT].upRight), r1.mutId, false) : -1;
            const int hOther = passL(ftree(ro.lower), ro.mutId, true);
            if (MAT && (vectUp == -2 || hOther == -2)) { S.state = FS_FALLBACK; return; }
            const bool crossO = MAT && ((it.dir == 1) ? r1.c1Frame : r1.c0Frame) != r1.frameOf, crossU = MAT && r1.upFrame != r1.frameOf;
            const bool scored = upT >= 0 && (r1.dist > P.effNon0 || r1.upIsRoot);
            if (scored) {
                int r = merge(hPassed, distance, false, hOther, ro.dist, ro.isTip != 0, false);
                if (r != 0) { it.flags |= FI_DEAD; return; }
                hBottom = fstore(fp, wr);
                if (hBottom < 0) { S.state = FS_FALLBACK; it.flags |= FI_DEAD; return; }
                r = merge(vectUp, r1.dist / 2, false, hBottom, r1.dist / 2, false, true);
                if (r != 0) { it.flags |= FI_DEAD; return; }
                int hm = -1;
                if (r1.totUp >= 0) {
                    const ListRef mid{scr.w, scr.a};
                    const FList tu = flist(av, fp, ftree(r1.totUp));
                    if (!differ_walk(c, mid, fref(tu))) upd = false;
                } else {
                    // "Node has no probVectTotUp ... calculating new one", M:7198-7200: midTot is compared with a list merged on
                    // the spot (midTot moves to the arena first: the scratch is needed for that merge)
                    hm = fstore(fp, wr);
                    if (hm < 0) { S.state = FS_FALLBACK; it.flags |= FI_DEAD; return; }
                    const int rc = merge(vectUp, r1.dist / 2, false, ftree(r1.lower), r1.dist / 2, false, true);
                    if (rc == 0) { const FList lm = flist(av, fp, hm); if (!differ_walk(c, fref(lm), ListRef{scr.w, scr.a})) upd = false; }
                    else if (S.state == FS_FALLBACK) { it.flags |= FI_DEAD; return; }
                }
                const FList lr = flist(av, fp, hRpr);
                if (hm >= 0) { const FList lm = flist(av, fp, hm); midProb = append_walk(c, fref(lm), fref(lr), rt, rbl); }
                else midProb = append_walk(c, ListRef{scr.w, scr.a}, fref(lr), rt, rbl);
                it.flags |= FI_SCORED;
                if (upd && midProb >= it.pathBest - P.thrOptTopo) {          // M:7293
                    if (hm < 0) hm = fstore(fp, wr);
                    if (hm < 0) { S.state = FS_FALLBACK; it.flags |= FI_DEAD; return; }
                    it.hA = vectUp; it.hB = hBottom; it.hMid = hm; it.recDist = r1.dist; it.flags |= FI_REC_UPD;
                }
            }
            it.midProb = midProb;
            if (upd) it.flags |= FI_UPD_OUT;
            const PRule pr = p_rule(P, scored, midProb, lastLK, it.failsP, it.pathBest);
            if (!pr.go) return;
            if (upT >= 0) {
                int hUp = -1;
                int hrO = hRpr, hrU = hRpr;
                if (upd) {
                    const int r = merge(vectUp, r1.dist, false, hPassed, distance, false, true);
                    if (r == -2) { S.state = FS_FALLBACK; return; }
                    if (r == 0) { hUp = fstore(fp, wr); if (hUp < 0) { S.state = FS_FALLBACK; return; } }
                    if (crossO && hUp >= 0) {                                // into the other child's frame, M:7342-7352
                        hUp = passL(hUp, ro.mutId, false); hrO = passR(hRpr, ro.mutId, false);
                        if (hUp == -2 || !fvalid(hrO)) { S.state = FS_FALLBACK; return; }
                    }
                } else hUp = ftree(it.dir == 1 ? r1.upLeft : r1.upRight);
                if (!fvalid(hUp)) return;
                it.child0 = upd ? fpush(fp, budget, q, true, other, 0, hUp, ro.dist, midProb, pr.fails, hrO, pr.pathBest)
                                : fpush(fp, budget, q, false, other, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, crossO);
                if (upd && hBottom < 0) {                                    // M:7376-7384
                    const int r = merge(hPassed, distance, false, hOther, ro.dist, ro.isTip != 0, false);
                    if (r != 0) return;
                    hBottom = fstore(fp, wr);
                    if (hBottom < 0) { S.state = FS_FALLBACK; return; }
                }
                if (upd && crossU) {                                         // out of t1's frame, M:7388-7395
                    hBottom = passL(hBottom, r1.mutId, true); hrU = passR(hRpr, r1.mutId, true);
                    if (hBottom == -2 || !fvalid(hrU)) { S.state = FS_FALLBACK; return; }
                }
                it.child1 = upd ? fpush(fp, budget, q, true, upT, (int)r1.whichChild + 1, hBottom, r1.dist, midProb, pr.fails, hrU, pr.pathBest)
                                : fpush(fp, budget, q, false, upT, (int)r1.whichChild + 1, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, crossU);
            } else {                                                         // t1 is the root, M:7406-7432
                if (upd) {
                    int hv = rootVector(hPassed, distance, false), hrO = hRpr;
                    if (hv >= 0 && crossO) { hv = passL(hv, ro.mutId, false); hrO = passR(hRpr, ro.mutId, false); }
                    if (hv < 0 || !fvalid(hrO)) { S.state = FS_FALLBACK; return; }
                    it.child0 = fpush(fp, budget, q, true, other, 0, hv, ro.dist, midProb, pr.fails, hrO, pr.pathBest);
                } else
                    it.child0 = fpush(fp, budget, q, false, other, 0, -1, 0.0, midProb, pr.fails, hRpr, pr.pathBest, crossO);
            }
        }
}

// (a real call: k_fr_updating_wave falls back to it from three places and is bound by its LDS, not its registers)
template <bool RV, bool U, bool SS, bool MAT>
__device__ __noinline__ void fr_upd_item_lane_call2(const Ctx<RV, U, SS> &c, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                                                    const FPools &fp, const int budget, const long long laneId, const long long i)
{
    fr_upd_item_lane<RV, U, SS, MAT>(c, av, T, P, fp, budget, laneId, i);
}
template <bool RV, bool U, bool SS>
__device__ __forceinline__ void fr_upd_item_lane_call(const Ctx<RV, U, SS> &c, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                                                      const FPools &fp, const int budget, const long long laneId, const long long i)
{
    if (fp.mat) fr_upd_item_lane_call2<RV, U, SS, true>(c, av, T, P, fp, budget, laneId, i);
    else fr_upd_item_lane_call2<RV, U, SS, false>(c, av, T, P, fp, budget, laneId, i);
}

template <bool RV, bool U, bool SS, bool MAT>
__global__ __launch_bounds__(FR_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_fr_updating(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, SearchParams P, FPools fp, int budget, int heavyMin)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const long long laneId = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long lo = (long long)fp.ctr->loU, hi = (long long)fp.ctr->hiU;
    // wavefronts of items moving down first, then wavefronts of items crawling up (k_fr_sort_level; heavy items are not listed)
    const long long nDown = (long long)fp.ctr->permDown, nUp = (long long)fp.ctr->permUp, padDown = (nDown + 63) & ~63ll;
    // the items with long lists first, 16 to a wavefront (lanes 0-15), then wavefronts of 64 moving down, then of 64 crawling up
    const long long nDownB = (long long)fp.ctr->permDownB, nUpB = (long long)fp.ctr->permUpB;
    const long long vDownB = ((nDownB + 15) >> 4) << 6, vUpB = ((nUpB + 15) >> 4) << 6, vBig = vDownB + vUpB;
    unsigned long long nU = 0, bU = 0;
    (void)heavyMin;
    for (long long v0 = laneId; v0 < vBig + padDown + nUp; v0 += (long long)gridDim.x * blockDim.x) {
        long long i;
        if (v0 < vBig) {
            const bool up = v0 >= vDownB;
            const long long w = up ? v0 - vDownB : v0;
            const int l = (int)(w & 63);
            const long long j = (w >> 6) * 16 + l;
            if (l >= 16 || j >= (up ? nUpB : nDownB)) continue;
            i = lo + (up ? fp.perm2[(hi - lo) - 1 - j] : fp.perm2[j]);
        } else {
            const long long v = v0 - vBig;
            if (v >= nDown && v < padDown) continue;
            i = lo + (v < nDown ? fp.perm[v] : fp.perm[(hi - lo) - 1 - (v - padDown)]);
        }
#ifdef MAPLE_SPR_PROFILE
        const long long t0 = wall_clock64();
        int sz = 0;
        {
            const FItem &it = fp.U[i];
            if (it.dir != 3) {
                const NodeRec &r1 = T.nd[it.t1];
                const int other = it.dir == 0 ? it.t1 : (it.dir == 1 ? r1.c1 : r1.c0);
                const int lw = T.nd[other].lower;
                sz = (lw >= 0 ? av.n_ent[lw] : 0) + (it.hPassed >= 0 ? fp.tn[it.hPassed] : (it.hPassed <= -10 ? av.n_ent[-it.hPassed - 10] : 0));
            }
        }
#endif
        fr_upd_item_lane<RV, U, SS, MAT>(c, av, T, P, fp, budget, laneId, i, &bU);
        nU++;
#ifdef MAPLE_SPR_PROFILE
        {
            const unsigned long long dt = (unsigned long long)(wall_clock64() - t0);
            const int b = sz < 64 ? 0 : sz < 96 ? 1 : sz < 128 ? 2 : sz < 192 ? 3 : sz < 256 ? 4 : sz < 384 ? 5 : sz < 512 ? 6 : 7;
            atomicAdd(&fp.ctr->dbgCnt[b], 1ull); atomicAdd(&fp.ctr->dbgT[b], dt); atomicMax(&fp.ctr->dbgMax[b], dt);
        }
#endif
    }
    // (what the launch did, for the roofline of the bench line: one atomic per wavefront)
    for (int off = 32; off > 0; off >>= 1) {
        nU += ((unsigned long long)(uint32_t)__shfl_down((int)(nU >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)nU, off, 64);
        bU += ((unsigned long long)(uint32_t)__shfl_down((int)(bU >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)bU, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && nU) { atomicAdd(&fp.ctr->itemsU, nU); atomicAdd(&fp.ctr->bytesU, bU); }
}

// ---- the same items by a whole wavefront: the few whose lists are long -------------------------------------------------------
// mergeVectors, areVectorsDifferent and appendProbNode cut along the merge path of the two lists (wave_update.h, wave_dev.h:
// lane d does step d of the walk; bit for bit the one-lane walks), every list of the item in LDS.  An item with a list beyond
// the staging limit is walked by lane 0 alone.
__device__ inline int fstore_wave(const FPools &fp, const unsigned long long *w, const double *a, int n, int na)
{
    const int lane = threadIdx.x & 63;
    unsigned long long id = 0, ow = 0, oa = 0;
    if (lane == 0) {
        id = atomicAdd(&fp.ctr->nLists, 1ull);
        ow = atomicAdd(&fp.ctr->usedW, (unsigned long long)n);
        oa = atomicAdd(&fp.ctr->usedA, (unsigned long long)na);
    }
    auto bc = [](unsigned long long x) {
        return ((unsigned long long)(uint32_t)__shfl((int)(x >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)x, 0, 64);
    };
    id = bc(id); ow = bc(ow); oa = bc(oa);
    if ((long long)id >= fp.capL || (long long)(ow + n) > fp.capW || (long long)(oa + na) > fp.capA) {
        if (lane == 0) fp.ctr->overflow = 1;
        return -2;
    }
    unsigned long long *dw = (unsigned long long *)(fp.tw + ow);
    double *da = fp.ta + oa;
    for (int k = lane; k < n; k += 64) dw[k] = w[k];
    for (int k = lane; k < na; k += 64) da[k] = a[k];
    if (lane == 0) { fp.toffW[id] = (long long)ow; fp.toffA[id] = (long long)oa; fp.tn[id] = n; fp.tna[id] = na; fp.tflag[id] = 0; }
    __threadfence();
    wave_sync();
    return (int)id;
}

template <bool RV, bool U, bool SS>
__global__ __launch_bounds__(64) void k_fr_updating_wave(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, SearchParams P, FPools fp,
                                                         int budget, int heavyMin, long long laneBase)
{
    __shared__ Lds lds;
    __shared__ WaveUpdLds L;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    WaveLds &W = *reinterpret_cast<WaveLds *>(L.baux);                     // (appendProbNode's staging: baux is free by then)
    static_assert(sizeof(WaveLds) <= sizeof(L.baux), "LDS alias");
    const long long lo = (long long)fp.ctr->loU;
    // the level's items that go a wavefront each, listed by k_fr_sort_level (perm3): dealt to the wavefronts one at a time
    const long long nHeavy = (long long)fp.ctr->permHeavy;
    (void)heavyMin;
    for (long long kk = blockIdx.x; kk < nHeavy; kk += gridDim.x) {
        {
            const long long i = lo + fp.perm3[kk];
            FItem &it = fp.U[i];
            FSearch &S = fp.S[it.q];
            if (!fs_live(S.state)) { if (lane == 0) it.flags |= FI_DEAD; continue; }
            const int q = it.q, t1 = it.t1, dir = it.dir;
            const NodeRec r1 = T.nd[t1];
            const int hPassed = it.hPassed, hRpr = it.hRpr;
            const double distance = it.distance, lastLK = it.lastLK;
            const bool rt = S.isRemovedTip != 0;
            const double rbl = S.removedBLen;
            const int other = dir == 0 ? -1 : (dir == 1 ? r1.c1 : r1.c0);
            const int upT = r1.up;
            // every list the item touches must fit the staging areas; else lane 0 walks the item alone
            const FList lp = fvalid(hPassed) ? flist(av, fp, hPassed) : FList{nullptr, nullptr, 0, 0};
            const FList lr = flist(av, fp, hRpr);
            bool fits = fvalid(hPassed) && lp.n <= MAPLE_WU_IN && lr.n <= MAPLE_WAVE_CAPW;
            // (an item next to a MAT reference branch re-expresses lists on the way: lane 0 walks it)
            if (fp.mat && (dir == 3 || r1.c0Frame != r1.frameOf || r1.c1Frame != r1.frameOf || r1.upFrame != r1.frameOf)) fits = false;
            {
                const int ids[6] = {r1.lower, r1.totUp, dir == 0 && r1.c0 >= 0 ? T.nd[r1.c0].lower : -1, dir == 0 && r1.c1 >= 0 ? T.nd[r1.c1].lower : -1,
                                    other >= 0 ? T.nd[other].lower : -1,
                                    (dir != 0 && upT >= 0) ? (r1.whichChild ? T.nd[upT].upLeft : T.nd[upT].upRight) : -1};
                for (int k = 0; k < 6; k++) if (ids[k] >= 0 && av.n_ent[ids[k]] > MAPLE_WU_IN) fits = false;
                // (crawling up, the merged lower list is an input of the next merge)
                if (dir != 0 && other >= 0 && T.nd[other].lower >= 0 && lp.n + av.n_ent[T.nd[other].lower] > MAPLE_WU_IN) fits = false;
            }
            if (!fits) {
                if (lane == 0) fr_upd_item_lane_call(c, av, T, P, fp, budget, laneBase + blockIdx.x, i);
                wave_sync();
                continue;
            }
            bool upd = true;
            double midProb = lastLK;
            int flagsAdd = 0;
            int nM = 0, naM = 0;
            // mergeVectors into L.m / L.maux: 0 ok, -1 None, -2 fatal
            auto merge = [&](const FList &l1, double b1, bool tp1, const FList &l2, double b2, bool tp2, bool upDown) -> int {
                wave_sync();
                const int r = wave_merge(c, fref(l1), l1.n, b1, tp1, fref(l2), l2.n, b2, tp2, upDown, L, naM);
                if (r < 0) return r == -1 ? -1 : -2;
                nM = r;
                return 0;
            };
            auto differs = [&](int listId) -> bool {                        // areVectorsDifferent(L.m, tree list)
                const FList tu = flist(av, fp, ftree(listId));
                const unsigned long long *tw = (const unsigned long long *)tu.w;
                for (int k = lane; k < tu.n; k += 64) L.old[k] = tw[k];
                wave_sync();
                return wave_differ(c, L.m, L.maux, nM, L.old, tu.aux, tu.n);
            };
            auto score = [&]() -> double {                                  // appendProbNode(L.m, removed list)
                wave_sync();
                return wave_append(c, ListRef{(const uint2 *)L.m, L.maux}, nM, fref(lr), lr.n, rt, rbl, W);
            };
            auto push1 = [&](bool u, int node, int d, int h, double dst, double mp, int fails, double pb) -> int {
                int ref = FR_NONE;
                if (lane == 0) ref = fpush(fp, budget, q, u, node, d, h, dst, mp, fails, hRpr, pb);
                return __shfl(ref, 0, 64);
            };
            bool dead = false, fallback = false;
            int child0 = FR_NONE, child1 = FR_NONE, hA = -1, hB = -1, hMid = -1;
            double recDist = 0.0;
            if (dir == 0) {
                const bool scored = !(upT == S.parent || upT < 0) && (r1.dist > P.effNon0 || r1.upIsRoot);
                if (scored) {
                    const FList ll = flist(av, fp, ftree(r1.lower));
                    if (merge(lp, distance / 2, false, ll, distance / 2, r1.isTip != 0, true) != 0) dead = true;
                    else {
                        if (r1.totUp >= 0 && !differs(r1.totUp)) upd = false;
                        midProb = score();
                        flagsAdd |= FI_SCORED;
                        if (upd && midProb >= it.pathBest - P.thrOptTopo) {
                            wave_sync();
                            hMid = fstore_wave(fp, L.m, L.maux, nM, naM);
                            if (hMid < 0) { fallback = true; dead = true; }
                            else { hA = hPassed; hB = ftree(r1.lower); recDist = distance; flagsAdd |= FI_REC_UPD; }
                        }
                    }
                }
                if (!dead) {
                    const PRule pr = p_rule(P, scored, midProb, lastLK, it.failsP, it.pathBest);
                    if (pr.go && r1.c0 >= 0) {
                        for (int k = 0; k < 2 && !fallback; k++) {
                            const int ch = k == 0 ? r1.c0 : r1.c1, oth = k == 0 ? r1.c1 : r1.c0;
                            int ref = FR_NONE;
                            if (upd) {
                                const NodeRec ro = T.nd[oth];
                                const FList lo2 = flist(av, fp, ftree(ro.lower));
                                const int r = ro.lower >= 0 ? merge(lp, distance, false, lo2, ro.dist, ro.isTip != 0, true) : -2;
                                if (r == -2) { fallback = true; break; }
                                if (r == 0) {
                                    wave_sync();
                                    const int hv = fstore_wave(fp, L.m, L.maux, nM, naM);
                                    if (hv < 0) { fallback = true; break; }
                                    ref = push1(true, ch, 0, hv, T.nd[ch].dist, midProb, pr.fails, pr.pathBest);
                                }
                            } else if ((k == 0 ? r1.upRight : r1.upLeft) >= 0)
                                ref = push1(false, ch, 0, -1, 0.0, midProb, pr.fails, pr.pathBest);
                            if (k == 0) child0 = ref; else child1 = ref;
                        }
                    }
                }
            } else {
                const NodeRec ro = T.nd[other];
                const int vectUp = upT >= 0 ? ftree(r1.whichChild ? T.nd[upT].upLeft : T.nd[upT].upRight) : -1;
                const bool scored = upT >= 0 && (r1.dist > P.effNon0 || r1.upIsRoot);
                int hBottom = -1;
                const FList lo2 = flist(av, fp, ftree(ro.lower));
                const FList lvu = fvalid(vectUp) ? flist(av, fp, vectUp) : FList{nullptr, nullptr, 0, 0};
                if (scored) {
                    if (!fvalid(vectUp) || r1.totUp < 0) {                  // (the on-the-spot probVectTotUp of M:7198-7200: one lane)
                        if (lane == 0) fr_upd_item_lane_call(c, av, T, P, fp, budget, laneBase + blockIdx.x, i);
                        wave_sync();
                        continue;
                    }
                    int r = merge(lp, distance, false, lo2, ro.dist, ro.isTip != 0, false);
                    if (r != 0) dead = true;
                    else {
                        wave_sync();
                        hBottom = fstore_wave(fp, L.m, L.maux, nM, naM);
                        if (hBottom < 0) { fallback = true; dead = true; }
                        else {
                            const FList lb = flist(av, fp, hBottom);            // (written by this wavefront, fenced in fstore_wave)
                            r = merge(lvu, r1.dist / 2, false, lb, r1.dist / 2, false, true);
                            if (r != 0) dead = true;
                            else {
                                if (!differs(r1.totUp)) upd = false;
                                midProb = score();
                                flagsAdd |= FI_SCORED;
                                if (upd && midProb >= it.pathBest - P.thrOptTopo) {
                                    wave_sync();
                                    hMid = fstore_wave(fp, L.m, L.maux, nM, naM);
                                    if (hMid < 0) { fallback = true; dead = true; }
                                    else { hA = vectUp; hB = hBottom; recDist = r1.dist; flagsAdd |= FI_REC_UPD; }
                                }
                            }
                        }
                    }
                }
                if (!dead) {
                    const PRule pr = p_rule(P, scored, midProb, lastLK, it.failsP, it.pathBest);
                    if (pr.go) {
                        if (upT >= 0) {
                            int hUp = -1;
                            bool stop = false;
                            if (upd) {
                                const int r = fvalid(vectUp) ? merge(lvu, r1.dist, false, lp, distance, false, true) : -2;
                                if (r == -2) { fallback = true; stop = true; }
                                else if (r == 0) { wave_sync(); hUp = fstore_wave(fp, L.m, L.maux, nM, naM); if (hUp < 0) { fallback = true; stop = true; } }
                            } else hUp = ftree(dir == 1 ? r1.upLeft : r1.upRight);
                            if (!stop && fvalid(hUp)) {
                                child0 = upd ? push1(true, other, 0, hUp, ro.dist, midProb, pr.fails, pr.pathBest)
                                             : push1(false, other, 0, -1, 0.0, midProb, pr.fails, pr.pathBest);
                                if (upd && hBottom < 0) {
                                    const int r = merge(lp, distance, false, lo2, ro.dist, ro.isTip != 0, false);
                                    if (r == 0) { wave_sync(); hBottom = fstore_wave(fp, L.m, L.maux, nM, naM); if (hBottom < 0) fallback = true; }
                                    else stop = true;
                                }
                                if (!stop && !fallback)
                                    child1 = upd ? push1(true, upT, (int)r1.whichChild + 1, hBottom, r1.dist, midProb, pr.fails, pr.pathBest)
                                                 : push1(false, upT, (int)r1.whichChild + 1, -1, 0.0, midProb, pr.fails, pr.pathBest);
                            }
                        } else if (upd) {                                   // t1 is the root and the item still updates: rootVector, one lane
                            // (nothing was pushed or stored yet that the one-lane walk would not redo)
                            if (lane == 0) fr_upd_item_lane_call(c, av, T, P, fp, budget, laneBase + blockIdx.x, i);
                            wave_sync();
                            continue;
                        } else
                            child0 = push1(false, other, 0, -1, 0.0, midProb, pr.fails, pr.pathBest);
                    }
                }
            }
            if (lane == 0) {
                if (fallback) S.state = FS_FALLBACK;
                it.midProb = midProb; it.recDist = recDist; it.child0 = child0; it.child1 = child1; it.hA = hA; it.hB = hB; it.hMid = hMid;
                it.flags |= (uint8_t)(flagsAdd | (upd ? FI_UPD_OUT : 0) | (dead ? FI_DEAD : 0));
            }
            wave_sync();
        }
    }
}

}  // namespace

int fr_launch_updating(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P, const FPools &fp,
                       int budget, int heavyMin)
{
    const bool rv_ = c->dm.useRateVariation, u_ = c->dm.usingErrorRate, ss_ = c->dm.errorRateSiteSpecific;
#define FR_UPD_LAUNCH(RV, U, SS)                                                                                        \
    do {                                                                                                               \
        if (fp.mat) k_fr_updating<RV, U, SS, true><<<grid, FR_BLOCK, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin); \
        else k_fr_updating<RV, U, SS, false><<<grid, FR_BLOCK, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin);       \
    } while (0)
    if (!rv_ && !u_) FR_UPD_LAUNCH(false, false, false);
    else if (rv_ && !u_) FR_UPD_LAUNCH(true, false, false);
    else if (!rv_ && u_ && !ss_) FR_UPD_LAUNCH(false, true, false);
    else if (!rv_ && u_ && ss_) FR_UPD_LAUNCH(false, true, true);
    else if (rv_ && u_ && !ss_) FR_UPD_LAUNCH(true, true, false);
    else FR_UPD_LAUNCH(true, true, true);
#undef FR_UPD_LAUNCH
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}

int fr_launch_updating_wave(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                            const FPools &fp, int budget, int heavyMin, long long laneBase)
{
    FR_DISPATCH3(c, k_fr_updating_wave, <<<grid, 64, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin, laneBase));
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}
