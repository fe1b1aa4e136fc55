// maple_amd/csrc/frontier_dev.h -- what the translation units of the frontier tier share: the item / search / pool records,
// the bump allocators of items and temporary lists, the permissive form of the reference's rules.  frontier.hip holds the
// host side and the small kernels (seeding, cached-regime scoring, layout, replay, refinement, final selection);
// frontier_upd.hip the kernels of the items that still update genome lists (the long compile).
#pragma once
#include "ctx_host.h"
#include "frontier.h"

#include <algorithm>
#include <cstring>

namespace frt {

#define FR_BLOCK 256
#ifndef FR_HEAVY_MULT
#define FR_HEAVY_MULT 24               // (quarters of the mean list length: lists of this many entries together go one wavefront per item)
#endif
#ifndef FR_BIG_MIN
#define FR_BIG_MIN 176
#endif

enum { FI_UPD_IN = 1, FI_UPD_OUT = 2, FI_SCORED = 4, FI_DEAD = 8, FI_REC_UPD = 16, FI_SEED = 32, FI_SEED_EMPTY = 64,
       FI_NEEDPASS = 128 };             // (trees with MAT local references; never cleared on an expanded item) pushed across a reference branch: hRpr is re-expressed by k_fr_pass
// FS_WIDE: a whole-tree search whose row of the dense score table is being made next to this tier.  Its items that still
// update genome lists are expanded here like any other's (they are what made such a search slow for one lane: up to 200
// updating steps in a row); an item that arrives in the cached regime on the way DOWN is left as a seed (FI_SEED) -- the clade
// below it is scanned over the score row by k_fr_replay_wide (wave_scan_clade, search_dev.h) when the exact walk gets there.
enum { FS_ACTIVE = 0, FS_FINAL = 1, FS_OVER = 2, FS_FALLBACK = 3, FS_WIDE = 4 };
__device__ __forceinline__ bool fs_live(int st) { return st == FS_ACTIVE || st == FS_WIDE; }
#define FR_NONE (-1)

// where a list is: offset of its words (low 40 bits) and their number (high 24); the same for its aux doubles
// (x: the words, y: the aux doubles.  The vector type itself, so that a record is stored and loaded as ONE 16-byte access of one
// type: a record stored as a struct and loaded through a cast to ulonglong2 let the compiler move the load of a list's record
// ahead of the store that had just made it -- fpass_removed reads the list fstore returned -- and a walk went off into the blue.)
typedef ulonglong2 LRec;
#define LREC_OFF(x) ((long long)((x) & ((1ull << 40) - 1)))
#define LREC_N(x) ((int32_t)((x) >> 40))
__device__ __host__ __forceinline__ LRec lrec_make(long long offW, int n, long long offA, int na)
{
    LRec r;
    r.x = (unsigned long long)offW | ((unsigned long long)(uint32_t)n << 40);
    r.y = (unsigned long long)offA | ((unsigned long long)(uint32_t)na << 40);
    return r;
}

struct alignas(32) FItem {
    // what the exact replay reads and writes per pop, in one 32-byte sector (the walk of the longest search is a chain of
    // dependent misses on these)
    int8_t dir;                        // 0 = moving from a parent to its child, 1 / 2 = crawling up from child 0 / 1
    uint8_t flags;
    int16_t failsA;                    // failedPasses the replay arrives with
    int32_t next;                      // stack link, then short-list link
    int32_t child0, child1;            // item refs in push order: >= 0 cached pool, <= -2 updating pool -(i + 2), -1 none
    double midProb, lastLK;
    // in (written by the parent item's lane)
    int32_t q, t1;
    int32_t hPassed, hRpr;             // list handles: >= 0 temporary list, <= -10 tree list -(id + 10), -1 None
    double distance, pathBest;
    int16_t failsP, pad;               // failedPasses under the permissive rules
    // out (written by the item's own lane)
    int32_t hA, hB, hMid;              // short-list record of an item that was still updating (M:7073 / 7295)
    double recDist;
    int32_t pad2[2];
};
static_assert(sizeof(FItem) == 96, "FItem");

// The items of a search once more, in the order the exact walk visits them (the child pushed last first): rank p + 1 is the first
// item visited after rank p, p + size skips the subtree -- the walk of k_fr_replay is a forward scan over these 32-byte records
// instead of a chase through 96-byte items scattered over gigabytes (one miss to HBM per pop).
struct alignas(32) FVisit {
    double midProb, lastLK;
    int32_t ref, size;
    int32_t parent;                    // rank of the item that pushed it (its failedPasses are handed on), -1 for a seed
    uint8_t flags; int8_t dir; int16_t failsOut;
};
static_assert(sizeof(FVisit) == 32, "FVisit");

struct FSearch {
    int32_t node, parent, sibling;     // pruned node, its parent (`node` of findBestParentTopology), its sibling
    int32_t hRpr0;
    int32_t seed0, seed1;
    int32_t nItems;                    // expanded so far (atomic)
    int32_t state;
    int32_t slHead, nApp, recBase, recCount;
    int32_t isRemovedTip;
    int32_t rprMerge0;                 // shorten() (M:7087) would change the pruned node's own lower list (never so for a stored list): shorten_would_merge's level
    double removedBLen, curLK;
};

struct FRec { int32_t q, ref; double optimized, top, bottom, app; int32_t ok;
              int32_t hRprS; };        // in: FR_SHORTEN = the reference shortened this record's removed list in place (M:7087); out: the shortened list
#define FR_SHORTEN (-3)
#define FR_FRAMES_EVENT (-4)            // ... and the branch beat the running best when it was visited (M:7087 applies to that list)
#define FR_FRAMES (-2)                  // (a record of k_fr_replay_wide whose removed list is still in its seed's frame: k_fr_wide_frames)

struct FCtr {                          // device-side bookkeeping of the level loop
    // (what every lane READS at the start of a kernel, and each counter the lanes bump, on cache lines of their own)
    alignas(128) unsigned long long loU;
    unsigned long long hiU;            // the current level of the items that still update lists (the stream of k_fr_updating*)
    // The cached-regime items run on a stream of their own, one launch after the other, never waited for by the updating levels:
    // a launch takes what was COMPLETE when it started -- the items earlier cached launches pushed (lower part of the pool:
    // loC .. hiC) and the "roots" pushed by the updating levels that have finished (upper part: loR .. hiR <= safeR).
    alignas(128) unsigned long long loC;
    unsigned long long hiC, loR, hiR, loPR, hiPR;
    alignas(128) unsigned long long safeR;    // roots / their pass entries pushed by kernels that have finished (published by k_fr_snap_u)
    unsigned long long safePassR;
    alignas(128) unsigned long long usedU;   // items allocated in the pools
    alignas(128) unsigned long long usedC;
    alignas(128) unsigned long long usedR;
    alignas(128) unsigned long long nPassR;       // roots pushed across a MAT reference branch (passListR)
    alignas(128) unsigned long long bigUsedC;     // shared scratch of k_fr_pass (reset by every cached launch)
    // An item that was pushed across a reference branch is scored one launch LATER than the launch that holds it: its removed
    // list is re-expressed by k_fr_pass on a stream of its own, next to the launch's k_fr_cached, and the item is handed to the
    // next launch through `deferred` (k_fr_cached passes over such an item where it finds it in its ranges).
    alignas(128) unsigned long long nDeferred;
    unsigned long long loD, hiD;
    alignas(128) unsigned long long permDown;     // the level's one-lane updating items by direction (k_fr_sort_level)
    alignas(128) unsigned long long permUp;
    alignas(128) unsigned long long permDownB;    // ... those with long lists (16 to a wavefront)
    alignas(128) unsigned long long permUpB;
    alignas(128) unsigned long long permHeavy;    // ... those that go a wavefront each, lists of <= 128 entries (k_fr_updating_wave_s)
    alignas(128) unsigned long long permHeavy2;   // ... with longer lists (k_fr_updating_wave: one wavefront per compute unit)
    alignas(128) unsigned long long nLists;       // temporary lists
    alignas(128) unsigned long long usedW;
    alignas(128) unsigned long long usedA;
    alignas(128) unsigned long long nRecs;
    alignas(128) unsigned long long nPass;        // cached-regime items pushed across a MAT reference branch (passList)
    unsigned long long loP, hiP;       // ... those of the current level
    unsigned long long bigUsed;        // entries taken from the shared scratch of over-long lists (reset every level)
    alignas(128) unsigned long long itemsU;
    unsigned long long bytesU;         // items k_fr_updating walked and the bytes of the lists their mergeVectors read and wrote
    alignas(128) unsigned long long scoredC;
    unsigned long long bytesC;         // cached-regime items scored by k_fr_cached and their SURVEY 8d bytes (8 E + 8 A + 8)
    int32_t overflow, nLevels, nLevelsC;
    int32_t fbReason[8];               // searches handed to the one-lane kernel by the exact walk, by reason (k_fr_replay; printed with verbose)
#ifdef MAPLE_SPR_PROFILE
    unsigned long long dbgCnt[8], dbgT[8], dbgMax[8];   // k_fr_updating's one-lane items by size (entries of the two lists): count, ticks, slowest
    unsigned long long dbgC[8];        // k_fr_cached per wavefront-iteration: iterations, ticks before the walk (item, search, node, list table),
                                       // ticks of the walk, ticks after it (rule, pushes), the longest lane's entries (two lists), scored lanes
    unsigned long long dbgWCnt[4], dbgWT[4], dbgWMax[4]; // the wavefront-wide items: small class, 512 class, walked by lane 0 alone (small / 512)
#endif
};

struct FPools {
    FItem *U, *C;
    long long capU, capC;
    long long capCC;                   // the cached pool's lower part [0, capCC): items pushed by cached-regime items; [capCC, capC): roots
    // temporary lists
    uint2 *tw; double *ta;
    // ONE 16-byte record per list -- where its words and its aux doubles begin and how many there are -- instead of four arrays:
    // an item reads the table for two or three lists, and four gathers from four arrays were four sectors of four different
    // cache lines per list.  trec: the batch's temporary lists; arec: the same for the genome-list arena, made from its four
    // arrays at the start of a call (k_fr_arena_recs) for the nArec lists the arena held then.
    LRec *trec;
    const LRec *arec; int32_t nArec;
    uint8_t *tflag;                    // per temporary list: a removed list that shorten() (M:7087) would change (shorten_would_merge's level)
    long long capW, capA, capL;
    FVisit *visit; long long capVisit;   // the layout of k_fr_layout_* (null: k_fr_replay chases the items)
    int32_t *lsize, *lpos, *lpar;      // per item (updating pool first, then the cached pool): items in the subtree it heads, its
                                       // rank in its search's visiting order (-1: not laid out), the rank of the item that pushed it
    unsigned long long *lvl;           // [maxLevels][4]: loU, hiU, loC, hiC of every level
    int32_t maxLevels;
    int32_t *tot; long long *vbase;    // per search: items in its two seed subtrees, and where its visiting order starts
    int32_t *perm, *perm2;             // the level's one-lane updating items: moving down from the front, crawling up from the back
                                       // (perm2: those with long lists)
    int32_t *perm3, *perm4;            // the level's items walked by a wavefront each: the two size classes
    int32_t waveAllBelow;              // a level with at most this many updating items: all of them by wavefronts
    // per-lane scratch
    uint2 *sw; double *sa; double *sais;
    int32_t capE;                      // entries one lane's scratch list takes (aux: 5 per entry; ais: 2 per entry)
    uint2 *bw; double *ba;             // shared scratch for the few lists longer than that (bump-allocated, reset every level)
    long long capBig;
    int32_t bigSide;                   // 0: the updating levels' region and counter; 1: k_fr_pass's own (it runs next to them)
    FCtr *ctr;
    FSearch *S;
    FRec *recs;
    long long capRecs;
    // trees with MAT local references (M:8296-8354): the lists of a node are written relative to the reference of its frame, and a
    // search that crosses the branch above a reference node re-expresses what it carries (passGenomeListThroughBranch,
    // M:6844-6847, 7111-7118, 7148-7155, 7359-7366, 7388-7395)
    int32_t mat;                       // 1: the tree has reference nodes
    MutViewS mv;
    int32_t *passList; long long capPass;   // refs of the cached-pool items whose removed list is re-expressed at the start of their level
    int32_t *passListR; long long capPassR; // ... of the roots among them
    int32_t *deferred; long long capDeferred;   // refs of the items whose removed list has been re-expressed: the next launch scores them
};

__device__ __forceinline__ FItem &item_of(const FPools &fp, int ref) { return ref >= 0 ? fp.C[ref] : fp.U[-(ref + 2)]; }

struct FList { const uint2 *w; const double *aux; int32_t n, na; };

__device__ __forceinline__ bool fvalid(int h) { return h >= 0 || h <= -10; }
__device__ __forceinline__ int ftree(int listId) { return listId < 0 ? -1 : -(listId + 10); }
__device__ __forceinline__ FList flist(const ArenaViewS &av, const FPools &fp, int h)
{
    if (h >= 0) { const LRec r = fp.trec[h]; return FList{fp.tw + LREC_OFF(r.x), fp.ta + LREC_OFF(r.y), LREC_N(r.x), LREC_N(r.y)}; }
    const int id = -h - 10;
    if (id < fp.nArec) { const LRec r = fp.arec[id]; return FList{av.words + LREC_OFF(r.x), av.aux + LREC_OFF(r.y), LREC_N(r.x), LREC_N(r.y)}; }
    return FList{av.words + av.ent_off[id], av.aux + av.aux_off[id], av.n_ent[id], av.n_aux[id]};   // (a list the arena got during the call)
}
// the number of entries of a list alone
__device__ __forceinline__ int flen(const ArenaViewS &av, const FPools &fp, int h)
{
    if (h >= 0) return LREC_N(fp.trec[h].x);
    const int id = -h - 10;
    return id < fp.nArec ? LREC_N(fp.arec[id].x) : av.n_ent[id];
}
__device__ __forceinline__ ListRef fref(const FList &l) { return ListRef{l.w, l.aux}; }

// room for one list of up to `need` entries: the lane's own slab, or -- for the few lists near the root that are longer --
// a piece of the shared scratch (false: none left)
struct FScr { uint2 *w; double *a; };
__device__ inline bool fscratch(const FPools &fp, long long laneId, int need, FScr &o)
{
    if (need <= fp.capE) { o.w = fp.sw + laneId * fp.capE; o.a = fp.sa + laneId * 5ll * fp.capE; return true; }
    const unsigned long long off = atomicAdd(fp.bigSide ? &fp.ctr->bigUsedC : &fp.ctr->bigUsed, (unsigned long long)need);
    if ((long long)(off + need) > fp.capBig) return false;
    o.w = fp.bw + off; o.a = fp.ba + 5ull * off;
    return true;
}

// a scratch list becomes a temporary list of the batch: exact room, one copy; -2 when the pools are full
__device__ inline int fstore(const FPools &fp, const Writer &wr)
{
    // The lanes of a wavefront that are here together take their room with ONE atomic per counter: the three counters share a
    // cache line with everything else the level loop counts, and a million single-lane atomics per level on it were what a
    // level of k_fr_updating waited for.
    const unsigned long long act = __ballot(1);
    const int lane = threadIdx.x & 63, leader = (int)__ffsll((long long)act) - 1;
    int preN = 0, preA = 0, totN = 0, totA = 0, rank = 0, cnt = 0;
    for (unsigned long long mm = act; mm; mm &= mm - 1) {
        const int j = (int)__ffsll((long long)mm) - 1;
        const int nj = __builtin_amdgcn_readlane(wr.n, j), aj = __builtin_amdgcn_readlane(wr.na, j);
        if (j < lane) { preN += nj; preA += aj; rank++; }
        totN += nj; totA += aj; cnt++;
    }
    unsigned long long id0 = 0, ow0 = 0, oa0 = 0;
    if (lane == leader) {
        id0 = atomicAdd(&fp.ctr->nLists, (unsigned long long)cnt);
        ow0 = atomicAdd(&fp.ctr->usedW, (unsigned long long)totN);
        oa0 = atomicAdd(&fp.ctr->usedA, (unsigned long long)totA);
    }
    auto bc = [&](unsigned long long x) {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(x >> 32), leader) << 32)
               | (uint32_t)__builtin_amdgcn_readlane((int)x, leader);
    };
    const unsigned long long id = bc(id0) + (unsigned long long)rank, ow = bc(ow0) + (unsigned long long)preN,
                             oa = bc(oa0) + (unsigned long long)preA;
    if ((long long)id >= fp.capL || (long long)(ow + wr.n) > fp.capW || (long long)(oa + wr.na) > fp.capA) {
        fp.ctr->overflow = 1;
        return -2;
    }
    uint2 *dw = fp.tw + ow;
    double *da = fp.ta + oa;
    for (int k = 0; k < wr.n; k++) dw[k] = wr.w[k];
    for (int k = 0; k < wr.na; k++) da[k] = wr.aux[k];
    fp.trec[id] = lrec_make((long long)ow, wr.n, (long long)oa, wr.na); fp.tflag[id] = 0;
    return (int)id;
}

__device__ inline int wave_excl_sum_f(int v, int lane, int &total)
{
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        incl += (lane >= off) ? t : 0;
    }
    total = __shfl(incl, 63);
    return incl - v;
}

// would shorten() (M:3721-3745) change this list?  (the absorb test of shorten_walk, genome_dev.h: a run of R entries of one
// kind collapses into its LAST entry, each candidate compared with the run's FIRST entry)
//   0  no;
//   1  yes, and every entry that goes away has exactly the tail of the entry that stays (none at all, or the same doubles and
//      flag): no appendProbNode can tell the two forms apart -- an R entry only enters a site factor through its tail;
//   2  yes, with tails that are equal only within the tolerance: the merged list is another list to every reader.
template <class C> __device__ inline int shorten_would_merge(const C &c, ListRef L, int nEnt)
{
    const double thr = c.m.thresholdProb;
    Cursor a;
    a.init(L);
    Ent head = a.e;
    int level = 0;
    bool exact = true;                                                      // (of the current run: every absorbed tail == the head's so far)
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
        if (absorb) {
            // (the run ends up with the LAST entry's tail: all its tails must be the same doubles for the merge to be invisible)
            if (nw.hasD0 && (nw.d0 != head.d0 || (nw.hasD1 && nw.d1 != head.d1))) exact = false;
            level = max(level, exact ? 1 : 2);
        } else { head = nw; exact = true; }
    }
    return level;
}

// passGenomeListThroughBranch (M:3749-3877) of list h through mutation list mutId: the handle of a new temporary list, h itself
// when the branch carries no mutations, -2 when there is no room.  (h must be a stored list: the lane's scratch is written.)
__device__ inline int fpass_store(const FPools &fp, const ArenaViewS &av, const int lRef, const long long laneId, const int h, const int mutId,
                                  const bool dirUp)
{
    if (!fvalid(h) || mutId < 0) return h;
    const int cnt = fp.mv.cnt[mutId];
    if (cnt == 0) return h;
    const FList l = flist(av, fp, h);
    FScr scr{nullptr, nullptr};
    if (!fscratch(fp, laneId, l.n + 2 * cnt, scr)) return -2;
    Writer wr;
    wr.init(scr.w, scr.a);
    pass_walk(lRef, fref(l), fp.mv.mut3 + 3 * fp.mv.off[mutId], cnt, dirUp, wr);
    return fstore(fp, wr);
}
// ... of the REMOVED list.  The reference shortens that list IN PLACE whenever a branch on the way down beats the running best
// (M:7087), and everything made from it afterwards sees the shortened form.  The expansion never shortens; a list that
// shorten() would change is only MARKED here, and the exact walk (k_fr_replay) hands a search to the one-lane kernel -- which
// does what the reference does -- if such a list is ever held by a branch that beats the running best.  (Beating the cost of
// the current placement is rare; a marked list is not: a list re-expressed in a frame that shares a mutation with it gains an
// R entry next to two others.)
template <class C>
__device__ inline int fpass_removed(const C &c, const FPools &fp, const ArenaViewS &av, const long long laneId, const int h, const int mutId,
                                    const bool dirUp)
{
    const int r = fpass_store(fp, av, c.m.lRef, laneId, h, mutId, dirUp);
    if (r == h || r < 0) return r;                                          // (unchanged, or -2: no room)
    const FList l = flist(av, fp, r);
    fp.tflag[r] = (uint8_t)shorten_would_merge(c, fref(l), l.n);
    return r;
}
// ---- passGenomeListThroughBranch (M:3749-3877) by a whole WAVEFRONT: the handle of a new temporary list, h itself when the
// branch carries no mutations, -2 when there is no room.  Every lane calls with the same arguments and gets the result.
// The walk of pass_walk (genome_dev.h) needs nothing from the entries before but the position reached, and every packed entry
// carries its last position: lane i takes entry i, finds the branch's mutations inside it (two binary searches over the sorted
// mutation list), counts what it writes -- an N entry or a single-site entry one entry; a reference run one entry per mutated
// site plus the stretches between them -- and, after a prefix sum over the wavefront, writes them where they belong in the new
// list.  The list is written straight into the temporary lists' pool (room for the most a pass can add: two entries per
// mutation; what is not used stays empty).  One lane takes ~1.2 us per entry (0.1-0.4 ms per list, up to four lists per item
// next to a reference branch: the slowest wavefront-wide items, 1.4 ms, and the 0.35 ms floor of every k_fr_pass launch).
// `removed`: the list is a removed list -- graded for the in-place shorten() of M:7087 (tflag, see shorten_would_merge).
template <class C>
__device__ inline int wave_pass(const C &c, const FPools &fp, const ArenaViewS &av, const int h, const int mutId, const bool dirUp, const bool removed)
{
    if (!fvalid(h) || mutId < 0) return h;
    const int cnt = fp.mv.cnt[mutId];
    if (cnt == 0) return h;
    const int lane = threadIdx.x & 63;
    const FList l = flist(av, fp, h);
    const int32_t *mut = fp.mv.mut3 + 3 * fp.mv.off[mutId];
    const unsigned long long *lw = (const unsigned long long *)l.w;
    auto bc = [](unsigned long long x) {
        return ((unsigned long long)(uint32_t)__shfl((int)(x >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)x, 0, 64);
    };
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // room for the most the pass can write: an entry per mutated site and one stretch before it (+ one after the last), each
    // piece of a run with the run's tail (<= 2 doubles)
    const int capN = l.n + 2 * cnt, capA = l.na + 4 * cnt;
    unsigned long long id = 0, ow = 0, oa = 0;
    if (lane == 0) {
        id = atomicAdd(&fp.ctr->nLists, 1ull);
        ow = atomicAdd(&fp.ctr->usedW, (unsigned long long)capN);
        oa = atomicAdd(&fp.ctr->usedA, (unsigned long long)capA);
    }
    id = bc(id); ow = bc(ow); oa = bc(oa);
    if ((long long)id >= fp.capL || (long long)(ow + capN) > fp.capW || (long long)(oa + capA) > fp.capA) {
        if (lane == 0) fp.ctr->overflow = 1;
        return -2;
    }
    uint2 *dw = fp.tw + ow;
    double *da = fp.ta + oa;
    auto lower = [&](int pos) {                                             // mutations before position `pos`
        int lo = 0, hi = cnt;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mut[3 * mid] < pos) lo = mid + 1; else hi = mid; }
        return lo;
    };
    int nOut = 0, nAux = 0;
    for (int base = 0; base < l.n; base += 64) {
        const int i = base + lane;
        int cntOut = 0, auxOut = 0, a = 0, b = 0, prevEnd = 0, tail = 0;
        Ent e;
        e.type = 5; e.pos = 0; e.ref = 0; e.hasD0 = e.hasD1 = e.flag = false; e.d0 = e.d1 = 0.0; e.vec = nullptr;
        if (i < l.n) {
            decode_word(lw[i], l.aux, e);
            prevEnd = i > 0 ? (int)(uint32_t)lw[i - 1] : 0;
            a = lower(prevEnd + 1); b = lower(e.pos + 1);                    // the mutations at prevEnd < position <= e.pos
            tail = (e.hasD0 ? 1 : 0) + (e.hasD1 ? 1 : 0);
            if (e.type == 5) cntOut = 1;                                     // (mutations inside a run of N are passed over)
            else if (e.type == 4) {
                int last = prevEnd;
                for (int k = a; k < b; k++) { const int mp = mut[3 * k]; cntOut += (mp > last + 1) ? 2 : 1; last = mp; }
                if (last < e.pos) cntOut++;
                auxOut = cntOut * tail;
            } else { cntOut = 1; auxOut = tail + (e.type == 6 ? 4 : 0); }
        }
        int totN, totA;
        int idx = nOut + wave_excl_sum_f(cntOut, lane, totN);
        int ao = nAux + wave_excl_sum_f(auxOut, lane, totA);
        if (i < l.n) {
            auto put = [&](int type, int pos, int ref, bool tails, bool vec) {
                const uint32_t meta = (uint32_t)type | ((uint32_t)ref << 3) | ((tails && e.hasD0) ? 1u << 5 : 0u) | ((tails && e.hasD1) ? 1u << 6 : 0u)
                                      | ((tails && e.flag) ? 1u << 7 : 0u) | ((uint32_t)ao << 8);
                dw[idx++] = make_uint2((uint32_t)pos, meta);
                if (tails && e.hasD0) da[ao++] = e.d0;
                if (tails && e.hasD1) da[ao++] = e.d1;
                if (vec) { da[ao] = e.vec[0]; da[ao + 1] = e.vec[1]; da[ao + 2] = e.vec[2]; da[ao + 3] = e.vec[3]; ao += 4; }
            };
            if (e.type == 5) put(5, e.pos, 0, false, false);
            else if (e.type == 4) {
                // split the reference run around mutated sites; each mutated site becomes an explicit nucleotide
                int last = prevEnd;
                for (int k = a; k < b; k++) {
                    const int mp = mut[3 * k], from = mut[3 * k + 1], to = mut[3 * k + 2];
                    if (mp > last + 1) put(4, mp - 1, 0, true, false);
                    put(dirUp ? to : from, mp, dirUp ? from : to, true, false);
                    last = mp;
                }
                if (last < e.pos) put(4, e.pos, 0, true, false);
            } else if (a < b) {                                              // a single-site entry on a mutated site
                const int newRef = dirUp ? mut[3 * a + 1] : mut[3 * a + 2];
                if (e.type == 6) put(6, e.pos, newRef, true, true);
                else if (e.type == newRef) put(4, e.pos, 0, true, false);    // equals the new reference -> R
                else put(e.type, e.pos, newRef, true, false);
            } else put(e.type, e.pos, e.ref, true, e.type == 6);
        }
        nOut += totN; nAux += totA;
    }
    __threadfence();
    sync();
    int level = 0;
    if (removed) {
        // would shorten() (M:7087) change the new list?  Neighbouring R entries of one kind without tails: yes, exactly (level 1);
        // with tails the answer depends on the run's first entry: lane 0 walks the list as shorten_would_merge does
        bool plain = false, tails = false;
        for (int j = 1 + lane; j < nOut; j += 64) {
            const uint32_t m1 = dw[j].y, m0 = dw[j - 1].y;
            if ((m1 & 7u) == 4u && (m0 & 7u) == 4u && ((m1 ^ m0) & 0x60u) == 0u) { if (m1 & 0x20u) tails = true; else plain = true; }
        }
        if (__ballot(tails)) {
            if (lane == 0) level = shorten_would_merge(c, ListRef{dw, da}, nOut);
            level = __shfl(level, 0, 64);
        } else level = __ballot(plain) ? 1 : 0;
    }
    if (lane == 0) {
        fp.trec[id] = lrec_make((long long)ow, nOut, (long long)oa, nAux); fp.tflag[id] = (uint8_t)level;
    }
    __threadfence();
    sync();
    return (int)id;
}

__device__ __forceinline__ int frpr_marked(const FPools &fp, const FSearch &S, const int h)
{
    return h >= 0 ? (int)fp.tflag[h] : (h <= -10 ? S.rprMerge0 : 0);
}

// one more expanded item of search q; false when the search is over its budget (it becomes a dense-tier search) or the pool
// is full (the search is handed back)
__device__ inline int fpush(const FPools &fp, const int budget, const int q, const bool upd, const int t1, const int dir,
                            const int hPassed, const double distance, const double lastLK, const int fails, const int hRpr,
                            const double pathBest, const bool needPass = false, const bool fromC = false)
{
    FSearch &S = fp.S[q];
    if (S.state != FS_WIDE && atomicAdd(&S.nItems, 1) >= budget) { S.state = FS_OVER; return FR_NONE; }
    FItem *it;
    int ref;
    // one atomic per wavefront and pool: the lanes that are here together take consecutive items
    // (fromC: the pusher is a cached-regime item -- the lower part of the cached pool; everybody else's cached-regime pushes are
    // roots: the upper part.  Uniform over a kernel.)
    unsigned long long *ctrp = upd ? &fp.ctr->usedU : (fromC ? &fp.ctr->usedC : &fp.ctr->usedR);
    const unsigned long long act = __ballot(1);
    const int lane = threadIdx.x & 63, leader = (int)__ffsll((long long)act) - 1;
    const unsigned long long same = __ballot(upd);                           // (lanes pushing into the updating pool)
    const unsigned long long mine = upd ? (act & same) : (act & ~same);
    const int lead2 = (int)__ffsll((long long)mine) - 1;
    unsigned long long base = 0;
    if (lane == lead2) base = atomicAdd(ctrp, (unsigned long long)__popcll(mine));
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), lead2, 64) << 32) | (uint32_t)__shfl((int)base, lead2, 64);
    (void)leader;
    const unsigned long long i = base + (unsigned long long)__popcll(mine & ((1ull << lane) - 1ull));
    if (upd) {
        if ((long long)i >= fp.capU) { S.state = FS_FALLBACK; fp.ctr->overflow = 1; return FR_NONE; }
        it = &fp.U[i]; ref = -((int)i + 2);
    } else if (fromC) {
        if ((long long)i >= fp.capCC) { S.state = FS_FALLBACK; fp.ctr->overflow = 1; return FR_NONE; }
        it = &fp.C[i]; ref = (int)i;
    } else {
        if ((long long)i >= fp.capC - fp.capCC) { S.state = FS_FALLBACK; fp.ctr->overflow = 1; return FR_NONE; }
        it = &fp.C[fp.capCC + i]; ref = (int)(fp.capCC + i);
    }
    it->q = q; it->t1 = t1; it->dir = (int8_t)dir; it->flags = upd ? FI_UPD_IN : 0; it->failsP = (int16_t)fails;
    it->hPassed = hPassed; it->hRpr = hRpr; it->distance = distance; it->lastLK = lastLK; it->pathBest = pathBest;
    it->child0 = it->child1 = FR_NONE; it->hA = it->hB = it->hMid = -1; it->next = FR_NONE; it->failsA = 0;
    it->midProb = lastLK; it->recDist = 0.0;
    if (needPass) {                                                         // (rare: ~1 push in 100 crosses a reference branch)
        it->flags |= FI_NEEDPASS;
        const unsigned long long k = atomicAdd(fromC ? &fp.ctr->nPass : &fp.ctr->nPassR, 1ull);
        if ((long long)k >= (fromC ? fp.capPass : fp.capPassR)) { S.state = FS_FALLBACK; fp.ctr->overflow = 1; }
        else (fromC ? fp.passList : fp.passListR)[k] = ref;
    }
    return ref;
}

// The permissive form of the reference's rules after an item was scored (M:7083-7103 / 7306-7323): failedPasses under
// pathBest, the new pathBest, and whether the item's relatives are pushed.
struct PRule { int fails; double pathBest; bool go; };
__device__ __forceinline__ PRule p_rule(const SearchParams &P, bool scored, double midProb, double lastLK, int fails, double pathBest)
{
    PRule r;
    r.fails = fails; r.pathBest = pathBest;
    if (scored) {
        if (midProb > pathBest) { r.pathBest = midProb; r.fails = 0; }
        else if (midProb < (lastLK - P.thrConsec)) r.fails = fails + 1;
    }
    // (the reference tests against the running best AFTER it took this score into account: pathBest is updated first, too)
    const bool within = midProb > (r.pathBest - P.thrLKtopology);
    r.go = P.strict ? (r.fails <= P.allowedFails && within) : (r.fails <= P.allowedFails || within);
    return r;
}

// Is this item one of the few with long lists (near the root)?  One lane walking two lists of several hundred entries takes
// milliseconds, and a level of the expansion lasts as long as its slowest item: those go to k_fr_updating_wave, a wavefront
// per item.
__device__ __forceinline__ int fr_upd_size(const ArenaViewS &av, const DevTree &T, const FPools &fp, const FItem &it)
{
    if (it.dir == 3) return 0;
    const NodeRec &r1 = T.nd[it.t1];
    const int other = it.dir == 0 ? it.t1 : (it.dir == 1 ? r1.c1 : r1.c0);
    const int lw = T.nd[other].lower;
    const int nTree = lw >= 0 ? flen(av, fp, ftree(lw)) : 0;
    const int nPass = fvalid(it.hPassed) ? flen(av, fp, it.hPassed) : 0;
    return nPass + nTree;
}
__device__ __forceinline__ bool fr_upd_heavy(const ArenaViewS &av, const DevTree &T, const FPools &fp, const FItem &it, int heavyMin)
{
    return heavyMin > 0 && it.dir != 3 && fr_upd_size(av, T, fp, it) >= heavyMin;
}
// Does a wavefront-wide walk with staging areas for input lists of wuIn entries (and capW entries for appendProbNode's two
// lists) take this item?  Every list the item touches must fit, with room for what a pass through a MAT reference branch adds.
__device__ inline bool fr_wave_fits(const ArenaViewS &av, const DevTree &T, const FPools &fp, const FItem &it, const int wuIn, const int capW)
{
    const int dir = it.dir;
    if (dir == 3 || !fvalid(it.hPassed)) return false;
    const NodeRec r1 = T.nd[it.t1];
    const int nP = flen(av, fp, it.hPassed), nR = flen(av, fp, it.hRpr);
    if (nP > wuIn || nR > capW) return false;
    const int other = dir == 0 ? -1 : (dir == 1 ? r1.c1 : r1.c0);
    const int upT = r1.up;
    // (next to a MAT reference branch a list is re-expressed before it is staged: passGenomeListThroughBranch adds at most two
    // entries per mutation of the branch)
    const bool mat = fp.mat && (r1.c0Frame != r1.frameOf || r1.c1Frame != r1.frameOf || r1.upFrame != r1.frameOf);
    auto grow = [&](int mutId) { return (mat && mutId >= 0) ? 2 * fp.mv.cnt[mutId] : 0; };
    const int g0 = (dir == 0 && r1.c0 >= 0) ? grow(T.nd[r1.c0].mutId) : 0, g1 = (dir == 0 && r1.c1 >= 0) ? grow(T.nd[r1.c1].mutId) : 0;
    const int gO = other >= 0 ? grow(T.nd[other].mutId) : 0, gU = (dir != 0 && upT >= 0) ? grow(r1.mutId) : 0;
    const int ids[6] = {r1.lower, r1.totUp, dir == 0 && r1.c0 >= 0 ? T.nd[r1.c0].lower : -1, dir == 0 && r1.c1 >= 0 ? T.nd[r1.c1].lower : -1,
                        other >= 0 ? T.nd[other].lower : -1,
                        (dir != 0 && upT >= 0) ? (r1.whichChild ? T.nd[upT].upLeft : T.nd[upT].upRight) : -1};
    const int add[6] = {0, 0, g0, g1, gO, gU};
    for (int k = 0; k < 6; k++) if (ids[k] >= 0 && av.n_ent[ids[k]] + add[k] > wuIn) return false;
    // (crawling up, the merged lower list is an input of the next merge)
    if (dir != 0 && ids[4] >= 0 && nP + av.n_ent[ids[4]] + gO > wuIn) return false;
    return true;
}
#define FR_WAVE_SMALL_IN 128           // the small class of the wavefront-wide items: staging for 128-entry lists (27 KB of LDS per
#define FR_WAVE_SMALL_CAPW 256         // wavefront, five wavefronts per compute unit); the rest: 512 entries, one per compute unit

// One lane takes milliseconds for an item whatever the GPU is doing, and a level waits for its slowest item: a level with few
// items -- every level past the first ten, where a few thousand searches near the root are still updating lists -- is walked by
// wavefronts altogether (~0.1 ms per item, a few hundred at a time).
__device__ __forceinline__ int fr_level_heavy_min(const FPools &fp, int heavyMin)
{
    const long long n = (long long)(fp.ctr->hiU - fp.ctr->loU);
    return (heavyMin > 0 && n <= fp.waveAllBelow) ? 1 : heavyMin;
}

}  // namespace frt

#define FR_DISPATCH3(c, KERNEL, ...)                                                                       \
    do {                                                                                                  \
        const bool rv_ = (c)->dm.useRateVariation, u_ = (c)->dm.usingErrorRate, ss_ = (c)->dm.errorRateSiteSpecific; \
        if (!rv_ && !u_) KERNEL<false, false, false> __VA_ARGS__;                                          \
        else if (rv_ && !u_) KERNEL<true, false, false> __VA_ARGS__;                                       \
        else if (!rv_ && u_ && !ss_) KERNEL<false, true, false> __VA_ARGS__;                               \
        else if (!rv_ && u_ && ss_) KERNEL<false, true, true> __VA_ARGS__;                                 \
        else if (rv_ && u_ && !ss_) KERNEL<true, true, false> __VA_ARGS__;                                 \
        else KERNEL<true, true, true> __VA_ARGS__;                                                         \
    } while (0)

// the level kernels of frontier_upd.hip, queued on stream s (kernels of another translation unit cannot be launched directly)
__attribute__((visibility("hidden")))
int fr_launch_updating(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P, const frt::FPools &fp,
                       int budget, int heavyMin);
__attribute__((visibility("hidden")))
int fr_launch_updating_wave(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                            const frt::FPools &fp, int budget, int heavyMin, long long laneBase);
__attribute__((visibility("hidden")))        // (frontier_updw128.hip: the small size class)
int fr_launch_updating_wave_small(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                                  const frt::FPools &fp, int budget, int heavyMin, long long laneBase);
