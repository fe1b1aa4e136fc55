// Batched placement of many query samples on one frozen tree: the depth-first traversal of
// findBestParentForNewSample (MAPLEv0.7.5.4.py:7912-8100) replayed on the device, one lane per query, over scores that
// a query-major appendProbNode launch has already produced for EVERY candidate branch (a superset of what the
// traversal visits).  Nothing here touches genome lists: it is pure control flow over int/f64 columns, bit-exact.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "search_dev.h"

namespace maple {

struct PlaceParams {
    double thrLK;                      // thresholdLogLK x log(lRef), M:3613
    double thrOpt;                     // thresholdLogLKoptimization x log(lRef), M:3610
    double thrConsec;                  // thresholdLogLKconsecutivePlacement, M:63
    int32_t allowedFails;              // M:50
    int32_t strict;                    // strictStopRules, M:56
    // computePlacementSupportOnly=True (M:7940, 7986, 8109): a leaf the query is a minor sequence of does not end the
    // search, and the short list is kept down to max(thresholdLogLKoptimization, thresholdLogLKoptimizationTopology)
    int32_t supportOnly;
    double thrFilter;                  // what the final filter of the short list keeps: score >= best - thrFilter
};

#define MAPLE_PLACE_SHORTLIST 128      // entries within thresholdLogLKoptimization of the best, per query

struct PlaceOut {
    int32_t *status;                   // 0 searched, 1 query is a minor sequence of minorNode, -6 short list overflow
    int32_t *minorNode;
    int32_t *bestNode;                 // before the short-list refinement
    double *bestLK, *originalLK;
    int32_t *nAppend, *missed, *nShort;
    int32_t *slNode;                   // [nQ][MAPLE_PLACE_SHORTLIST]
    double *slLK;
    uint8_t *slShort;                  // was the query list of that node's frame shortened (M:8066) by the end of the walk
    uint8_t *bestShort;
    uint32_t *fromBits;                // one bit per MAT frame: the frame's query list was made (passGenomeListThroughBranch at the
                                       // push of the frame's node, M:8084-8092) from a parent-frame list that HAD been shortened
                                       // by then -- laid out like the frame bits; may be null
};

// The tree in the traversal's own depth-first order (the child pushed last, child 1, first): rank r+1 is the first child
// visited after rank r, and r + size skips r's clade.  A descent is then a forward scan over this array (sequential
// loads that can be issued ahead) instead of a chase through child pointers, and the LIFO stack of the reference
// becomes one (lastLK, fails) slot per depth: the state a node leaves for its children.
struct alignas(32) ScanRec {
    int32_t node;                      // external node id
    int32_t size;                      // nodes in the clade rooted here (itself included)
    int32_t depth;                     // root = 0
    int32_t candCol;                   // column in the score matrix or -1 (M:8049: dist > effectivelyNon0BLen, up != None)
    int32_t leafCol;                   // column in the minor-sequence matrix; >= 0 exactly for leaves
    int32_t frame;                     // MAT reference frame of the node
    int32_t childFrame[2];             // frames of the two children where they differ from the node's own, else -1
};

// The traversal of ONE query over the scan array.  Lane q of nQ keeps element i of its per-depth state / frame bits at
// [i * nQ + q] (coalesced when a wavefront runs 64 queries; the host calls it with nQ = 1, q = 0 on plain arrays for
// very small batches, where one lane's latency would dominate the call).  frameOf[v] = frame of node v.
__host__ __device__ inline void place_replay_one(const ScanRec *R, int nReach, const PlaceParams &P, int q, int nQ,
                                                 const double *sc, int rootCol, const uint8_t *mn, const int32_t *frameOf,
                                                 int nF, int stateCap, double *stLK, int16_t *stFails, uint32_t *frameBits,
                                                 const PlaceOut &o)
{
    const int words = (nF + 31) >> 5;
    for (int i = 0; i < words; i++) frameBits[(long long)i * nQ + q] = 0u;
    if (o.fromBits) for (int i = 0; i < words; i++) o.fromBits[(long long)i * nQ + q] = 0u;
    int32_t *slN = o.slNode + (long long)q * MAPLE_PLACE_SHORTLIST;
    double *slL = o.slLK + (long long)q * MAPLE_PLACE_SHORTLIST;
    int nSl = 0, status = 0, minorNode = -1, missed = 0, nAppend = 1;
    const ScanRec rr = R[0];
    const int root = rr.node;
    double bestLK = sc[rootCol];
    const double originalLK = bestLK;
    int bestNode = root;
    if (!P.supportOnly && rr.leafCol >= 0 && mn[rr.leafCol] == 1) { status = 1; minorNode = root; nAppend = 0; }
    // (lastLK, fails) a node hands to its children: in registers for the node just visited, in the per-depth slots for
    // the ancestors a skipped or finished clade returns to
    double curLK = bestLK;
    int curFails = 0, prevDepth = 0;
    stLK[q] = bestLK; stFails[q] = 0;                                     // depth 0
    int r = rr.leafCol >= 0 ? nReach : 1;
    ScanRec rec = r < nReach ? R[r] : rr;
    while (r < nReach && status == 0) {                                   // M:7972-8100
        // the traversal is almost always r -> r+1 (a leaf's clade is itself): the next record is issued before it is known
        // to be needed.  (A 4-deep window of records AND their scores was slower, 4.7 vs 3.3 ms per batch: loads return
        // in order, so the one load a visit does wait for queues behind the prefetches.)
        const ScanRec ahead = R[r + 1 < nReach ? r + 1 : r];
        const int d = rec.depth;
        if (d >= stateCap) { status = -6; break; }
        double parentLK;
        int fails;
        if (d == prevDepth + 1) { parentLK = curLK; fails = curFails; }
        else { parentLK = stLK[(long long)(d - 1) * nQ + q]; fails = stFails[(long long)(d - 1) * nQ + q]; }
        const double sci = rec.candCol >= 0 ? sc[rec.candCol] : 0.0;
        const int cmp = rec.leafCol >= 0 ? mn[rec.leafCol] : 0;
        const int t1 = rec.node;
        if (rec.leafCol >= 0) {
            if (cmp == 1) {                                               // M:7986-8003
                if (!P.supportOnly) { status = 1; minorNode = t1; break; }
            } else if (cmp == 2) missed++;
        }
        double lk = parentLK;
        if (rec.candCol >= 0) {
            lk = sci;
            nAppend++;
            bool keep = false;
            if (lk >= bestLK) {                                           // M:8065-8073
                const int f = rec.frame;
                frameBits[(long long)(f >> 5) * nQ + q] |= 1u << (f & 31);
                bestLK = lk; bestNode = t1; fails = 0; keep = true;
            } else if (lk > bestLK - P.thrOpt) keep = true;               // M:8074-8075
            if (keep) {
                if (nSl == MAPLE_PLACE_SHORTLIST) {                       // drop what the final filter (M:8109) would drop anyway
                    int k = 0;
                    for (int i = 0; i < nSl; i++)
                        if (slL[i] >= bestLK - P.thrFilter) { slN[k] = slN[i]; slL[k] = slL[i]; k++; }
                    nSl = k;
                }
                if (nSl == MAPLE_PLACE_SHORTLIST) { status = -6; break; }
                slN[nSl] = t1; slL[nSl] = lk; nSl++;
            }
            if (lk < parentLK - P.thrConsec) fails++;                     // M:8076-8077
        }
        const bool within = lk > bestLK - P.thrLK;
        const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);   // M:8080-8093
        stLK[(long long)d * nQ + q] = lk; stFails[(long long)d * nQ + q] = (int16_t)fails;
        curLK = lk; curFails = fails; prevDepth = d;
        if (go && o.fromBits && (rec.childFrame[0] >= 0 || rec.childFrame[1] >= 0)) {
            // the children are pushed now: a child in a frame of its own gets its list from this frame's list as it is NOW
            const int f = rec.frame;
            if ((frameBits[(long long)(f >> 5) * nQ + q] >> (f & 31)) & 1u)
                for (int k = 0; k < 2; k++)
                    if (rec.childFrame[k] >= 0) o.fromBits[(long long)(rec.childFrame[k] >> 5) * nQ + q] |= 1u << (rec.childFrame[k] & 31);
        }
        if ((go && rec.leafCol < 0) || rec.size == 1) { r += 1; rec = ahead; }   // into the clade, or past a leaf
        else { r += rec.size; if (r < nReach) rec = R[r]; }               // past a pruned clade
    }
    // final filter of the short list (M:8109) and the state of each entry's query list object
    int k = 0;
    for (int i = 0; i < nSl; i++)
        if (slL[i] >= bestLK - P.thrFilter) { slN[k] = slN[i]; slL[k] = slL[i]; k++; }
    nSl = k;
    uint8_t *slS = o.slShort + (long long)q * MAPLE_PLACE_SHORTLIST;
    for (int i = 0; i < nSl; i++) {
        const int f = frameOf[slN[i]];
        slS[i] = (frameBits[(long long)(f >> 5) * nQ + q] >> (f & 31)) & 1u;
    }
    const int fb = frameOf[status == 1 ? minorNode : bestNode];
    o.bestShort[q] = (frameBits[(long long)(fb >> 5) * nQ + q] >> (fb & 31)) & 1u;
    o.status[q] = status; o.minorNode[q] = minorNode; o.bestNode[q] = bestNode;
    o.bestLK[q] = bestLK; o.originalLK[q] = originalLK;
    o.nAppend[q] = nAppend; o.missed[q] = missed; o.nShort[q] = nSl;
}

static __global__ __launch_bounds__(64) void k_place_replay(const ScanRec *__restrict__ R, int nReach, PlaceParams P, int nQ, int nCols,
                                                     int rootCol, const double *__restrict__ score, int nLeaf,
                                                     const uint8_t *__restrict__ minor, const int32_t *__restrict__ frameOf,
                                                     int nF, int stateCap, double *stLK, int16_t *stFails, uint32_t *frameBits,
                                                     PlaceOut o)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nQ) return;
    place_replay_one(R, nReach, P, q, nQ, score + (long long)q * nCols, rootCol, minor + (long long)q * nLeaf, frameOf, nF,
                     stateCap, stLK, stFails, frameBits, o);
}

}  // namespace maple
