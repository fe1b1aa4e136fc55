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
};

// node columns: candIdx[v] = column of v in the score matrix or -1 (M:8049: dist > effectivelyNon0BLen, up != None);
// leafIdx[v] = column of v in the minor-sequence matrix or -1; frameOf[v] = index of v's MAT reference frame.
// The traversal of ONE query.  Lane q of nQ keeps element i of its stack / frame bits at [i * nQ + q] (coalesced when a
// wavefront runs 64 queries; the host calls it with nQ = 1, q = 0 on plain arrays for very small batches, where one
// lane's ~1 us per visit would dominate the call).
__host__ __device__ inline void place_replay_one(const NodeRec *nd, int root, const PlaceParams &P, int q, int nQ,
                                                 const double *sc, int rootCol, const int32_t *candIdx, const uint8_t *mn,
                                                 const int32_t *leafIdx, const int32_t *frameOf, int nF, int stackCap,
                                                 int32_t *stNode, double *stLK, int16_t *stFails, uint32_t *frameBits,
                                                 const PlaceOut &o)
{
    const int words = (nF + 31) >> 5;
    for (int i = 0; i < words; i++) frameBits[(long long)i * nQ + q] = 0u;
    int32_t *slN = o.slNode + (long long)q * MAPLE_PLACE_SHORTLIST;
    double *slL = o.slLK + (long long)q * MAPLE_PLACE_SHORTLIST;
    int nSl = 0, status = 0, minorNode = -1, missed = 0, nAppend = 1;
    const NodeRec rr = nd[root];
    double bestLK = sc[rootCol];
    const double originalLK = bestLK;
    int bestNode = root;
    int sp = 0;
    // the item pushed last is popped next: it stays in registers (top*), so a visit waits for the node record only
    int topNode = -1, topFails = 0;
    double topLK = 0.0;
    bool haveTop = false;
#define MAPLE_PLACE_PUSH(NODE, LK, FAILS)                                                                              \
    do {                                                                                                               \
        if (haveTop) {                                                                                                 \
            stNode[(long long)sp * nQ + q] = topNode; stLK[(long long)sp * nQ + q] = topLK;                            \
            stFails[(long long)sp * nQ + q] = (int16_t)topFails; sp++;                                                 \
        }                                                                                                              \
        topNode = (NODE); topLK = (LK); topFails = (FAILS); haveTop = true;                                            \
    } while (0)
    if (rr.c0 < 0) {
        if (leafIdx[root] >= 0 && mn[leafIdx[root]] == 1) { status = 1; minorNode = root; nAppend = 0; }
    } else {
        MAPLE_PLACE_PUSH(rr.c0, bestLK, 0);
        MAPLE_PLACE_PUSH(rr.c1, bestLK, 0);
    }
    while ((haveTop || sp > 0) && status == 0) {                          // M:7972-8100
        int t1, fails;
        double parentLK;
        if (haveTop) { t1 = topNode; parentLK = topLK; fails = topFails; haveTop = false; }
        else {
            sp--;
            t1 = stNode[(long long)sp * nQ + q];
            parentLK = stLK[(long long)sp * nQ + q];
            fails = stFails[(long long)sp * nQ + q];
        }
        // three independent loads, then two: the visit's dependent chain is two memory latencies deep
        const int ci = candIdx[t1];
        const int li = leafIdx[t1];
        const NodeRec r = nd[t1];
        const double sci = ci >= 0 ? sc[ci] : 0.0;
        const int cmp = li >= 0 ? mn[li] : 0;
        if (r.c0 < 0) {
            if (cmp == 1) { status = 1; minorNode = t1; break; }           // M:7986-8003
            if (cmp == 2) missed++;
        }
        double lk = parentLK;
        if (ci >= 0) {
            lk = sci;
            nAppend++;
            bool keep = false;
            if (lk >= bestLK) {                                           // M:8065-8073
                const int f = r.frameOf;
                frameBits[(long long)(f >> 5) * nQ + q] |= 1u << (f & 31);
                bestLK = lk; bestNode = t1; fails = 0; keep = true;
            } else if (lk > bestLK - P.thrOpt) keep = true;               // M:8074-8075
            if (keep) {
                if (nSl == MAPLE_PLACE_SHORTLIST) {                       // drop what the final filter (M:8109) would drop anyway
                    int k = 0;
                    for (int i = 0; i < nSl; i++)
                        if (slL[i] >= bestLK - P.thrOpt) { slN[k] = slN[i]; slL[k] = slL[i]; k++; }
                    nSl = k;
                }
                if (nSl == MAPLE_PLACE_SHORTLIST) { status = -6; break; }
                slN[nSl] = t1; slL[nSl] = lk; nSl++;
            }
            if (lk < parentLK - P.thrConsec) fails++;                     // M:8076-8077
        }
        const bool within = lk > bestLK - P.thrLK;
        const bool go = P.strict ? (fails <= P.allowedFails && within) : (fails <= P.allowedFails || within);   // M:8080-8093
        if (go && r.c0 >= 0) {
            if (sp + 3 > stackCap) { status = -6; break; }
            MAPLE_PLACE_PUSH(r.c0, lk, fails);
            MAPLE_PLACE_PUSH(r.c1, lk, fails);
        }
    }
#undef MAPLE_PLACE_PUSH
    // final filter of the short list (M:8109) and the state of each entry's query list object
    int k = 0;
    for (int i = 0; i < nSl; i++)
        if (slL[i] >= bestLK - P.thrOpt) { slN[k] = slN[i]; slL[k] = slL[i]; k++; }
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

__global__ __launch_bounds__(64) void k_place_replay(DevTree T, PlaceParams P, int nQ, int nCols, int rootCol,
                                                     const double *__restrict__ score, const int32_t *__restrict__ candIdx,
                                                     int nLeaf, const uint8_t *__restrict__ minor,
                                                     const int32_t *__restrict__ leafIdx, const int32_t *__restrict__ frameOf,
                                                     int nF, int stackCap, int32_t *stNode, double *stLK, int16_t *stFails,
                                                     uint32_t *frameBits, PlaceOut o)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nQ) return;
    place_replay_one(T.nd, T.root, P, q, nQ, score + (long long)q * nCols, rootCol, candIdx, minor + (long long)q * nLeaf,
                     leafIdx, frameOf, nF, stackCap, stNode, stLK, stFails, frameBits, o);
}

}  // namespace maple
