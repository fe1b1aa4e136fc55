/*
 * oracle/maple_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the genome-list arithmetic of
 * MAPLE v0.7.5.4 (reference file MAPLEv0.7.5.4.py, cited below as M:<line>).
 * It exists to check the HIP path (maple_amd/csrc) and to serve as the timed
 * CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product never does.
 *
 * Pinned: every function here is checked against call records harvested from
 * the unmodified reference running in the build container
 * (tests/golden/make_golden.py -> tests/golden/calls_*.json.gz,
 *  tests/test_oracle_golden.py).
 *
 * The data model follows the reference literally (array of tuple-like
 * structs that remember the Python tuple length) and is deliberately NOT the
 * packed CSR layout the HIP library uses.
 */
#ifndef MAPLE_ORACLE_H
#define MAPLE_ORACLE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One genome-list entry, M:378-390.  `len` is the Python tuple length. */
typedef struct {
    int type;      /* 0..3 = A,C,G,T ; 4 = R ; 5 = N ; 6 = O              */
    int x;         /* entry[1]: last position of the run (types 4,5) or   */
                   /* the local-reference nucleotide (types 0-3, 6)       */
    int len;       /* tuple length: 2..5                                  */
    int flag;      /* trailing error flag (only when usingErrorRate)      */
    double d0;     /* entry[2] when it is a branch length                 */
    double d1;     /* entry[3] when it is a branch length                 */
    double vec[4]; /* entry[-1] for type 6                                */
} OEntry;

typedef struct {
    int lRef;
    int useRateVariation;
    int usingErrorRate;
    int errorRateSiteSpecific;
    const unsigned char *refIdx;        /* refIndeces, M:3681-3686         */
    const double *siteRates;            /* [lRef] or NULL                  */
    const double *errorRates;           /* [lRef] or NULL                  */
    const double *cumulativeRate;       /* [lRef+1], M:6350-6370           */
    const double *cumulativeErrorRate;  /* [lRef+1] or NULL, M:6373-6390   */
    double Q[16];                       /* mutMatrixGlobal row-major       */
    double rootFreqs[4];
    double errorRate;                   /* errorRateGlobal                 */
    double totError;
    double globalTotRate;               /* -lRef, M:3607                   */
    double minimumCarryOver;            /* DBL_MIN*1e50, M:3623            */
    double thresholdProb;               /* M:51                            */
    double minBLenSensitivity;          /* already scaled by 1/lRef, M:3619*/
    double thresholdDiffForUpdate;      /* M:61                            */
    double thresholdFoldChangeUpdate;   /* M:62                            */
} OModel;

/* Each function returns 0 on success, <0 on a state the reference treats as
 * fatal (raise Exception("exit")), and documents its own extra codes.       */

void  omo_getPartialVec(const OModel *m, int i12, double totLen, const double *mat16,
                        double errorRate, const double *vect, int upNode, int flag, double *out4);
int   omo_simplify(const OModel *m, const double *vec4, int refA, int *state);
int   omo_shorten(const OModel *m, OEntry *vec, int n);                 /* returns new length */
int   omo_passGenomeListThroughBranch(const OModel *m, const OEntry *pv, int n,
                                      const int *mut3, int nMut, int dirIsUp, OEntry *out);  /* returns n_out */
int   omo_appendProbNode(const OModel *m, const OEntry *P, int nP, const OEntry *C, int nC,
                         int isTipC, double bLen, double *outLK);
/* returns n_out >=0, or -1 when the reference returns None, or <-1 on fatal */
int   omo_mergeVectors(const OModel *m, const OEntry *pv1, int n1, double bLen1, int fromTip1,
                       const OEntry *pv2, int n2, double bLen2, int fromTip2,
                       int returnLK, int isUpDown, int numMinor1, int numMinor2,
                       OEntry *out, double *outLK);
/* *isFalse=1 when the reference returns False */
int   omo_estimateBranchLength(const OModel *m, const OEntry *P, int nP, const OEntry *C, int nC,
                               int fromTipC, double *t, int *isFalse, double *scratch /* >= nP+nC doubles */);
int   omo_areVectorsDifferent(const OModel *m, const OEntry *pv1, int n1, const OEntry *pv2, int n2);
/* path mutations: nPath lists (node first, root last), CSR offsets into mut3 */
int   omo_rootVector(const OModel *m, const OEntry *pv, int n, double bLen, int isFromTip,
                     const int *mut3, const int *pathOff, int nPath, OEntry *out, OEntry *tmp, int cap);
int   omo_findProbRoot(const OModel *m, const OEntry *pv, int n, const int *mut3, const int *pathOff, int nPath,
                       const int *cumulativeBases /* [(lRef+1)*4] */, const double *rootFreqsLogErrorCumulative,
                       OEntry *tmpA, OEntry *tmpB, int cap, double *outLK);
/* evaluatePlacement, M:6790-6806: out4 = appendingCost,bestBottomLength,bestTopLength,bestAppendingLength
 * (a branch length the reference returns as False is reported as 0.0) */
int   omo_evaluatePlacement(const OModel *m, const OEntry *midTot, int nMid, const OEntry *down, int nDown,
                            const OEntry *up, int nUp, double distance, const OEntry *rem, int nRem,
                            int isRemovedTip, int fromTip1, double defaultBLen, double *out4,
                            OEntry *tmp /* 3*cap */, int cap, double *scratch);

/* ---- SPR search (oracle/maple_oracle_search.c) ---------------------------------------------------------- */
typedef struct {
    int n, root;
    const int *up, *c0, *c1;             /* -1 = None                                          */
    const double *dist;
    const int *nMinor;                   /* len(minorSequences[node])                           */
    const OEntry *ent;                   /* all genome lists of the tree, concatenated          */
    const long long *start[4];           /* per kind (0 probVect, 1 probVectUpRight, 2 probVectUpLeft, 3 probVectTotUp) */
    const int *len[4];                   /* 0 = None                                            */
    const int *mut3;                     /* tree.mutations[], triples, CSR by node              */
    const long long *mutOff;             /* [n+1]                                               */
} OTree;

typedef struct {
    int strict, allowedFails;
    double thrLKtopology, thrPlacement, thrOptTopo, thrConsec, effNon0, defaultBLen;
} OSearchParams;

typedef struct {
    int bestNode, placement, status, nAppend;
    double bestScore, improvement, currentLK;
    double blen[3];
    OEntry *rpr;                         /* optional caller buffer for bestRemovedPartials      */
    int rprCap, rprN;
} OSearchResult;

/* findBestParentTopology(tree, node, child, bestLKdiff, removedBLen), M:6817-7724; 0 ok, -1 the reference raises,
 * -3 arena too small */
int omo_findBestParentTopology(const OModel *m, const OTree *t, const OSearchParams *p, int node, int child,
                               double bestLKdiff, double removedBLen, OSearchResult *res, void *arenaMem, size_t arenaBytes);
/* worker body of startTopologyUpdatesParallel, M:9615-9711, for the pruned nodes `nodes` */
int omo_sprWorker(const OModel *m, const OTree *t, const OSearchParams *p, int n, const int *nodes, OSearchResult *out,
                  void *arenaMem, size_t arenaBytes);

/* ... dealt to OpenMP threads (every thread owns arenaBytesPerThread bytes of arenaMem); same results */
int omo_sprWorker_mt(const OModel *m, const OTree *t, const OSearchParams *p, int n, const int *nodes, OSearchResult *out,
                     void *arenaMem, size_t arenaBytesPerThread, int threads);

/* batch driver (lists concatenated, off[] = CSR offsets by list index) used to time the CPU baseline */
int   omo_appendProbNode_batch(const OModel *m, const OEntry *all, const long long *off, int n, const int *pl,
                               const int *cl, const unsigned char *tip, const double *bl, double *out);
int omo_appendProbNode_batch_mt(const OModel *m, const OEntry *all, const long long *off, int n, const int *pl,
                                const int *cl, const unsigned char *tip, const double *bl, double *out, int threads);

/* packed lists of the HIP library (include/maple_hip.h) -> OEntry tuples, for whole trees (a conversion, no arithmetic) */
int omo_entries_from_packed(long long nLists, const long long *entOff, const int *pos, const unsigned *meta,
                            const long long *auxOff, const double *aux, int usingErrorRate, OEntry *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
