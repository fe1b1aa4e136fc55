/*
 * include/maple_hip.h -- C ABI of libmaple_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the sample-placement / SPR candidate-scoring path of
 * MAPLE v0.7.5.4 (reference: MAPLEv0.7.5.4.py, cited as M:<line>).  The
 * reference has no FFI; its de-facto operator boundary is the set of
 * top-level pure functions the search loops call.  Each
 * entry point below is the batched form of one of them and names it.
 *
 * Conventions: every function returns 0 on success and a negative code on
 * error (text via maple_last_error); nothing throws or exits.  All `const T*`
 * arguments are HOST buffers owned by the caller unless the name ends in
 * `_dev` (then they are device pointers on the context's GPU and the call is
 * asynchronous on `stream`).  A context is bound to one GPU and is not
 * thread-safe.
 *
 * Genome lists (M:378-390) live in device memory as a CSR arena:
 *   word  = { int32 pos ; uint32 meta }            8 bytes per entry
 *   meta  = type[0:2] | ref[3:4] | hasD0[5] | hasD1[6] | flag[7] | auxoff[8:31]
 *   aux   = per list, f64 stream: for each entry, in order,
 *           [d0 if hasD0][d1 if hasD1][vec0..3 if type==6]; auxoff = index of
 *           the entry's first double inside the list's aux block.
 * `pos` is the 1-based LAST genome position the entry covers (for the
 * single-site types 0-3 and 6 that is the site itself); `ref` is the local
 * reference nucleotide of single-site entries; `flag` is the error-model flag
 * that exists only when the model uses error rates.  The Python tuple length
 * of the reference is recovered from hasD0/hasD1 (+1 when the error model is
 * on and a tail exists).
 */
#ifndef MAPLE_HIP_H
#define MAPLE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): maple_tuning begins with its own size, so that a caller compiled against an older header (a shorter struct) is
 * read for exactly what it passed; the measurement aids of versions <= 3 (maple_debug_*) live in maple_hip_debug.h /
 * libmaple_hip_debug.so.  A binding checks maple_abi_version() == MAPLE_ABI_VERSION after loading the library. */
#define MAPLE_ABI_VERSION 4

enum {
    MAPLE_OK = 0,
    MAPLE_ERR_ARG = -1,      /* bad argument                                   */
    MAPLE_ERR_HIP = -2,      /* HIP runtime error (message has the call)       */
    MAPLE_ERR_NOMEM = -3,    /* arena exhausted                                */
    MAPLE_ERR_STATE = -4,    /* model / lists not set                          */
    MAPLE_ERR_FATAL = -5     /* a state the reference treats as raise Exception("exit") */
};

#define MAPLE_META_TYPE(m)   ((m) & 7u)
#define MAPLE_META_REF(m)    (((m) >> 3) & 3u)
#define MAPLE_META_HASD0     (1u << 5)
#define MAPLE_META_HASD1     (1u << 6)
#define MAPLE_META_FLAG      (1u << 7)
#define MAPLE_META_AUXOFF(m) ((m) >> 8)

typedef struct maple_ctx maple_ctx;

/* thresholds and derived constants of M:51-62, M:3606-3624 */
typedef struct {
    double thresholdProb;              /* --thresholdProb, M:51                 */
    double minBLenSensitivity;         /* already multiplied by 1/lRef, M:3619  */
    double thresholdDiffForUpdate;     /* M:61                                  */
    double thresholdFoldChangeUpdate;  /* M:62                                  */
    double defaultBLen;                /* M:77 (evaluatePlacement fallback)     */
} maple_params;

/* ---- life cycle ----------------------------------------------------------- */
int maple_abi_version(void);
/* refIdx = refIndeces (M:3681-3686), rootFreqs (M:3677-3680 / JC M:3687-3689).
 * arena_bytes = device memory reserved for genome lists (0 = default 1 GiB). */
int maple_create(maple_ctx **out, int device, int32_t lRef, const uint8_t *refIdx,
                 const double *rootFreqs4, const maple_params *params, uint64_t arena_bytes);
int maple_destroy(maple_ctx *ctx);
/* How the library schedules its work -- never WHAT it computes (every setting gives bit-identical results; the tests flip
 * them to compare kernels).  Zero-initialise, set what is wanted, the rest keeps the library's choice. */
typedef struct {
    uint32_t structSize;        /* = sizeof(maple_tuning) of the header the CALLER was compiled with: fields beyond it keep the
                                   library's choice (0), fields the library does not know are ignored; 0 is an error */
    int32_t wavePerItemMax;     /* the explicit-pair operators (append / merge / blen / differ / shorten), maple_update_partials'
                                   levels and evaluatePlacement batches of at most this many items run one WAVEFRONT per item
                                   (lowest latency), larger ones one lane per item; 0 = the library's own thresholds, -1 = never */
    int32_t placementChunkMax;  /* queries per chunk of maple_placement_search_batch (0 = by free memory) */
    int32_t noCladeScan;        /* 1: whole-tree SPR searches are replayed one branch at a time instead of by the
                                   wavefront-wide clade scan */
    int32_t verbose;            /* 1: progress lines on stderr (also switched on by the environment variable MAPLE_DEBUG) */
    int32_t wideOutsideFrontier; /* 1: whole-tree SPR searches leave the frontier tier at once and run one wavefront per search
                                   from their first step (k_spr_search), instead of sharing its batched updating steps */
    int32_t denseWideScoring;   /* 1: the whole-tree SPR searches known beforehand are scored against EVERY branch by the dense kernel
                                   instead of only against the branches their witness filter cannot rule out (witness.hip) */
    int32_t waveAllBelow;       /* frontier tier of the SPR search: a level of the expansion with at most this many items that still
                                   update genome lists is walked one WAVEFRONT per item (0 = the library's choice, -1 = never) */
    int32_t noOverHint;         /* 1: with an error model, a node whose SPR search ran over the whole-tree budget the last time is
                                   NOT sent to the dense tier at once the next time (the library remembers that per node between
                                   calls on one tree: maple_spr_search_batch) */
    int32_t noAheadExpansion;   /* 1: maple_placement_ahead scores EVERY branch for the samples it is given, instead of only the
                                   branches an expansion under permissive rules reaches (the tests compare the two); 2: the
                                   expansion stops after six levels (tests: every search then takes the full-row path) */
    int32_t noAheadSpeculation; /* 1: the traversal of the next announced sample (maple_placement_ahead) is not run ahead of its search by a
                                   host thread of the library while the caller places the sample before it */
} maple_tuning;
int maple_set_tuning(maple_ctx *ctx, const maple_tuning *t);
const char *maple_last_error(maple_ctx *ctx);

/* Model tables: replaces the *Passed / *Global keyword arguments of the
 * reference's functions (M:4446, 5040, 6505, 6817) and derives
 * cumulativeRate / cumulativeErrorRate / totError on the way exactly as
 * updateMutMatrices (M:6350-6370) and updateErrorRates (M:6373-6390) do.
 * siteRates != NULL <=> useRateVariation; errorRates != NULL <=> errorRateSiteSpecific. */
int maple_set_model(maple_ctx *ctx, const double *Q16, const double *siteRates, int usingErrorRate,
                    double errorRateGlobal, const double *errorRates);
/* read back derived tables (for parity checks): any pointer may be NULL */
int maple_get_model(maple_ctx *ctx, double *cumulativeRate /*[lRef+1]*/, double *cumulativeErrorRate /*[lRef+1]*/,
                    double *totError);

/* ---- genome-list arena ------------------------------------------------------ */
/* New contents for existing lists (re-upload of dirty lists): ids[i] keeps its number -- tree columns,
 * candidate sets and an uploaded tree that refer to it stay valid -- and names the new words from now on.  A list that fits
 * in the room of the old one is overwritten in place, otherwise it gets fresh room at the end of the arena.  Packed CSR
 * input as for maple_lists_upload. */
int maple_lists_update(maple_ctx *ctx, int32_t n, const int32_t *ids, const int64_t *ent_off, const int32_t *pos,
                       const uint32_t *meta, const int64_t *aux_off, const double *aux);
/* Upload n_lists packed lists; ids first_id .. first_id+n_lists-1 are assigned.
 * ent_off / aux_off are CSR offsets with n_lists+1 elements. */
int maple_lists_upload(maple_ctx *ctx, int32_t n_lists, const int64_t *ent_off, const int32_t *pos,
                       const uint32_t *meta, const int64_t *aux_off, const double *aux, int32_t *first_id);
int maple_lists_sizes(maple_ctx *ctx, int32_t n, const int32_t *ids, int32_t *n_ent, int32_t *n_aux);
/* Download into caller buffers sized from maple_lists_sizes (CSR, same layout as upload). */
int maple_lists_download(maple_ctx *ctx, int32_t n, const int32_t *ids, const int64_t *ent_off, int32_t *pos,
                         uint32_t *meta, const int64_t *aux_off, double *aux);
/* Stack discipline for temporaries: everything created after `mark` is dropped -- genome lists AND MAT mutation lists
 * (maple_mutations_upload); the mark is opaque. */
int maple_arena_mark(maple_ctx *ctx, int64_t *mark);
int maple_arena_release(maple_ctx *ctx, int64_t mark);
int maple_arena_stats(maple_ctx *ctx, int64_t *n_lists, int64_t *n_entries, int64_t *n_aux, int64_t *cap_entries);
/* Compaction: keep only the lists live[0 .. nLive) (no duplicates; -1 entries stay -1), copied to the bottom of the arena in
 * that order and renumbered 0, 1, 2, ...: newIds[i] = the new id of live[i].  Every other list id, every arena mark, resident
 * candidate sets and the uploaded tree are gone (upload the tree again with the new ids).  maple_update_partials and the
 * single-sample placement loop bump-allocate every replaced list; this is how a long run gives the room back. */
int maple_arena_compact(maple_ctx *ctx, int64_t nLive, const int32_t *live, int32_t *newIds);

/* MAT branch mutation lists (tree.mutations[node], M:336): triples (pos, from, to). CSR upload. */
int maple_mutations_upload(maple_ctx *ctx, int32_t n_lists, const int64_t *off, const int32_t *mut3,
                           int32_t *first_id);

/* Where the reference wraps a call in try/except (findBestRoot, M:7793-7828; the SPR worker, M:9703), an item that
 * hits a state the reference raises on must not fail the whole batch: with tolerate != 0 the list-producing batch
 * operators return list id -2 for such an item instead of MAPLE_ERR_FATAL.  Off by default. */
int maple_set_fatal_policy(maple_ctx *ctx, int tolerate);

/* ---- batched operators (host index arrays in, host results out) -------------- */
/* appendProbNode(probVectP, probVectC, isTipC, bLen), M:6505-6785 -> log-LK (may be -inf) */
int maple_append_batch(maple_ctx *ctx, int32_t n, const int32_t *parentList, const int32_t *childList,
                       const uint8_t *isTipC, const double *bLen, double *outLK);
/* mergeVectors(pv1,bLen1,fromTip1,pv2,bLen2,fromTip2,isUpDown=...), M:4446-4859.
 * outList[i] = new list id, or -1 where the reference returns None.
 * outLK may be NULL; when given, the returnLK=True value (M:4856) is also produced
 * (numMinor1/2 may be NULL = 0). */
int maple_merge_batch(maple_ctx *ctx, int32_t n, const int32_t *list1, const double *bLen1, const uint8_t *fromTip1,
                      const int32_t *list2, const double *bLen2, const uint8_t *fromTip2, const uint8_t *isUpDown,
                      const int32_t *numMinor1, const int32_t *numMinor2, int32_t *outList, double *outLK);
/* estimateBranchLengthWithDerivative(P, C, fromTipC), M:5040-5358; isFalse[i]=1 where it returns False */
int maple_blen_batch(maple_ctx *ctx, int32_t n, const int32_t *parentList, const int32_t *childList,
                     const uint8_t *fromTipC, double *t, uint8_t *isFalse);
/* areVectorsDifferent(pv1, pv2), M:5419-5472; list2 == -1 means None -> different */
int maple_differ_batch(maple_ctx *ctx, int32_t n, const int32_t *list1, const int32_t *list2, uint8_t *out);
/* Resident candidate sets for the placement loop (M:7972-8100): `lists[k]` is the parent-side list of candidate k
 * (probVectTotUp of a branch, or probVect of a leaf for the minor-sequence test) and `frameIdx[k]` the index of the MAT
 * reference frame it lives in.  A query is then scored against the whole set in ONE launch, given its genome list in
 * every frame (frameLists[nFrames], e.g. produced level by level with maple_pass_branch_batch). */
int maple_candset_create(maple_ctx *ctx, int32_t n, const int32_t *lists, const int32_t *frameIdx, int32_t nFrames,
                         int32_t *setId);
int maple_candset_destroy(maple_ctx *ctx, int32_t setId);   /* frees the set's device arrays (the id is not reused) */
int maple_append_candset(maple_ctx *ctx, int32_t setId, const int32_t *frameLists, int isTipC, double bLen, double *outLK);
int maple_minor_candset(maple_ctx *ctx, int32_t setId, const int32_t *frameLists, int onlyFindIdentical, uint8_t *out);
/* findProbRoot(probVect) for lists already expressed in the root frame, M:4865-4912 */
int maple_root_prob_batch(maple_ctx *ctx, int32_t n, const int32_t *list, double *outLK);
/* isMinorSequence(probVect1, probVect2, onlyFindIdentical), M:5919-6004 -> 0 / 1 / 2 */
int maple_minor_batch(maple_ctx *ctx, int32_t n, const int32_t *list1, const int32_t *list2, int onlyFindIdentical,
                      uint8_t *out);
/* passGenomeListThroughBranch(probVect, mutations, dirIsUp), M:3749-3877 */
int maple_pass_branch_batch(maple_ctx *ctx, int32_t n, const int32_t *list, const int32_t *mutList,
                            const uint8_t *dirIsUp, int32_t *outList);
/* shorten(vec), M:3721-3745 (produces a new list; the input is left untouched) */
int maple_shorten_batch(maple_ctx *ctx, int32_t n, const int32_t *list, int32_t *outList);
/* rootVector(probVect, bLen, isFromTip, tree, node), M:4916-4996.  The walk to the root
 * (M:4930-4940, 4988-4993) is given as a CSR of mutation-list ids per call, node first, root last. */
int maple_root_vector_batch(maple_ctx *ctx, int32_t n, const int32_t *list, const double *bLen,
                            const uint8_t *isFromTip, const int64_t *pathOff, const int32_t *pathMutLists,
                            int32_t *outList);
/* updatePartials(tree, nodeList), M:5479-5815, for ANY number of simultaneous local changes (maple_amd/csrc/update_host.h).
 * The tree is the caller's own: n nodes as plain columns -- up / child0 / child1 (-1 = none), isTip (leaf without minor
 * sequences), mutList (mutation-list id of the branch above the node, -1 = none), depth (branches from the root), dist,
 * and the four list-id columns lower / upRight / upLeft / totUp (-1 = None).  changed[] = nodes whose lower list (already
 * replaced in lower[]) and/or branch length (already replaced in dist[]) changed.  The invalidated lists are repaired level
 * by level -- lower lists upwards while areVectorsDifferent(new, old) (M:5793), then probVectTotUp / probVectUpRight /
 * probVectUpLeft downwards while areVectorsDifferent(old, new) (M:5645-5658) -- and lower / upRight / upLeft / totUp
 * (and dist[], where a None merge between two zero-length branches makes the reference re-estimate a length,
 * M:5385-5414) are updated IN PLACE with the ids of the new lists; *nReplaced = lists replaced.  New lists are bump-
 * allocated in the arena; the old ones stay where they are (maple_arena_mark / _release around a trial change). */
int maple_update_partials(maple_ctx *ctx, int32_t n, int32_t root, const int32_t *up, const int32_t *child0,
                          const int32_t *child1, const uint8_t *isTip, const int32_t *mutList, const int32_t *depth,
                          double *dist, int32_t *lower, int32_t *upRight, int32_t *upLeft, int32_t *totUp, int32_t nChanged,
                          const int32_t *changed, int32_t *nReplaced);
/* The nodes whose lists (or branch length) the last maple_update_partials replaced, each once, ascending (*n of them; an
 * error if they do not fit in cap): with the nodes of the tree edit itself, what maple_tree_patch has to be told. */
int maple_update_partials_touched(maple_ctx *ctx, int32_t cap, int32_t *nodes, int32_t *n);

/* reCalculateAllGenomeLists (M:6013-6347) inside the library: from the tips' lower lists (lower[tip] on entry) every internal
 * node's probVect, deepest level first (M:6031-6200), then probVectUpRight / probVectUpLeft / probVectTotUp of every node from the
 * root down (M:6226-6345) -- one fused launch (mergeVectors -> shorten) per level on the caller's tree columns (as for
 * maple_update_partials; mut may be NULL: no MAT local references), the four id columns rewritten for every node reachable from
 * root.  The new lists are allocated after the caller's arena mark like any other.  bumpLen = 0: a None between zero-length
 * branches is MAPLE_ERR_FATAL (the reference calls updateBLen there, M:6087 / 6279).  bumpLen > 0 (a synthetic tree being built):
 * in the lower pass the two child branches are lengthened to bumpLen (dist is updated) and merged again; a None that remains, or
 * one in the upper pass, ends the call with *nBad > 0 and the nodes in badNodes (capBad >= 4) for the caller to lengthen; *nBad counts
 * every such node, badNodes holds the first capBad of them (capBad = n never cuts the list short).  `up` must agree with the
 * children columns (checked). */
int maple_tree_rebuild_lists(maple_ctx *ctx, int32_t n, int32_t root, const int32_t *up, const int32_t *child0, const int32_t *child1,
                             const uint8_t *isTip, const int32_t *mutList /* or NULL */, double *dist, int32_t *lower, int32_t *upRight,
                             int32_t *upLeft, int32_t *totUp, double bumpLen, int32_t capBad, int32_t *badNodes, int32_t *nBad);
/* evaluatePlacement(midTot, downVect, upVect, distance, removedPartials, isRemovedTip, ..., fromTip1), M:6790-6806
 * out4[i*4..] = appendingCost, bestBottomLength, bestTopLength, bestAppendingLength (False -> 0.0) */
int maple_evaluate_placement_batch(maple_ctx *ctx, int32_t n, const int32_t *midTot, const int32_t *downVect,
                                   const int32_t *upVect, const double *distance, const int32_t *removedPartials,
                                   const uint8_t *isRemovedTip, const uint8_t *fromTip1, double *out4);

/* ---- tree mirror and the device-resident SPR search ------------------------------- */
/* Topology of the tree whose genome lists are in the arena (struct-of-arrays Tree, M:331-376):
 * up / child0 / child1 use -1 for None; isTip[n] = leaf with no minor sequences (the `isTip` tests of
 * M:6986, 7129, 9645); lower/upRight/upLeft/totUp are list ids (probVect, probVectUpRight,
 * probVectUpLeft, probVectTotUp; -1 = None); mutList[n] = mutation-list id of tree.mutations[n], -1 if empty. */
/* (validated: index ranges, both-or-neither children, children pointing back to their parent, no cycle below the root --
 * MAPLE_ERR_ARG otherwise; node slots not reachable from the root are ignored) */
int maple_tree_upload(maple_ctx *ctx, int32_t n, int32_t root, const int32_t *up, const int32_t *child0,
                      const int32_t *child1, const double *dist, const uint8_t *isTip, const int32_t *lower,
                      const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp, const int32_t *mutList);

typedef struct {
    int32_t strictTopologyStopRules;            /* M:57  */
    int32_t allowedFailsTopology;               /* M:54  */
    double thresholdLogLKtopology;              /* M:53, already multiplied by log(lRef) (M:3612) */
    double thresholdTopologyPlacement;          /* M:56  */
    double thresholdLogLKoptimizationTopology;  /* M:66, x log(lRef) (M:3609), data-adaptive M:11770 */
    double thresholdLogLKconsecutivePlacement;  /* M:63  */
    double effectivelyNon0BLen;                 /* M:3614 */
    int32_t wideSearchBudget;                   /* searches scoring more branches than this are batch-scored first
                                                   (0 = default 256, < 0 = never); results do not depend on it */
    int32_t searchTier;                         /* 0 = automatic (the frontier tier where it applies: trees without MAT local
                                                   references); 1 = one lane per search (k_spr_search) for every search;
                                                   results do not depend on it */
} maple_search_params;

/* A local change of the uploaded tree -- what placeSampleOnTree (M:8300-8722) and the updatePartials after it leave behind:
 * a few nodes with new relatives, branch lengths or list ids, one or two new nodes.  nodes[i] gets the record
 * (up, child0, child1, dist, isTip, lower, upRight, upLeft, totUp)[i]; ids >= the old node count are new nodes (all of them
 * listed; nTotal = the new count).  The root and the mutation lists (MAT reference nodes) do not change this way: re-upload
 * the tree for that.  The single-query placement search (maple_placement_search_batch with <= 4 queries: the serial
 * placement phase, M:11744-11752) then runs on the patched tree at once; the tables of the batched placement search and of
 * the SPR search are rebuilt from the library's own copy of the tree before their next use. */
int maple_tree_patch(maple_ctx *ctx, int32_t nTotal, int32_t nTouched, const int32_t *nodes, const int32_t *up,
                     const int32_t *child0, const int32_t *child1, const double *dist, const uint8_t *isTip,
                     const int32_t *lower, const int32_t *upRight, const int32_t *upLeft, const int32_t *totUp);
/* The worker body of startTopologyUpdatesParallel (M:9615-9711) for n pruned nodes, each running
 * findBestParentTopology (M:6817-7724) entirely on the GPU (one lane per query).  Per query:
 *   bestNode/bestScore/blen3 = findBestParentTopology's (bestNode, bestScore, bestBranchLengths);
 *   placement (-1 = None) / improvement = the proposed move after the accept rule and vetoes (M:9681-9700);
 *   currentLK = bestCurrentLK (M:9646); nAppend = appendProbNode evaluations issued by the search
 *   (the "candidate placements" of the metric); status: 0 searched, 1 root, 2 not searched (M:9674),
 *   -1 the reference would have raised inside the search (its worker swallows it, M:9703),
 *   -3 per-lane workspace exhausted (retry with a larger ws_entries_per_lane).
 * outRprList (may be NULL): new list ids of bestRemovedPartials.  ws_entries_per_lane: 0 = default.
 * State kept between calls (scheduling only, never results): with an error model the context remembers, per node of the uploaded
 * tree, that the node's search ran over the whole-tree budget, and sends it to the dense tier at once the next time -- kept across
 * maple_tree_patch, dropped by maple_tree_upload, switched off by maple_tuning.noOverHint. */
int maple_spr_search_batch(maple_ctx *ctx, int32_t n, const int32_t *nodes, const maple_search_params *params,
                           int32_t ws_entries_per_lane, int32_t *bestNode, double *bestScore, double *blen3,
                           int32_t *placement, double *improvement, double *currentLK, int32_t *nAppend,
                           int32_t *status, int32_t *outRprList);

/* What the searches of the last maple_spr_search_batch may have READ: (index into that call's nodes[], tree node) for every
 * branch a search visited or could have visited under any outcome of its order-dependent rules (the frontier tier's expanded
 * items) -- a superset of the visited branches.  A caller that applies proposed moves one after the other (applySPRMovesParallel,
 * M:9470-9484: every move is re-searched on the tree the earlier ones left) can re-search a BATCH of moves speculatively and
 * keep the result of a later one as long as no node these lists name for it, nor a relative of one, was touched by the moves
 * applied before it.  Only after a call that ran wholly in the frontier tier (wideSearchBudget < 0, a tree without MAT local
 * references); *n pairs, MAPLE_ERR_ARG if they do not fit in cap. */
int maple_spr_search_visited(maple_ctx *ctx, int64_t cap, int32_t *query, int32_t *node, int64_t *n);

typedef struct {
    double oneMutBLen;                          /* M:3606 */
    double effectivelyNon0BLen;                 /* M:3607 */
    double thresholdLogLK;                      /* x log(lRef), M:3613 */
    double thresholdLogLKoptimization;          /* x log(lRef), M:3610 */
    double thresholdLogLKconsecutivePlacement;  /* M:63 */
    int32_t allowedFails;                       /* M:50 */
    int32_t strictStopRules;                    /* M:56 */
    int32_t onlyFindIdentical;                  /* the minor-sequence mode of M:7976 (any error-model flag, HnZ, ...) */
} maple_placement_params;

/* findBestParentForNewSample (M:7912-8292) for nQ query samples against the uploaded (frozen) tree -- the batch shape
 * of --findSamplePlacements / --lineageRefs (M:11190-11220).  qLists = the samples' genome lists in the root's
 * reference frame.  Every query is scored against every branch in one launch (each in that branch's MAT reference
 * frame), the reference's traversal, stop rules and tie-breaks are replayed on the device (one lane per query), and the
 * short lists are refined in one batch (M:8101-8187).  Per query: bestNode, bestScore, blen3 = (top, bottom,
 * appending; False -> 0.0), bestDiffs = id of the list the reference returns as bestDiffs (a new or an input list),
 * nAppend = appendProbNode evaluations the REFERENCE would have issued, status: 0 placed by likelihood,
 * 1 = the query is a minor sequence of bestNode (score 1.0, M:7986-8003), -6 short list or stack overflow. */
/* Optional: derive the per-tree tables of the placement search (candidate columns, traversal order) and the root vector
 * (rootVector(probVect[root]), M:7958, one new arena list) now, OUTSIDE any arena mark of the caller, so that the
 * searches that follow do not recompute them; they are dropped when the tree is uploaded again or the arena is released
 * below them. */
int maple_placement_prepare(maple_ctx *ctx, const maple_placement_params *params);
/* The serial placement loop (M:11692-11752: one sample placed, the tree edited, the next sample placed) scores every sample
 * against every branch of the tree as it is then.  A placement changes a handful of lists; every other branch scores what it
 * scored before.  maple_placement_ahead scores the next nQ samples (qLists, in the order they will be searched) against the
 * CURRENT tree in one launch; while those rows live, maple_tree_patch notes the columns whose list changes and the columns
 * it adds, and a maple_placement_search_batch call for ONE sample that is the next of the announced ones only scores those
 * columns again (for all samples still waiting) and then runs the traversal over its row: the same scores, the same
 * result as without the announcement.  *nTaken = the leading samples the library made rows for (what fits its page-locked
 * tables; 0 on a tree with MAT reference frames, where the call changes nothing); announce the rest when those are done.
 * The rows hold the branches an expansion of all announced samples under permissive rules reaches (every branch the
 * reference's traversal can visit on the tree as it is then; maple_tuning.noAheadExpansion: every branch of the tree).
 * While rows live, the library runs the traversal of the NEXT announced sample in a host thread of its own as soon as a search
 * has finished its own -- on library-owned memory only, joined at the start of maple_tree_patch and of every call that could
 * change what it reads; it is used if the patch in between touched no node it had visited (maple_tuning.noAheadSpeculation
 * switches it off).
 * The rows are dropped by anything that renumbers the columns or changes what a score means (maple_tree_upload,
 * maple_set_model, other parameters, a release of the samples' lists), and by a search of any other sample. */
int maple_placement_ahead(maple_ctx *ctx, int32_t nQ, const int32_t *qLists, const maple_placement_params *params, int32_t *nTaken);
/* What the rows made ahead were used for since the context was created: out5 = searches that took a row, those among them whose
 * row had to be scored in full after all (the traversal asked for a branch the expansion had not reached), items the
 * expansions scored, traversals made ahead of their search that were used, and that were dropped (the placement in between
 * touched a node they had visited). */
int maple_placement_ahead_stats(maple_ctx *ctx, int64_t *out5);
int maple_placement_search_batch(maple_ctx *ctx, int32_t nQ, const int32_t *qLists, const maple_placement_params *params,
                                 int32_t *bestNode, double *bestScore, double *blen3, int32_t *bestDiffs,
                                 int32_t *nAppend, int32_t *status);

/* The same search through the computePlacementSupportOnly=True exit of findBestParentForNewSample (M:7940, 7986,
 * 8101-8290) -- what process_chunk consumes for --lineageRefs / --findSamplePlacements (M:11190-11220): a leaf the query
 * is a minor sequence of does not end the search, every refined branch within max(thresholdLogLKoptimization,
 * thresholdLogLKoptimizationTopology) of the best becomes a possible placement, zero-top-length placements are moved to
 * the top of their polytomy, supports are exp(score) normalised.  Per query g: possiblePlacements =
 * (outNode, outSupport, outBlen3 = top, bottom, appending)[outOff[g] .. outOff[g+1]) with support >= minBranchSupport, in
 * the reference's order (unsorted); bestTotalLh[g] = list id of bestPlacementTotalLh (-1 = its empty list).
 * cap = capacity of the three output arrays (entries). */
int maple_placement_supports_batch(maple_ctx *ctx, int32_t nQ, const int32_t *qLists, const maple_placement_params *params,
                                   double thresholdLogLKoptimizationTopology, double minBranchSupport, int64_t cap,
                                   int64_t *outOff, int32_t *outNode, double *outSupport, double *outBlen3,
                                   int32_t *bestTotalLh, int32_t *status);

/* ---- device-resident forms (inputs already in HBM; asynchronous on `stream`) ---
 * `stream` is the caller's hipStream_t, used verbatim: NULL is the legacy default stream (what
 * torch.cuda.current_stream().cuda_stream is unless the caller switched streams), so the launch is ordered with the
 * caller's other work on that stream.  (The context's own stream, used by the host-buffer entry points, is
 * non-blocking: it is NOT ordered against the default stream; those entry points return only after their work is done.) */
int maple_append_batch_dev(maple_ctx *ctx, int32_t n, const int32_t *parentList_dev, const int32_t *childList_dev,
                           const uint8_t *isTipC_dev, const double *bLen_dev, double *outLK_dev, void *stream);
/* Q queries x C candidates (the placement loop, M:8050, and the cached regime of the SPR search, M:6993-7011):
 * out_dev[q*C + k] = appendProbNode(list cand[k], list qList[q], isTipC, bLen); one lane per pair, no index arrays. */
int maple_append_queries_dev(maple_ctx *ctx, int32_t nQ, const int32_t *qList_dev, int32_t nC, const int32_t *cand_dev,
                             int isTipC, double bLen, double *out_dev, void *stream);
/* The same pairs without the score matrix: per query the best score and the index (into cand_dev) of the candidate that
 * has it -- a wavefront reduction over each tile of 64 candidates, one 16-byte record per (query, tile), then one
 * wavefront per query over its tiles.  Exact ties go to the smallest visitRank_dev[k] (NULL: the smallest k), the
 * reference's "first of equal scores wins" (strict >, M:7083 / 8065). */
int maple_append_queries_argmax_dev(maple_ctx *ctx, int32_t nQ, const int32_t *qList_dev, int32_t nC, const int32_t *cand_dev,
                                    const int32_t *visitRank_dev, int isTipC, double bLen, double *bestScore_dev,
                                    int32_t *bestIdx_dev, void *stream);
/* Multi-GPU arg-max over RCCL (level 2: the candidates of one query sharded over the GPUs; one
 * process per GPU).  maple_comm_unique_id on one rank, broadcast the 128 bytes by any means, maple_comm_init on every
 * rank.  maple_argmax_allreduce_dev, in place and asynchronous on `stream`: score[i] = max over ranks, idx[i] = the
 * smallest idx among the ranks holding that score (idx = depth-first visit index: the earliest visit wins a tie).  Two
 * ncclAllReduce of n 8-byte words (max of order-preserving keys, min of offered indices).  RCCL is dlopen'ed. */
int maple_comm_unique_id(maple_ctx *ctx, uint8_t *id128);
int maple_comm_init(maple_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id128);
int maple_argmax_allreduce_dev(maple_ctx *ctx, int32_t n, double *score_dev, int32_t *idx_dev, void *stream);
/* Every *_dev launch is bracketed by a pair of HIP events recorded on the launch's own stream.
 * maple_timing_read sums the elapsed time of all launches since the last maple_timing_reset. */
int maple_timing_reset(maple_ctx *ctx);
int maple_timing_read(maple_ctx *ctx, int32_t *n_launches, double *total_ms);
int maple_timing_read_each(maple_ctx *ctx, int32_t cap, float *ms, int32_t *n_launches);   /* one value per launch */
/* The same, restricted to one kind of launch since the last reset, with the work it did:
 *   1 = dense scoring inside maple_spr_search_batch (k_append_queries / k_place_score: units = (query, branch) pairs,
 *       alg_bytes = SURVEY 8d bytes: 8 E + 8 A + 8 per candidate branch per query, each query list once per launch),
 *   2 = budgeted lane searches, 3 = searches replayed over cached scores (units = searches),
 *   4 = maple_append_queries_dev, 5 = maple_append_batch_dev, 6 = scoring inside maple_placement_search_batch (units = pairs),
 *   the frontier tier of the SPR search (kind 2 = the tier as a whole; its kernels, one record per launch):
 *   7 = k_fr_updating (items that still update genome lists), 8 = k_fr_cached (units = cached-regime placements scored,
 *   alg_bytes = their SURVEY 8d bytes, counted on the device), 9 = exact replay + refinement + final selection,
 *   10 = k_fr_replay_wide (the whole-tree searches replayed over their rows of the dense score table: units = placements). */
int maple_timing_read_kind(maple_ctx *ctx, int32_t kind, int32_t *n_launches, double *total_ms, double *units,
                           double *alg_bytes);
/* algorithmic bytes (8*E + 8*B + 32*O + 8 per candidate, child list once per query) */
int maple_append_algorithmic_bytes(maple_ctx *ctx, int32_t n, const int32_t *parentList, const int32_t *childList,
                                   int child_once, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif
