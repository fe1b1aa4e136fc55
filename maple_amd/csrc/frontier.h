// maple_amd/csrc/frontier.h -- interface of the frontier tier of the SPR search (frontier.hip) towards maple_hip.hip.
#pragma once
#include "ctx_host.h"

#define FR_STATUS_FALLBACK (-8)        // (internal) the search is handed to the one-lane-per-search kernel

struct FrontierStats {
    int levels = 0, overflow = 0;
    long long itemsUpdating = 0, itemsCached = 0, tempLists = 0, tempWords = 0, tempAux = 0, records = 0;
};

// whole-tree searches whose rows of the dense (query x branch) score table are made next to the tier: their cached-regime
// clades are scanned over those rows inside the tier (k_fr_replay_wide) instead of being handed back with status -5
struct FrontierWide {
    const int32_t *rowOf;              // host, per search of the batch: its row of the table, or -1
    const double *cacheS;              // device: the table, rows of dtree.n scores by preRank
    FiniteRows fin;                    // bitmap form of the rows (or nulls)
    hipEvent_t rowsReady;              // recorded behind the launches that write the rows (or null: same stream)
    int forceWide;                     // 1: every search with a row is a whole-tree search (they were found by running over their
                                       // budget, with any model); 0: those the kernel's own routing hint names (zero-length branch, no error model)
    // trees with MAT local references: the frames' nesting (device arrays by frame index; frame 0 = the root's), for the removed
    // lists of the short-listed branches a clade scan finds in frames below its seed's
    const int32_t *frameParent = nullptr, *frameNode = nullptr;
    int nFrames = 0;
};

__attribute__((visibility("hidden")))
int frontier_search(maple_ctx *c, const SearchParams &P, int m, const int32_t *nodes, int budget, int zeroBudget, SearchOut *hostOut,
                    uint2 *poolW, double *poolA, unsigned long long *poolUsed, long long poolCapW, long long poolCapA,
                    FrontierStats *stats, long long itemsHint, const FrontierWide *wide = nullptr,    // itemsHint: expected expanded items (0 = by the budget)
                    const uint8_t *overHint = nullptr);   // host, per search (or null): leave at once as "over the budget" (status -5) -- the caller knows it would
__attribute__((visibility("hidden"))) void frontier_scratch_free(maple_ctx *c);
__attribute__((visibility("hidden"))) int frontier_export(maple_ctx *c, long long cap, int32_t *q, int32_t *node, long long *n);
__attribute__((visibility("hidden")))
int frontier_level_profile(maple_ctx *c, int cap, long long *itemsU, long long *itemsC, float *msU, float *msC, int *n,
                           long long *waveSmall = nullptr, long long *waveBig = nullptr);
