// maple_amd/csrc/frontier.h -- interface of the frontier tier of the SPR search (frontier.hip) towards maple_hip.hip.
#pragma once
#include "ctx_host.h"

#define FR_STATUS_FALLBACK (-8)        // (internal) the search is handed to the one-lane-per-search kernel

struct FrontierStats {
    int levels = 0, overflow = 0;
    long long itemsUpdating = 0, itemsCached = 0, tempLists = 0, tempWords = 0, tempAux = 0, records = 0;
};

__attribute__((visibility("hidden")))
int frontier_search(maple_ctx *c, const SearchParams &P, int m, const int32_t *nodes, int budget, int zeroBudget, SearchOut *hostOut,
                    uint2 *poolW, double *poolA, unsigned long long *poolUsed, long long poolCapW, long long poolCapA,
                    FrontierStats *stats, long long itemsHint);   // itemsHint: expected expanded items (0 = by the budget)
__attribute__((visibility("hidden"))) void frontier_scratch_free(maple_ctx *c);
__attribute__((visibility("hidden"))) int frontier_export(maple_ctx *c, long long cap, int32_t *q, int32_t *node, long long *n);
