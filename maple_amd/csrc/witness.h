// maple_amd/csrc/witness.h -- interface of the witness filter of the whole-tree SPR searches (witness.hip) towards maple_hip.hip.
#pragma once
#include "ctx_host.h"

__attribute__((visibility("hidden")))
int witness_score(maple_ctx *c, hipStream_t st, int nQ, const int32_t *qList, const uint8_t *qTip, const double *qBLen, int nC,
                  const int32_t *cand, const int32_t *outCol, double *out, long long ldOut, unsigned long long *finMask, int nWords,
                  double meanCandBytes, double queryBytes, long long *pairsOut);
__attribute__((visibility("hidden"))) void witness_scratch_free(maple_ctx *c);
