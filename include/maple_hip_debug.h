/* include/maple_hip_debug.h -- measurement aids and test hooks of libmaple_hip_debug.so.
 *
 * NOT part of the operator boundary (include/maple_hip.h): the product library libmaple_hip.so does not export these.
 * __graft_entry__.build() links a second library, libmaple_hip_debug.so, from the same sources with -DMAPLE_DEBUG_ABI: every
 * entry point of maple_hip.h plus the ones below.  The tests of the two innermost device functions (getPartialVec M:4073-4141,
 * simplify M:3697-3717: SURVEY 8a rows a3 / a4, which no batched operator exposes on their own), of the wavefront-wide
 * appendProbNode, the PMC calibration (tools/calib_fetch.py) and the level profile (tools/level_profile.py) load that one
 * (maple_amd.runtime.Device(..., debug=True)). */
#ifndef MAPLE_HIP_DEBUG_H
#define MAPLE_HIP_DEBUG_H
#include "maple_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Profile of the last frontier-tier pass of maple_spr_search_batch (until the next maple_timing_reset): per level of the
 * expansion, the items that still updated genome lists, the items in the cached regime, and the HIP-event time (ms) of the
 * level's two kernels; waveItems*: how many of the list-updating items were walked a wavefront each, by size class.  *n levels
 * (the first min(*n, cap) are written).  A measurement aid; nothing is computed with it. */
int maple_debug_frontier_levels(maple_ctx *ctx, int32_t cap, int64_t *itemsUpdating, int64_t *itemsCached, float *msUpdating,
                                float *msCached, int32_t *n, int64_t *waveItemsSmall /* or NULL */, int64_t *waveItemsBig /* or NULL */);

/* appendProbNode for n (parent list, child list) pairs with ONE WAVEFRONT per pair (maple_amd/csrc/wave_dev.h: the walk cut
 * along its merge path, the factors of all steps at once, the running product in walk order): the same results as
 * maple_append_batch bit for bit; *ms (optional) = kernel time.  A test and timing aid. */
int maple_debug_wave_append_batch(maple_ctx *ctx, int32_t n, const int32_t *parentList, const int32_t *childList,
                                  const uint8_t *isTipC, const double *bLen, double *outLK, float *ms);
/* Debugging aid: record the visit sequence of query index `query` of the next maple_spr_search_batch
 * (per visited item: t1, direction, needsUpdating, failedPasses | lastLK, midProb); -1 switches it off. */
int maple_debug_trace_query(maple_ctx *ctx, int32_t query);
/* PMC calibration: `repeats` launches of a kernel that reads `bytes` bytes with this library's access pattern
 * (one lane = one contiguous 512-byte list, dependent 8-byte loads); returns the total time. */
int maple_debug_calib_walk(maple_ctx *ctx, uint64_t bytes, int32_t repeats, float *ms);
/* ... and of WRITE_SIZE: mode 1 = `bytes` written as a coalesced stream, mode 2 = one 8-byte store into every 64-byte line
 * of a `bytes`-long buffer (the way the score matrix is written: bytes / 8 useful bytes). */
int maple_debug_calib_write(maple_ctx *ctx, uint64_t bytes, int32_t mode, int32_t repeats, float *ms);
int maple_debug_trace_read(maple_ctx *ctx, int32_t *n, int32_t *items4 /*[4*4096]*/, double *vals2 /*[2*4096]*/);
/* Parity hooks for the two innermost device functions (one lane per call):
 * getPartialVec(i12, totLen, mutMatrix, errorRate, vect, upNode, flag), M:4073-4141, with the call's own 4x4 matrix
 * (M16[16*i..], row-major) and vect4[4*i..] (read when i12 == 6); whether `flag` matters follows the model's usingErrorRate
 * (M:4109).  simplify(vec, refA), M:3697-3717 -> 0-3 nucleotide, 4 = R, 6 = keep the vector, -1 = the reference raises. */
int maple_debug_gpv_batch(maple_ctx *ctx, int32_t n, const int32_t *i12, const double *totLen, const double *M16,
                          const double *errorRate, const double *vect4, const uint8_t *upNode, const uint8_t *flag,
                          double *out4);
int maple_debug_simplify_batch(maple_ctx *ctx, int32_t n, const double *vec4, const int32_t *refA, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
