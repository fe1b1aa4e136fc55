// maple_amd/csrc/lds_lane.hip -- measurement kernels: ONE lane per mergeVectors, with the input lists in global memory or
// staged in LDS by the whole wavefront (coalesced, independent loads) and the merged list written to global memory or to LDS.
// The question they answer: is a step of the walk bound by where the lists live?  It is not -- 1.1-1.4 us per step
// of 64 different pairs in every form: the divergent fp64 step body is the bound.  maple_debug_merge_lds times the forms on the
// same pairs (tools/merge_latency_lds.py, profiles/r04_merge_step_lds.txt).
#include "ctx_host.h"

namespace {

template <bool RV, bool U, bool SS, int MODE>
__global__ __launch_bounds__(64) void k_merge_lane_exp(const DevModel *__restrict__ mp, ArenaView av, int n, const int32_t *l1, const double *b1,
                                                       const uint8_t *t1, const int32_t *l2, const double *b2, const uint8_t *t2,
                                                       const uint8_t *ud, uint2 *outWp, double *outAp, int capOut, int32_t *nOut,
                                                       int slabW, int slabA, int outW, int outA, int lanes)
{
    constexpr bool LDSIN = MODE == 1 || MODE == 2;
    constexpr bool LDSOUT = MODE == 2 || MODE == 3;
    __shared__ Lds lds;
    extern __shared__ unsigned long long dyn[];
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const int lane = threadIdx.x;
    const int slabIn = LDSIN ? slabW + slabA : 0;
    const int slab = slabIn + (LDSOUT ? outW + outA : 0);                    // 8-byte units per lane
    // (`lanes` of the 64 lanes carry a pair: the LDS of a compute unit holds that many slabs)
    for (int base = blockIdx.x * lanes; base < n; base += gridDim.x * lanes) {
        const int i = base + lane;
        const bool live = i < n && lane < lanes;
        const int id1 = live ? l1[i] : -1, id2 = live ? l2[i] : -1;
        const int n1 = live ? av.n_ent[id1] : 0, n2 = live ? av.n_ent[id2] : 0;
        const int a1 = live ? av.n_aux[id1] : 0, a2 = live ? av.n_aux[id2] : 0;
        const long long o1 = live ? av.ent_off[id1] : 0, o2 = live ? av.ent_off[id2] : 0;
        const long long x1 = live ? av.aux_off[id1] : 0, x2 = live ? av.aux_off[id2] : 0;
        ListRef L1{av.words + o1, av.aux + x1}, L2{av.words + o2, av.aux + x2};
        bool live2 = live;
        if (LDSIN) {
            const bool fits = live && n1 + n2 <= slabW && a1 + a2 <= slabA;
            __syncthreads();
            for (int j = 0; j < 64; j++) {
                const int fj = __shfl((int)fits, j, 64);
                if (!fj) continue;
                const int jn1 = __shfl(n1, j, 64), jn2 = __shfl(n2, j, 64), ja1 = __shfl(a1, j, 64), ja2 = __shfl(a2, j, 64);
                const long long jo1 = __shfl(o1, j, 64), jo2 = __shfl(o2, j, 64), jx1 = __shfl(x1, j, 64), jx2 = __shfl(x2, j, 64);
                unsigned long long *d = dyn + (size_t)j * slab;
                const unsigned long long *w = (const unsigned long long *)av.words;
                const unsigned long long *a = (const unsigned long long *)av.aux;
                for (int k = lane; k < jn1; k += 64) d[k] = w[jo1 + k];
                for (int k = lane; k < jn2; k += 64) d[jn1 + k] = w[jo2 + k];
                for (int k = lane; k < ja1; k += 64) d[slabW + k] = a[jx1 + k];
                for (int k = lane; k < ja2; k += 64) d[slabW + ja1 + k] = a[jx2 + k];
            }
            __syncthreads();
            {   // (unconditionally LDS, so that the walk's loads are ds_read and not flat_load; a pair that does not fit is skipped)
                unsigned long long *d = dyn + (size_t)lane * slab;
                L1 = ListRef{(const uint2 *)d, (const double *)(d + slabW)};
                L2 = ListRef{(const uint2 *)(d + n1), (const double *)(d + slabW + a1)};
            }
            if (live && !fits) nOut[i] = -9;
            live2 = fits;
        }
        if (live2) {
            uint2 *gw = outWp + (size_t)i * capOut;
            double *ga = outAp + (size_t)i * capOut * 5;
            if (LDSOUT) {
                // (unconditionally LDS: the compiler can then issue ds_write for the merged list -- a store to global memory
                // shares its counter with the loads, and the next load's wait becomes a wait for the store's round trip)
                if (n1 + n2 <= outW) {
                    unsigned long long *d = dyn + (size_t)lane * slab + slabIn;
                    Writer w;
                    w.init((uint2 *)d, (double *)(d + outW));
                    const int r = merge_walk(c, L1, b1[i], t1[i] != 0, L2, b2[i], t2[i] != 0, ud[i] != 0, false, 0, 0, w, nullptr);
                    if (r > 0) {
                        for (int k = 0; k < w.n; k++) gw[k] = w.w[k];
                        for (int k = 0; k < w.na; k++) ga[k] = w.aux[k];
                    }
                    nOut[i] = r;
                } else nOut[i] = -9;
            } else {
                Writer w;
                w.init(gw, ga);
                nOut[i] = merge_walk(c, L1, b1[i], t1[i] != 0, L2, b2[i], t2[i] != 0, ud[i] != 0, false, 0, 0, w, nullptr);
            }
        }
    }
}

}  // namespace

// Times `reps` launches of the one-lane mergeVectors kernel over n pairs (ids of stored lists), input lists walked where they
// are (mode 0) or staged in LDS per lane first (mode 1; slabW words + slabA aux doubles per lane).  *ms = mean per launch;
// nOut[i] = entries of the merged list (or the status).  A measurement aid: nothing is committed to the arena.
extern "C" int maple_debug_merge_lds(maple_ctx *c, int32_t n, const int32_t *l1, const double *b1, const uint8_t *t1, const int32_t *l2,
                                     const double *b2, const uint8_t *t2, const uint8_t *ud, int32_t mode, int32_t slabW, int32_t slabA,
                                     int32_t reps, int32_t grid, float *ms, int32_t *nOut)
{
    if (!c || n <= 0 || !l1 || !l2 || !ms || !nOut || reps < 1 || mode < 0 || mode > 3) return MAPLE_ERR_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int capOut = 0;
    for (int i = 0; i < n; i++) capOut = std::max(capOut, c->h_n_ent[l1[i]] + c->h_n_ent[l2[i]]);
    // (grow-only scratch that frees itself on every way out)
    struct Scratch {
        DevBuf<int32_t> l1, l2, n; DevBuf<double> b1, b2, oa; DevBuf<uint8_t> t1, t2, ud; DevBuf<uint2> ow;
        ~Scratch() { l1.release(); l2.release(); n.release(); b1.release(); b2.release(); oa.release(); t1.release(); t2.release(); ud.release(); ow.release(); }
    } S;
    HIPCK(c, S.l1.reserve_exact(n)); HIPCK(c, S.l2.reserve_exact(n)); HIPCK(c, S.n.reserve_exact(n));
    HIPCK(c, S.b1.reserve_exact(n)); HIPCK(c, S.b2.reserve_exact(n));
    HIPCK(c, S.t1.reserve_exact(n)); HIPCK(c, S.t2.reserve_exact(n)); HIPCK(c, S.ud.reserve_exact(n));
    HIPCK(c, S.ow.reserve_exact((size_t)n * capOut)); HIPCK(c, S.oa.reserve_exact((size_t)n * capOut * 5));
    int32_t *dl1 = S.l1.p, *dl2 = S.l2.p, *dn = S.n.p; double *db1 = S.b1.p, *db2 = S.b2.p, *oa = S.oa.p;
    uint8_t *dt1 = S.t1.p, *dt2 = S.t2.p, *dud = S.ud.p; uint2 *ow = S.ow.p;
    HIPCK(c, hipMemcpy(dl1, l1, n * 4, hipMemcpyHostToDevice)); HIPCK(c, hipMemcpy(dl2, l2, n * 4, hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(db1, b1, n * 8, hipMemcpyHostToDevice)); HIPCK(c, hipMemcpy(db2, b2, n * 8, hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(dt1, t1, n, hipMemcpyHostToDevice)); HIPCK(c, hipMemcpy(dt2, t2, n, hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(dud, ud, n, hipMemcpyHostToDevice));
    const bool ldsIn = mode == 1 || mode == 2, ldsOut = mode == 2 || mode == 3;
    const int outWn = ldsOut ? slabW : 0, outAn = ldsOut ? 2 * slabW : 0;      // (the merged list: as many entries as both inputs, 2 aux each)
    const size_t perLane = (size_t)((ldsIn ? slabW + slabA : 0) + outWn + outAn) * 8;
    int lanes = 64;
    while (perLane * lanes > (size_t)150 * 1024 && lanes > 8) lanes /= 2;
    const size_t dynB = perLane * lanes;
    if (grid < 1) grid = (n + lanes - 1) / lanes;
    hipEvent_t e0, e1;
    HIPCK(c, hipEventCreate(&e0)); HIPCK(c, hipEventCreate(&e1));
    const bool rv = c->dm.useRateVariation, u = c->dm.usingErrorRate;
    if (u) return fail(c, MAPLE_ERR_ARG, "maple_debug_merge_lds: no error model");
#define LAUNCH(RV, L)                                                                                                         \
    do {                                                                                                                     \
        if (dynB) HIPCK(c, hipFuncSetAttribute((const void *)k_merge_lane_exp<RV, false, false, L>,                            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynB));                       \
        k_merge_lane_exp<RV, false, false, L><<<grid, 64, dynB, c->stream>>>(c->d_model, view(c), n, dl1, db1, dt1, dl2, db2, dt2, dud, ow, \
                                                                             oa, capOut, dn, slabW, slabA, outWn, outAn, lanes); \
    } while (0)
    for (int r = 0; r < reps + 1; r++) {
        if (r == 1) HIPCK(c, hipEventRecord(e0, c->stream));
        if (rv) { if (mode == 1) LAUNCH(true, 1); else if (mode == 2) LAUNCH(true, 2); else if (mode == 3) LAUNCH(true, 3); else LAUNCH(true, 0); }
        else { if (mode == 1) LAUNCH(false, 1); else if (mode == 2) LAUNCH(false, 2); else if (mode == 3) LAUNCH(false, 3); else LAUNCH(false, 0); }
    }
#undef LAUNCH
    HIPCK(c, hipEventRecord(e1, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventElapsedTime(ms, e0, e1));
    *ms /= (float)reps;
    HIPCK(c, hipMemcpy(nOut, dn, n * 4, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return MAPLE_OK;
}
