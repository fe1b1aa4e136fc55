// maple_amd/csrc/frontier_upd.hip -- frontier tier of the SPR search (see frontier.hip): the kernels of the items that arrive with
// needsUpdating == True (M:6982-7091, 7182-7304): lists merged along the path, by one lane per item (k_fr_updating) or, for the few
// with the longest lists, by a wavefront per item (k_fr_updating_wave).
#include "frontier_dev.h"
// (this translation unit's wavefront-wide walks serve the FEW items with the longest lists, one wavefront per compute unit:
// staging areas for lists of 512 entries, 110 KB of LDS)
#define MAPLE_WAVE_CAPW 512
#define MAPLE_WU_IN 512
#include "wave_dev.h"
#include "wave_update.h"

using namespace frt;

namespace {

#include "frontier_upd_lane.inc"

#ifndef FR_UPD_WAVES
#define FR_UPD_WAVES 4                // wavefronts per SIMD k_fr_updating is compiled for (128 registers, the rest of its state in scratch)
#endif

template <bool RV, bool U, bool SS, bool MAT>
__global__ __launch_bounds__(FR_BLOCK) __attribute__((amdgpu_waves_per_eu(FR_UPD_WAVES, FR_UPD_WAVES)))
void k_fr_updating(const DevModel *__restrict__ mp, ArenaViewS av, DevTree T, SearchParams P, FPools fp, int budget, int heavyMin)
{
    __shared__ Lds lds;
    const DevModel &m = *mp;
    stage_model(m, lds);
    Ctx<RV, U, SS> c(m, lds);
    const long long laneId = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long lo = (long long)fp.ctr->loU, hi = (long long)fp.ctr->hiU;
    // wavefronts of items moving down first, then wavefronts of items crawling up (k_fr_sort_level; heavy items are not listed)
    const long long nDown = (long long)fp.ctr->permDown, nUp = (long long)fp.ctr->permUp, padDown = (nDown + 63) & ~63ll;
    // the items with long lists first, 16 to a wavefront (lanes 0-15), then wavefronts of 64 moving down, then of 64 crawling up
    const long long nDownB = (long long)fp.ctr->permDownB, nUpB = (long long)fp.ctr->permUpB;
    const long long vDownB = ((nDownB + 15) >> 4) << 6, vUpB = ((nUpB + 15) >> 4) << 6, vBig = vDownB + vUpB;
    unsigned long long nU = 0, bU = 0;
    (void)heavyMin;
    for (long long v0 = laneId; v0 < vBig + padDown + nUp; v0 += (long long)gridDim.x * blockDim.x) {
        long long i;
        if (v0 < vBig) {
            const bool up = v0 >= vDownB;
            const long long w = up ? v0 - vDownB : v0;
            const int l = (int)(w & 63);
            const long long j = (w >> 6) * 16 + l;
            if (l >= 16 || j >= (up ? nUpB : nDownB)) continue;
            i = lo + (up ? fp.perm2[(hi - lo) - 1 - j] : fp.perm2[j]);
        } else {
            const long long v = v0 - vBig;
            if (v >= nDown && v < padDown) continue;
            i = lo + (v < nDown ? fp.perm[v] : fp.perm[(hi - lo) - 1 - (v - padDown)]);
        }
#ifdef MAPLE_SPR_PROFILE
        const long long t0 = wall_clock64();
        int sz = 0;
        {
            const FItem &it = fp.U[i];
            if (it.dir != 3) {
                const NodeRec &r1 = T.nd[it.t1];
                const int other = it.dir == 0 ? it.t1 : (it.dir == 1 ? r1.c1 : r1.c0);
                const int lw = T.nd[other].lower;
                sz = (lw >= 0 ? flen(av, fp, ftree(lw)) : 0) + (fvalid(it.hPassed) ? flen(av, fp, it.hPassed) : 0);
            }
        }
#endif
        fr_upd_item_lane<RV, U, SS, MAT>(c, av, T, P, fp, budget, laneId, i, &bU);
        nU++;
#ifdef MAPLE_SPR_PROFILE
        {
            const unsigned long long dt = (unsigned long long)(wall_clock64() - t0);
            const int b = sz < 64 ? 0 : sz < 96 ? 1 : sz < 128 ? 2 : sz < 192 ? 3 : sz < 256 ? 4 : sz < 384 ? 5 : sz < 512 ? 6 : 7;
            atomicAdd(&fp.ctr->dbgCnt[b], 1ull); atomicAdd(&fp.ctr->dbgT[b], dt); atomicMax(&fp.ctr->dbgMax[b], dt);
        }
#endif
    }
    // (what the launch did, for the roofline of the bench line: one atomic per wavefront)
    for (int off = 32; off > 0; off >>= 1) {
        nU += ((unsigned long long)(uint32_t)__shfl_down((int)(nU >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)nU, off, 64);
        bU += ((unsigned long long)(uint32_t)__shfl_down((int)(bU >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)bU, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && nU) { atomicAdd(&fp.ctr->itemsU, nU); atomicAdd(&fp.ctr->bytesU, bU); }
}

#define FRW_KERNEL k_fr_updating_wave
#define FRW_PERM perm4
#define FRW_COUNT permHeavy2
#include "frontier_upd_wave.inc"
}  // namespace


int fr_launch_updating(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P, const FPools &fp,
                       int budget, int heavyMin)
{
    const bool rv_ = c->dm.useRateVariation, u_ = c->dm.usingErrorRate, ss_ = c->dm.errorRateSiteSpecific;
#define FR_UPD_LAUNCH(RV, U, SS)                                                                                        \
    do {                                                                                                               \
        if (fp.mat) k_fr_updating<RV, U, SS, true><<<grid, FR_BLOCK, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin); \
        else k_fr_updating<RV, U, SS, false><<<grid, FR_BLOCK, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin);       \
    } while (0)
    if (!rv_ && !u_) FR_UPD_LAUNCH(false, false, false);
    else if (rv_ && !u_) FR_UPD_LAUNCH(true, false, false);
    else if (!rv_ && u_ && !ss_) FR_UPD_LAUNCH(false, true, false);
    else if (!rv_ && u_ && ss_) FR_UPD_LAUNCH(false, true, true);
    else if (rv_ && u_ && !ss_) FR_UPD_LAUNCH(true, true, false);
    else FR_UPD_LAUNCH(true, true, true);
#undef FR_UPD_LAUNCH
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}

int fr_launch_updating_wave(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                            const FPools &fp, int budget, int heavyMin, long long laneBase)
{
    FR_DISPATCH3(c, k_fr_updating_wave, <<<grid, 64, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin, laneBase));
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}
