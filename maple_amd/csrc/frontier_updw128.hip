// maple_amd/csrc/frontier_updw128.hip -- frontier tier of the SPR search: the list-updating items of a level that are walked a
// WAVEFRONT each, small size class (every list the item touches has at most 128 entries: most items of every level but the
// ones next to the root).  27 KB of LDS per wavefront -- five wavefronts per compute unit, 1 280 items at a time -- against
// 110 KB and one per compute unit for the 512-entry class in frontier_upd.hip.  Same code (frontier_upd_wave.inc), other sizes.
#include "frontier_dev.h"
#define MAPLE_WAVE_CAPW FR_WAVE_SMALL_CAPW
#define MAPLE_WU_IN FR_WAVE_SMALL_IN
#include "wave_dev.h"
#include "wave_update.h"

using namespace frt;

namespace {

#include "frontier_upd_lane.inc"
#define FRW_KERNEL k_fr_updating_wave_s
#define FRW_PERM perm3
#define FRW_COUNT permHeavy
#include "frontier_upd_wave.inc"
}  // namespace

int fr_launch_updating_wave_small(maple_ctx *c, hipStream_t s, int grid, const ArenaViewS &av, const DevTree &T, const SearchParams &P,
                                  const FPools &fp, int budget, int heavyMin, long long laneBase)
{
    FR_DISPATCH3(c, k_fr_updating_wave_s, <<<grid, 64, 0, s>>>(c->d_model, av, T, P, fp, budget, heavyMin, laneBase));
    HIPCK(c, hipGetLastError());
    return MAPLE_OK;
}
