// sd_fast_wide_fl.hip -- instantiations of sd_fast_fill_wide (sd_fast_wide_fill.hpp) that take the maximum of a
// slot's diagonal input with the start term only in the first FL slots (see sd_fast_fl.hip for the argument).
// In the wide layout a virtual lane holds a whole template, so FastPlan::floor_slots is the latest first
// occurrence of a base in any template: behind it a slot costs 3 packed ops (table conversion, add, maximum3)
// instead of 4.  fp16 cells only.
#include "sd_fast_wide_fill.hpp"

namespace sd {

bool launch_fast_fill_wide_fl(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks,
                              int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                              const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                              int* queue, const int* order, const uint32_t* cendoff, const uint32_t* crank) {
    if (!plan.f16 || !plan.wide || plan.waves > 1 || plan.floor_slots < 1) return false;
    const int fl = plan.floor_slots <= 32 ? 32 : plan.floor_slots <= 64 ? 64 : 0;
    if (fl == 0) return false;
    const bool ranked = cendoff != nullptr;
#define SD_WFL_K(PP, RK, FF)                                                                           \
    {                                                                                                 \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wide<PP, RK, true, FF>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
        hipLaunchKernelGGL((sd_fast_fill_wide<PP, RK, true, FF>), dim3(grid), dim3(512), lds, st,      \
                           chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt, ckbase,  \
                           queue, order, cendoff, crank);                                             \
        return true;                                                                                  \
    }
#define SD_WFL(PP)                                                                                     \
    case PP:                                                                                          \
        if (fl == 32) { if (ranked) SD_WFL_K(PP, true, 32) else SD_WFL_K(PP, false, 32) }              \
        else { if (ranked) SD_WFL_K(PP, true, 64) else SD_WFL_K(PP, false, 64) }                       \
        break;
    switch (plan.P) {
        SD_WFL(80) SD_WFL(96) SD_WFL(112) SD_WFL(128) SD_WFL(144) SD_WFL(160) SD_WFL(176) SD_WFL(192) SD_WFL(208)
        SD_WFL(224)
        default: break;
    }
#undef SD_WFL
#undef SD_WFL_K
    return false;
}

}  // namespace sd
