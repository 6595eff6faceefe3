// sd_fast_u16.hip -- the narrow fill with the start-term maximum in every slot (what sd_fast.hip instantiates for fp16 and
// int16 cells) for the biased-u16 cell format (CellOps<CF_U16>, sd_fast_dev.hpp): slot counts below 30, template sets whose
// floor_slots exceeds every FL level of sd_fast_fl_u16.hip, sets with 1-bp templates (the FLC_ONE form), SD_FLAG_FULL_FLOOR.
#include "sd_fast_fill.hpp"

namespace sd {

void launch_fast_fill_full_u16(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                               int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                               const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                               int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                               const uint32_t* crank) {
    const bool ranked = cendoff != nullptr;
    const bool has1 = ((plan.Hx >> 10) & 1) != 0;   // 1-bp templates
#define SD_U16_K(PP, RK, ON)                                                                          \
    {                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill<PP, RK, CF_U16, PP, ON>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        hipLaunchKernelGGL((sd_fast_fill<PP, RK, CF_U16, PP, ON>), dim3(grid), dim3(nw * 64), lds, st, chunks, \
                           n_chunks, bases2, nmask, table, lane_consts, sc, plan.Hx, B, argV, ckpt,   \
                           ckbase, queue, order, cendoff, crank);                                    \
    }
#define SD_U16(PP)                                                                                    \
    case PP:                                                                                         \
        if (has1) { if (ranked) SD_U16_K(PP, true, true) else SD_U16_K(PP, false, true) }             \
        else { if (ranked) SD_U16_K(PP, true, false) else SD_U16_K(PP, false, false) }                \
        break;
    switch (plan.P) {
        SD_U16(4) SD_U16(8) SD_U16(12) SD_U16(16) SD_U16(20) SD_U16(24) SD_U16(28) SD_U16(30)
        SD_U16(31) SD_U16(32) SD_U16(33) SD_U16(34) SD_U16(35) SD_U16(36) SD_U16(37) SD_U16(38)
        SD_U16(39) SD_U16(40) SD_U16(42) SD_U16(44) SD_U16(46) SD_U16(48) SD_U16(52) SD_U16(56)
        SD_U16(60) SD_U16(64)
        default: break;
    }
#undef SD_U16
#undef SD_U16_K
}

}  // namespace sd
