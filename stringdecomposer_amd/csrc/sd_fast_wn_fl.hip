// sd_fast_wn_fl.hip -- instantiations of sd_fast_fill_wn (sd_fast_wn_fill.hpp) that take the maximum of a slot's
// diagonal input with the start term only in the first 48 slots (see sd_fast_fl.hip for the argument): 3.5
// instead of 4.5 packed ops per slot behind them.
#include "sd_fast_wn_fill.hpp"

namespace sd {

bool launch_fast_fill_wn_fl(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks,
                            int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                            const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                            int* queue, const int* order, const uint32_t* cendoff, const uint32_t* crank,
                            const int* n_ptr) {
    if (!plan.wide || plan.waves < 2 || plan.floor_slots < 1 || plan.floor_slots > 48) return false;
    const int W = plan.waves;
    const bool ranked = cendoff != nullptr;
#define SD_WNFL_K(PP, RK)                                                                                           \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, RK, 48>),                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, RK, 48>), dim3(grid), dim3(W * 64), lds, st, chunks, n_chunks,      \
                           bases2, nmask, table, lane_consts, sc, W, plan.bf8_match, plan.bf8_mismatch, B, ckpt,   \
                           ckbase, queue, order, cendoff, crank, n_ptr, nullptr, nullptr, nullptr, nullptr, 0);       \
        return true;                                                                                               \
    }
#define SD_WNFL(PP)                                                 \
    case PP:                                                       \
        if (ranked) SD_WNFL_K(PP, true) else SD_WNFL_K(PP, false)   \
        break;
    switch (plan.P) {
        SD_WNFL(80) SD_WNFL(96) SD_WNFL(112) SD_WNFL(128) SD_WNFL(144) SD_WNFL(160) SD_WNFL(176) SD_WNFL(192)
        SD_WNFL(208) SD_WNFL(224)
        default: break;
    }
#undef SD_WNFL
#undef SD_WNFL_K
    return false;
}

}  // namespace sd
