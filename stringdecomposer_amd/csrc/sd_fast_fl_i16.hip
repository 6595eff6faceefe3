// sd_fast_fl_i16.hip -- the variants of sd_fast_fl.hip / sd_fast_fl_long.hip for the packed-int16 cell format
// (scorings whose range does not fit fp16): 3 instead of 4 packed ops per slot behind the first FL slots.
// Fewer levels than the fp16 variants: FL = 16 / 24 for P = 30..40, FL = 24 for P = 42..64.
#include "sd_fast_fill.hpp"

namespace sd {

bool launch_fast_fill_fl_i16(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                             int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                             const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                             int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                             const uint32_t* crank) {
    if (plan.f16 || plan.u16 || plan.wide || plan.P < 30 || plan.floor_slots < 1) return false;
    const int fl = plan.P <= 40 ? (plan.floor_slots <= 16 ? 16 : plan.floor_slots <= 24 ? 24 : 0)
                                : (plan.floor_slots <= 24 ? 24 : 0);
    if (fl == 0) return false;
    const bool ranked = cendoff != nullptr;
#define SD_FL_K(PP, RK, FF)                                                                           \
    {                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill<PP, RK, false, FF>),    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        hipLaunchKernelGGL((sd_fast_fill<PP, RK, false, FF>), dim3(grid), dim3(nw * 64), lds, \
                           st, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, plan.Hx, B,   \
                           argV, ckpt, ckbase, queue, order, cendoff, crank);                        \
        return true;                                                                                 \
    }
#define SD_FL_F(PP, FF)                                                                               \
    if (fl == FF) {                                                                                  \
        if (ranked) SD_FL_K(PP, true, FF) else SD_FL_K(PP, false, FF)                                 \
    }
#define SD_FL_S(PP) case PP: SD_FL_F(PP, 16) SD_FL_F(PP, 24) break;
#define SD_FL_L(PP) case PP: SD_FL_F(PP, 24) break;
    switch (plan.P) {
        SD_FL_S(30) SD_FL_S(31) SD_FL_S(32) SD_FL_S(33) SD_FL_S(34) SD_FL_S(35) SD_FL_S(36) SD_FL_S(37) SD_FL_S(38)
        SD_FL_S(39) SD_FL_S(40)
        SD_FL_L(42) SD_FL_L(44) SD_FL_L(46) SD_FL_L(48) SD_FL_L(52) SD_FL_L(56) SD_FL_L(60) SD_FL_L(64)
        default: break;
    }
#undef SD_FL_S
#undef SD_FL_L
#undef SD_FL_F
#undef SD_FL_K
    return false;
}

}  // namespace sd
