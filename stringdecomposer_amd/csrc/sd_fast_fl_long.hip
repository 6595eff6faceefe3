// sd_fast_fl_long.hip -- the variants of sd_fast_fl.hip for P = 42..64 slots per lane (sets of 13 to 23 monomers
// of ~170 bp, or longer monomers), in their own translation unit so that the library builds in parallel.
#include "sd_fast_fill.hpp"

// (sd_fast_fl_long_u16.hip compiles this file again for the biased-u16 cell format)
#ifndef SD_FL_STEP
#define SD_FL_STEP 0      /* one floor level for every row (fp16 cells); the u16 units set 4: three levels by read symbol */
#endif
#ifndef SD_FL_CF
#define SD_FL_CF CF_F16
#define SD_FL_ENTRY_LONG launch_fast_fill_fl_long
#define SD_FL_TAKES(plan) ((plan).f16)
#endif

namespace sd {

bool SD_FL_ENTRY_LONG(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                              int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                              const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                              int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                              const uint32_t* crank) {
    if (!SD_FL_TAKES(plan) || plan.wide || plan.P <= 40 || plan.floor_slots < 1) return false;
    const int fl = plan.floor_slots <= 16 ? 16 : plan.floor_slots <= 24 ? 24 : plan.floor_slots <= 32 ? 32 : 0;
    if (fl == 0) return false;
    const bool ranked = cendoff != nullptr;
    // the level of every read symbol (two bits each from bit 22 of Hx): 0 = FL, 1 = FL - step, 2 = FL - 2 steps, 3 = FL - 3 steps
    int hx = plan.Hx;
    if (SD_FL_STEP > 0)
        for (int b = 0; b < 5; ++b) {
            int lv = 0;
            while (lv < 3 && fl - (lv + 1) * SD_FL_STEP >= std::max(1, plan.floor_sym[b])) ++lv;
            hx |= lv << (22 + 2 * b);
        }
#define SD_FL_K(PP, RK, FF)                                                                           \
    {                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill<PP, RK, SD_FL_CF, FF, false, SD_FL_STEP>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        hipLaunchKernelGGL((sd_fast_fill<PP, RK, SD_FL_CF, FF, false, SD_FL_STEP>), dim3(grid), dim3(nw * 64), lds,  \
                           st, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, hx, B,   \
                           argV, ckpt, ckbase, queue, order, cendoff, crank);                        \
        return true;                                                                                 \
    }
#define SD_FL_F(PP, FF)                                                                               \
    if (fl == FF) {                                                                                  \
        if (ranked) SD_FL_K(PP, true, FF) else SD_FL_K(PP, false, FF)                                 \
    }
#define SD_FL(PP)                                                                                     \
    case PP:                                                                                         \
        SD_FL_F(PP, 16) SD_FL_F(PP, 24) SD_FL_F(PP, 32)                                               \
        break;
    switch (plan.P) {
        SD_FL(42) SD_FL(44) SD_FL(46) SD_FL(48) SD_FL(52) SD_FL(56) SD_FL(60) SD_FL(64)
        default: break;
    }
#undef SD_FL
#undef SD_FL_F
#undef SD_FL_K
    return false;
}

}  // namespace sd
