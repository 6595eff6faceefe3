// sd_fast_wn_ck.hip -- the compacted form of the multi-wave wide fill (sd_fast_wn_fill.hpp, COMPACT): with
// --ed_thr and more than 128 templates, a chunk whose kept templates number at most 128 is filled by ONE wave
// holding exactly those templates, in their filtered order, instead of W waves holding all of them.
#include "sd_fast_wn_fill.hpp"

namespace sd {

void launch_fast_fill_wn_compact(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, const uint32_t* bases2,
                                 const uint32_t* nmask, const uint32_t* lane_consts, ScoreArgs sc, int32_t* B,
                                 uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order1, const int* n_ptr,
                                 int n_cu, const uint16_t* klist, const uint8_t* tcodes, const int32_t* toff,
                                 const int32_t* tlen) {
    const size_t lds = ((size_t)(plan.P / 16) * 512 + 64) * sizeof(uint32_t);
    const int grid = 7 * n_cu;                       // 22.6 KB of LDS per one-wave workgroup at P = 176: seven per CU
    const bool fl = plan.floor_slots >= 1 && plan.floor_slots <= 48 && !getenv("SD_FILL_FULLFLOOR");
#define SD_CK_K(PP, FF)                                                                                             \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, false, FF, true>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, false, FF, true>), dim3(grid), dim3(64), lds, st, chunks, 0, bases2, \
                           nmask, nullptr, lane_consts, sc, plan.waves, plan.bf8_match, plan.bf8_mismatch, B, ckpt, \
                           ckbase, queue, order1, nullptr, nullptr, n_ptr, klist, tcodes, toff, tlen);              \
    }
#define SD_CK(PP)                                           \
    case PP:                                               \
        if (fl) SD_CK_K(PP, 48) else SD_CK_K(PP, PP)        \
        break;
    switch (plan.P) {
        SD_CK(80) SD_CK(96) SD_CK(112) SD_CK(128) SD_CK(144) SD_CK(160) SD_CK(176) SD_CK(192) SD_CK(208) SD_CK(224)
        default: break;
    }
#undef SD_CK
#undef SD_CK_K
}

}  // namespace sd
