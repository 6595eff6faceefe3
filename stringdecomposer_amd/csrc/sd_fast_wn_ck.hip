// sd_fast_wn_ck.hip -- the compacted form of the multi-wave wide fill (sd_fast_wn_fill.hpp, COMPACT): with
// --ed_thr and more than 128 templates, a chunk is filled by ceil(kept / 128) waves holding exactly its kept
// templates, in their filtered order, instead of W waves holding all of them.
#include "sd_fast_wn_fill.hpp"

namespace sd {

void launch_fast_fill_wn_compact(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, const uint32_t* bases2,
                                 const uint32_t* nmask, const uint32_t* lane_consts, ScoreArgs sc, int32_t* B,
                                 uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order_w, const int* n_ptr,
                                 int n_cu, const uint16_t* klist, const uint8_t* tcodes, const int32_t* toff,
                                 const int32_t* tlen, int wb) {
    // one class of chunks: those whose kept templates need wb waves (order_w / n_ptr: its chunk list and size)
    const size_t lds = ((size_t)wb * (plan.P / 16) * 512 + 64) * sizeof(uint32_t);
    const int per_cu = std::max(1, std::min(8 / wb, (int)((150 * 1024) / lds)));   // two waves per SIMD, LDS
    const int grid = per_cu * n_cu;
    const bool fl = plan.floor_slots >= 1 && plan.floor_slots <= 48 && !plan.full_floor;
#define SD_CK_K(PP, FF)                                                                                             \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, false, FF, true>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, false, FF, true>), dim3(grid), dim3(64 * wb), lds, st, chunks, 0,    \
                           bases2, nmask, nullptr, lane_consts, sc, plan.waves, plan.bf8_match, plan.bf8_mismatch, \
                           B, ckpt, ckbase, queue, order_w, nullptr, nullptr, n_ptr, klist, tcodes, toff, tlen,     \
                           plan.T);                                                                                \
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
