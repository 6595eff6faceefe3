// sd_fast_wn_i16.hip -- the multi-wave fills (sd_fast_wn_fill.hpp) with packed int16 cells and int8 table bytes: the form a
// template set beyond one wave takes when its scoring leaves the exact-integer range of fp16 (or its table values are not
// exact in bf8), and the form an engine repeats a batch in after its fp16 range guard tripped.  The reference takes any
// scoring on any monomer set (main.cpp:187-207: long long cells); until round 5 these cases ran on the generic family.
#include "sd_fast_wn_fill.hpp"

namespace sd {

// plain W-wave layout: one template per virtual lane, more than 128 templates
bool launch_fast_fill_wn_i16(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks, int n_chunks,
                             const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table, const uint32_t* lane_consts,
                             ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order,
                             const uint32_t* cendoff, const uint32_t* crank, const int* n_ptr) {
    const int W = plan.waves;
    const bool ranked = cendoff != nullptr;
#define SD_K(PP, RK)                                                                                               \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, RK, PP, false, false, false>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, RK, PP, false, false, false>), dim3(grid), dim3(W * 64), lds, st,   \
                           chunks, n_chunks, bases2, nmask, table, lane_consts, sc, W, plan.bf8_match,             \
                           plan.bf8_mismatch, B, ckpt, ckbase, queue, order, cendoff, crank, n_ptr, nullptr,       \
                           nullptr, nullptr, nullptr, 0);                                                          \
        return true;                                                                                               \
    }
#define SD_C(PP) case PP: if (ranked) SD_K(PP, true) else SD_K(PP, false) break;
    switch (plan.P) {
        SD_C(80) SD_C(96) SD_C(112) SD_C(128) SD_C(144) SD_C(160) SD_C(176) SD_C(192) SD_C(208) SD_C(224)
        default: break;
    }
#undef SD_C
#undef SD_K
    return false;
}

// tiled layout: a template over consecutive virtual lanes
bool launch_fast_fill_wt_i16(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks, int n_chunks,
                             const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table, const uint32_t* lane_consts,
                             ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order,
                             const uint32_t* cendoff, const uint32_t* crank, const int* n_ptr) {
    const int W = plan.waves;
    const bool ranked = cendoff != nullptr;
#define SD_K(PP, RK)                                                                                               \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, RK, PP, false, true, false>),  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, RK, PP, false, true, false>), dim3(grid), dim3(W * 64), lds, st,    \
                           chunks, n_chunks, bases2, nmask, table, lane_consts, sc, W, plan.bf8_match,             \
                           plan.bf8_mismatch, B, ckpt, ckbase, queue, order, cendoff, crank, n_ptr, nullptr,       \
                           nullptr, nullptr, nullptr, 0, plan.H | (((plan.Hx >> 10) & 1) << 8));                    \
        return true;                                                                                               \
    }
#define SD_C(PP) case PP: if (ranked) SD_K(PP, true) else SD_K(PP, false) break;
    switch (plan.P) {
        SD_C(96) SD_C(128) SD_C(160) SD_C(192) SD_C(224)
        default: break;
    }
#undef SD_C
#undef SD_K
    return false;
}

}  // namespace sd
