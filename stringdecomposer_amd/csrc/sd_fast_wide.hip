// sd_fast_wide.hip -- wide variant of the fast fill for large template sets (up to 128 templates,
// e.g. the 64-monomer suprachromosomal-family configuration): ONE template per virtual lane
// (P >= Lmax slots, no cross-lane chain, no lazy carry).
//
// The (mm - del - ins) table is int8 ({lo plane, hi plane} byte pairs) so that 128 x 176 cells x
// 5 read symbols fit the 160 KB LDS; it is streamed 16 slots at a time into a double-buffered
// 16-register window.  5 VALU ops per cell pair (the add is two SDWA byte adds).  Tail slots behind
// a template's last cell hold the table byte -128: in the row-shifted domain
// S_new[last] >= max(S_old[last], KB) - 127, so such a slot is a transparent copy of the last cell
// and the template end is always read from slot P-1.
//
// Same outputs as sd_fast_fill (packed B/arg-max words, checkpoints) -> same traceback kernel.
// Replaces reference stringdecomposer/src/main.cpp:171-216 like sd_fast_fill does.
#include "sd_fast_wide_fill.hpp"

namespace sd {

void launch_fast_fill_wide(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                           const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                           const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt,
                           int32_t* ckbase, int* queue, const int* order, int n_cu,
                           const uint32_t* cendoff, const uint32_t* crank) {
    const int NW = 8;
    const int grid = std::min((n_chunks + NW - 1) / NW, n_cu);  // persistent: one workgroup per CU (LDS)
    const size_t lds = (size_t)5 * (plan.P / 16) * 512 * sizeof(uint32_t) + 128;   // + FairShare's words
    const bool ranked = cendoff != nullptr;
    if (!plan.full_floor &&
        launch_fast_fill_wide_fl(plan, st, grid, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt,
                                 ckbase, queue, order, cendoff, crank))
        return;
#define SD_FILLW_K(PP, RK, HF)                                                                       \
    {                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wide<PP, RK, HF>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        hipLaunchKernelGGL((sd_fast_fill_wide<PP, RK, HF>), dim3(grid), dim3(NW * 64), lds, st,       \
                           chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt, ckbase, \
                           queue, order, cendoff, crank);                                            \
    }
#define SD_FILLW(PP)                                                                                 \
    case PP:                                                                                         \
        if (plan.f16) {                                                                              \
            if (ranked) SD_FILLW_K(PP, true, true) else SD_FILLW_K(PP, false, true)                  \
        } else {                                                                                     \
            if (ranked) SD_FILLW_K(PP, true, false) else SD_FILLW_K(PP, false, false)                \
        }                                                                                            \
        break;
    switch (plan.P) {
        SD_FILLW(80) SD_FILLW(96) SD_FILLW(112) SD_FILLW(128) SD_FILLW(144) SD_FILLW(160)
        SD_FILLW(176) SD_FILLW(192) SD_FILLW(208) SD_FILLW(224)
        default: break;
    }
#undef SD_FILLW
#undef SD_FILLW_K
}

}  // namespace sd
