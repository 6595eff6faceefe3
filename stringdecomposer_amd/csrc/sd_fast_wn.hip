// sd_fast_wn.hip -- multi-wave wide variant of the fast fill: template sets beyond 128 templates
// (hundreds of monomers; the reference takes any monomer set, main.cpp:187-207).
//
// A workgroup of W waves (W = ceil(T / 128), up to 8 = 1024 templates) owns one chunk at a time; wave w
// holds templates [128 w, 128 w + 128), ONE template per virtual lane as in sd_fast_wide.hip (P >= Lmax
// slots in registers, no cross-lane chain).  Rows are strictly sequential because B_i is the maximum over
// ALL template ends (main.cpp:184-186): every wave reduces its own ends (DPP), the wave results meet in
// LDS, one workgroup barrier per row, and every wave derives the same B_i / arg-max.
//
// The per-symbol score table of the single-wave kernel (5 x 128 x P bf8 bytes = 112 KB of LDS) cannot be
// replicated per wave, so LDS holds, per wave, the template base CODES only (one byte per slot and plane,
// symbol-independent, 22.5 KB at P = 176), and the row's table bytes are produced on the fly: the eight
// bf8 bytes {code c == row symbol ? match : mismatch (c = 0..4), -inf for padding (code 7)} sit in two
// registers and ONE v_perm_b32 turns a dword of four codes into the four bf8 table bytes of two slots;
// v_cvt_scalef32_pk_f16_bf8 then expands a slot's {lo, hi} pair as in the single-wave kernel.  4.5 VALU
// ops per slot (perm/2 + cvt + max + add + max3) instead of 4.
//
// Outputs as sd_fast_fill_wide (checkpoints every FAST_R rows: [checkpoint][wave][P][64]; one word per
// row, here (B_i << 10) | (wave << 7 | virtual lane)) -> the same traceback kernel (bshift = 10).
#include "sd_fast_wn_fill.hpp"

namespace sd {

void launch_fast_fill_wn(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                         const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                         const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                         int* queue, const int* order, int n_cu, const uint32_t* cendoff, const uint32_t* crank,
                         const int* n_ptr) {
    const int W = plan.waves;
    const int per_cu = std::max(1, 8 / W);          // 213 VGPRs: two waves per SIMD, eight per CU
    const int grid = std::min(n_chunks, per_cu * n_cu);
    const size_t lds = ((size_t)W * (plan.P / 16) * 512 + 64) * sizeof(uint32_t);
    const bool ranked = cendoff != nullptr;
    if (!plan.f16) {   // integer cells (sd_fast_wn_i16.hip)
        (void)launch_fast_fill_wn_i16(plan, st, grid, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt, ckbase,
                                      queue, order, cendoff, crank, n_ptr);
        return;
    }
    if (!plan.full_floor &&
        launch_fast_fill_wn_fl(plan, st, grid, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt,
                               ckbase, queue, order, cendoff, crank, n_ptr))
        return;
#define SD_FILLWN_K(PP, RK)                                                                                      \
    {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, RK>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, RK>), dim3(grid), dim3(W * 64), lds, st, chunks, n_chunks, bases2, \
                           nmask, table, lane_consts, sc, W, plan.bf8_match, plan.bf8_mismatch, B, ckpt, ckbase,  \
                           queue, order, cendoff, crank, n_ptr, nullptr, nullptr, nullptr, nullptr, 0);             \
    }
#define SD_FILLWN(PP)                                                   \
    case PP:                                                            \
        if (ranked) SD_FILLWN_K(PP, true) else SD_FILLWN_K(PP, false)   \
        break;
    switch (plan.P) {
        SD_FILLWN(80) SD_FILLWN(96) SD_FILLWN(112) SD_FILLWN(128) SD_FILLWN(144) SD_FILLWN(160)
        SD_FILLWN(176) SD_FILLWN(192) SD_FILLWN(208) SD_FILLWN(224)
        default: break;
    }
#undef SD_FILLWN
#undef SD_FILLWN_K
}

}  // namespace sd
