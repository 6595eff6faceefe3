// sd_fast_wt.hip -- tiled multi-wave variant of the fast fill: template sets that neither the narrow layout (at most
// 128 virtual lanes of up to 64 slots) nor the wide ones (one template per virtual lane of up to 224 slots) hold --
// e.g. sixty 340-bp templates, or two hundred 500-bp ones (the reference takes any monomer set, main.cpp:187-207).
//
// The kernel is sd_fast_fill_wn (sd_fast_wn_fill.hpp: W waves per chunk, template base codes in LDS, table bytes made
// on the fly, one workgroup barrier per row for B_i) with its TILED parameter: a template lies over V = ceil(L / P)
// consecutive virtual lanes of one plane of one wave and the deletion chain crosses the lanes through the lazily
// applied carry of the narrow fills (sd_fast_fill.hpp).  fast_plan_build() picks P from FAST_TILED_P_LIST as the slot
// count with the least SIMD time per row at the occupancy it gets, among those that fit eight waves and the LDS of a CU;
// W may be 1.
//
// Outputs as the multi-wave wide fill: checkpoints [checkpoint][wave][P][64] (true values, max(cell, carry)), one word
// per row (B_i << 10) | (wave << 7 | virtual lane) -> sd_fast_trace (bshift = 10), whose cell -> (wave, lane, slot) map
// (FastPlan::slot_of) knows the tiling.  --ed_thr: the chunks whose kept templates need fewer than W waves are filled by
// that many (launch_fast_fill_wt_compact below: per-chunk lane table from sd_tiled_place), the others by the ranked form
// (per-chunk end offsets and ranks on every lane of a template, sd_rank_keep).
#include "sd_fast_wn_fill.hpp"

namespace sd {

void launch_fast_fill_wt(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                         const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                         const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                         int* queue, const int* order, int n_cu, const uint32_t* cendoff, const uint32_t* crank,
                         const int* n_ptr) {
    const int W = plan.waves;
    const size_t lds = ((size_t)W * (plan.P / 16) * 512 + 64) * sizeof(uint32_t);
    // two waves per SIMD (the register budget of the kernel), and as many workgroups per CU as their LDS allows
    const int per_cu = std::max(1, std::min(8 / W, (int)((size_t)160 * 1024 / lds)));
    const int grid = std::min(n_chunks, per_cu * n_cu);
    const bool ranked = cendoff != nullptr;
    if (!plan.f16) {   // integer cells (sd_fast_wn_i16.hip)
        (void)launch_fast_fill_wt_i16(plan, st, grid, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt, ckbase,
                                      queue, order, cendoff, crank, n_ptr);
        return;
    }
    const bool fl48 = !plan.full_floor && plan.floor_slots >= 1 && plan.floor_slots <= 48;
#define SD_FILLWT_K(PP, RK, FLV)                                                                                   \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, RK, FLV, false, true>),        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, RK, FLV, false, true>), dim3(grid), dim3(W * 64), lds, st, chunks,  \
                           n_chunks, bases2, nmask, table, lane_consts, sc, W, plan.bf8_match, plan.bf8_mismatch, B, \
                           ckpt, ckbase, queue, order, cendoff, crank, n_ptr, nullptr, nullptr, nullptr, nullptr, 0,   \
                           plan.H | (((plan.Hx >> 10) & 1) << 8));                                                                                \
    }
#define SD_FILLWT(PP)                                                               \
    case PP:                                                                        \
        if (fl48) { if (ranked) SD_FILLWT_K(PP, true, 48) else SD_FILLWT_K(PP, false, 48) } \
        else { if (ranked) SD_FILLWT_K(PP, true, PP) else SD_FILLWT_K(PP, false, PP) }      \
        break;
    switch (plan.P) {
        SD_FILLWT(96) SD_FILLWT(128) SD_FILLWT(160) SD_FILLWT(192) SD_FILLWT(224)
        default: break;
    }
#undef SD_FILLWT
#undef SD_FILLWT_K
}

// --ed_thr: the chunks whose kept templates need wb < W waves, filled by wb waves that hold exactly those (the point of
// the reference's prefilter, main.cpp:128-149: less DP work) -- the compacted form of sd_fast_wn_ck.hip with the
// per-chunk lane table of sd_tiled_place in place of "one kept template per lane".
void launch_fast_fill_wt_compact(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, const uint32_t* bases2,
                                 const uint32_t* nmask, const uint32_t* lane_consts, ScoreArgs sc, int32_t* B,
                                 uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order_w, const int* n_ptr,
                                 int n_cu, const uint32_t* lane_t, const uint8_t* tcodes, const int32_t* toff,
                                 const int32_t* tlen, int wb) {
    const size_t lds = ((size_t)wb * (plan.P / 16) * 512 + 64) * sizeof(uint32_t);
    const int per_cu = std::max(1, std::min(8 / wb, (int)((size_t)160 * 1024 / lds)));
    const int grid = per_cu * n_cu;
    const bool fl48 = !plan.full_floor && plan.floor_slots >= 1 && plan.floor_slots <= 48;
#define SD_CKT_K(PP, FLV)                                                                                          \
    {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill_wn<PP, false, FLV, true, true>),      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sd_fast_fill_wn<PP, false, FLV, true, true>), dim3(grid), dim3(64 * wb), lds, st, chunks, \
                           0, bases2, nmask, nullptr, lane_consts, sc, plan.waves, plan.bf8_match, plan.bf8_mismatch, \
                           B, ckpt, ckbase, queue, order_w, nullptr, nullptr, n_ptr, nullptr, tcodes, toff, tlen,   \
                           plan.T, plan.H | (((plan.Hx >> 10) & 1) << 8), lane_t);                                  \
    }
#define SD_CKT(PP)                                              \
    case PP:                                                    \
        if (fl48) SD_CKT_K(PP, 48) else SD_CKT_K(PP, PP)        \
        break;
    switch (plan.P) {
        SD_CKT(96) SD_CKT(128) SD_CKT(160) SD_CKT(192) SD_CKT(224)
        default: break;
    }
#undef SD_CKT
#undef SD_CKT_K
}

}  // namespace sd
