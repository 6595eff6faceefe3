// sd_fast_fl.hip -- instantiations of sd_fast_fill (sd_fast_fill.hpp) that take the maximum of a slot's
// diagonal input with the start term only in the first FL slots of a lane.
//
// In the recurrence  S'[x] = max3(S'[x-1], max(S[x-1], KB) + tbl[x], S[x])  the term KB + tbl[x] (KB = the
// row's start term B_i + del joined with the lane's lazy carry) is one more candidate of slot x.  S' is a
// prefix maximum along the slots, and tbl takes two values per read symbol, so KB + tbl[x] can only raise
// S'[x] where tbl[x] exceeds every tbl[x'] of an earlier slot x' >= 1 of the lane -- slot 1 and the first
// slot holding the read's base (main.cpp:187-207 evaluates the start term in every cell; this is the same
// maximum with the dominated candidates left out).  fast_plan_build finds the last such slot over all lanes
// and the five read symbols (FastPlan::floor_slots: 16 on the synthetic 12-monomer set, 24 on the DXZ1
// monomers of the reference's test data); behind it a slot costs 2 packed ops instead of 3.
//
// Only slot counts P >= 30 (P = 30..40 here, 42..64 in sd_fast_fl_long.hip) and the fp16 cell format get these
// variants; everything else runs the full kernels of sd_fast.hip.
#include "sd_fast_fill.hpp"

// (sd_fast_fl_u16.hip compiles this file again for the biased-u16 cell format: SD_FL_CF = CF_U16, its own entry names)
#ifndef SD_FL_STEP
#define SD_FL_STEP 0      /* one floor level for every row (fp16 cells); the u16 units set 4: three levels by read symbol */
#endif
#ifndef SD_FL_CF
#define SD_FL_CF CF_F16
#define SD_FL_ENTRY launch_fast_fill_fl
#define SD_FL_ENTRY_LONG launch_fast_fill_fl_long
#define SD_FL_TAKES(plan) ((plan).f16)
#endif

namespace sd {

bool SD_FL_ENTRY(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                         int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                         const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                         int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                         const uint32_t* crank) {
    if (!SD_FL_TAKES(plan) || plan.wide || plan.P < 30 || plan.floor_slots < 1) return false;
    if (plan.P > 40)
        return SD_FL_ENTRY_LONG(plan, st, grid, nw, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B,
                                        argV, ckpt, ckbase, queue, order, cendoff, crank);
    int fl = 0;
    for (int c : {12, 16, 20, 24, 28})
        if (plan.floor_slots <= c && c + 2 < plan.P) { fl = c; break; }
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
    if (fl == FF && FF + 2 < PP) {                                                                   \
        if (ranked) SD_FL_K(PP, true, FF) else SD_FL_K(PP, false, FF)                                 \
    }
#define SD_FL(PP)                                                                                     \
    case PP:                                                                                         \
        SD_FL_F(PP, 12) SD_FL_F(PP, 16) SD_FL_F(PP, 20) SD_FL_F(PP, 24) SD_FL_F(PP, 28)               \
        break;
    switch (plan.P) {
        SD_FL(30) SD_FL(31) SD_FL(32) SD_FL(33) SD_FL(34) SD_FL(35) SD_FL(36) SD_FL(37) SD_FL(38) SD_FL(39) SD_FL(40)
        default: break;
    }
#undef SD_FL
#undef SD_FL_F
#undef SD_FL_K
    return false;
}

}  // namespace sd
