// sd_fast.hpp -- layout plan + launch wrappers of the fast kernel family (sd_fast.hip).
//
// Lane layout ("virtual lanes"): a wave owns one chunk.  Every 32-bit VGPR holds two int16 DP
// cells, one of the "lo plane" (bits 0..15) and one of the "hi plane" (bits 16..31); plane h,
// lane l is virtual lane v = 64*h + l and owns P consecutive slots of the flattened template axis
// (P registers).  Template j occupies V_j = ceil(L_j / P) consecutive virtual lanes of ONE plane,
// starting at slot 0 of its first virtual lane; the unused tail slots of its last virtual lane are
// "transparent" padding cells that carry E[L_j-1] to the last slot.  Templates 0..s-1 live in the
// lo plane, s..T-1 in the hi plane, in order, so that "smallest virtual lane" == "first template".
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "sd_device.hpp"

namespace sd {

constexpr int FAST_R = 32;          // checkpoint interval (rows)
constexpr int FAST_REBASE = 128;    // int16 rebase interval (rows), a multiple of FAST_R
constexpr int FAST_LANE_WORDS = 8;  // dwords of per-lane constants

// per-lane constant block (dword index), every dword = packed {lo plane, hi plane} int16
enum FastLaneConst {
    FLC_STARTMASK = 0, // 0xffff where the virtual lane starts a template (or is idle)
    FLC_CONTMASK,      // 0xffff where virtual lane v-1 belongs to the same template
    FLC_ENDOFF,        // (L-1)*del on the last virtual lane of a template, NEG elsewhere
    FLC_ROW0,          // row-0 adjustment of slot 0: ins+del on start lanes, ins elsewhere
    FLC_TMPL,          // template index per plane (for the traceback: vlane -> template)
    FLC_CONT2,         // 0xffff where virtual lane v-2 belongs to the same template
    FLC_ENDALL,        // (L-1)*del on EVERY virtual lane of a template, NEG on idle lanes
    FLC_ONE,           // 0xffff where the virtual lane holds a 1-bp template: its end is slot 0, not the last slot
};

struct FastPlan {
    bool ok = false;
    bool wide = false;    // one template per virtual lane, int8 table streamed from LDS (P > 64)
    bool f16 = false;     // narrow layout with packed-fp16 cells (3 ops per cell pair, v_pk_maximum3_f16)
    bool u16 = false;     // narrow layout with biased unsigned 16-bit cells (CellOps<CF_U16>: the same 3 ops, the add a plain
                          // v_add_u32, exact range +-15 k); takes precedence over f16 and int16 where the range fits
    int u16_lim = 0;      // |stored cell| the u16 format's run-time guard allows (>= range_bound)
    int P = 0;            // slots per virtual lane
    int P4 = 0;           // P rounded up to a multiple of 4 (LDS table row)
    int H = 0;            // carry hops of the cross-lane chain: Vmax-1
    int Hx = 0;           // what the narrow fills get: H | (carry scan through ds_bpermute) << 8 | (scan form in the last
                          // round too, developer A/B) << 9 | (the set has 1-bp templates) << 10 | idle lane << 16
    int T = 0;
    int split = 0;        // templates [0,split) in the lo plane
    int Lmax = 0;
    int Qk = 0;           // traceback: template cells per lane = ceil(Lmax/64)
    int waves = 1;        // waves per chunk: 1, or ceil(T/128) for the multi-wave wide layout (sd_fast_wn.hip)
    bool tiled = false;   // multi-wave layout with templates tiled over consecutive virtual lanes (sd_fast_wt.hip); wide is set too
    bool filter_only = false;   // tiled, and the whole set does NOT fit eight waves: only with --ed_thr, every chunk in the
                                // compacted form (its kept templates re-dealt, sd_tiled_place); a chunk whose kept templates
                                // do not fit either raises the guard flag and the batch is repeated on the generic family
    int bshift = 7;       // B words: (B_i << bshift) | arg-max (wave << 7 | virtual lane)
    int rebase = FAST_REBASE;  // rows between two rebases of the stored cells (ScoreArgs::rebase_mask + 1; the narrow fills read it from Hx bit 11): 128 or 64
    int range_bound = 0;       // proven bound on |stored cell| between two rebases (fp16 formats need <= 2040)
    bool full_floor = false;   // launch the fills that take the start-term maximum in every slot (SD_FLAG_FULL_FLOOR: A/B, parity test)
    int floor_slots = 0;  // last slot of a lane whose diagonal input needs the max with the start term (see sd_fast_fill)
    int floor_sym[5] = {0, 0, 0, 0, 0};   // the same per read symbol (A C G T N); floor_slots = their maximum
    bool table_nonneg = false;            // every table value (mm - del - ins) >= 0: the fills may apply the floor in place (sd_fast_fill: FLS)
    uint32_t bf8_match = 0, bf8_mismatch = 0;   // multi-wave wide layout: the two table values as bf8 bytes (f16) or int8 bytes (integer cells)
    std::vector<int32_t> vlane0;         // first virtual lane of template j
    std::vector<uint32_t> table;         // narrow: [5][P4/4][64][4] packed int16 (mm - del - ins), NEG on padding
                                         // wide:   [5][P/16][2][64][4] dwords of int8 {lo,hi} pairs, -128 on padding
    std::vector<uint32_t> lane_consts;   // [64][FAST_LANE_WORDS]
    std::vector<uint32_t> slot_of;       // per template cell x=toff[j]+k: (wave << 16) | (slot << 7) | vlane
    std::vector<uint8_t> tcodes;         // per template cell: base code
    std::vector<int32_t> end_vlane;      // virtual lane holding the end of template j
    std::vector<int32_t> end_off;        // (L_j - 1) * del
    // traceback, second form (sd_fast_trace2.hip): packed 16-bit recomputation, two blocks per wave
    bool tr2_ok = false;                 // one wave per chunk in the fill (narrow layout, wide / tiled layouts of one wave), templates <= 256 bp, scores inside the 16-bit tagged range
    int tr2_qm = 0;                      // ceil(Lmax / 64): registers per lane at the widest level
    int tr2_xlim = 0;                    // |E' - base| a checkpoint cell may have (run-time check of the range proof)
    int tr2_bound = 0;                   // the proven bound on |E' - base| for this template set and scoring (<= tr2_xlim)
    std::vector<uint32_t> tr2_tab;       // per template and level: table [5][QQ][32] + checkpoint map [QQ][2][32]
};

// the tables of sd_fast_trace_pk for a built narrow plan (tr2_ok = false when it does not apply)
void fast_plan_trace2(const std::vector<std::string>& tseq, ScoreArgs sc, FastPlan& plan);
bool launch_fast_trace2(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks, const uint32_t* bases2,
                        const uint32_t* nmask, const uint32_t* lane_consts, const uint8_t* tcodes, const int32_t* toff,
                        const int32_t* tlen, ScoreArgs sc, const int32_t* B, const uint32_t* ckpt, const int32_t* ckbase,
                        const uint32_t* tr2_tab, DevRec* recs, int32_t* rec_cnt, int* queue, const int* order, int n_cu);

// slots-per-virtual-lane values the fill kernels are instantiated for
static const int FAST_P_LIST[] = {4, 8, 12, 16, 20, 24, 28, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39,
                                  40, 42, 44, 46, 48, 52, 56, 60, 64};
static const int FAST_WIDE_P_LIST[] = {80, 96, 112, 128, 144, 160, 176, 192, 208, 224};
static const int FAST_TILED_P_LIST[] = {96, 128, 160, 192, 224};   // slots per lane of the tiled multi-wave layout

// Builds the plan; returns false (with the reason) when the fast family cannot represent the
// input exactly (then the generic family is used).
bool fast_plan_build(const std::vector<std::string>& tseq, ScoreArgs sc, int max_rows,
                     FastPlan& plan, std::string& why, bool allow_f16 = true, bool allow_tr2 = true,
                     bool filter_only_ok = false,    // --ed_thr is on: a set beyond eight waves may take the filter-only form
                     bool allow_u16 = true);         // false: SD_FLAG_NO_U16 (the narrow layout's fp16 / int16 cells as in rounds 1-5)

// --ed_thr prefilter on the device (sd_filter.hip): infix edit distances, kept set and ranks per
// chunk -> per-chunk lane constants of the fast family (cendoff, crank: [chunk][64] packed {lo,hi}
// int16; grank == nullptr) or the rank table of the generic family (grank: [chunk][T], 0xffff = dropped)
void build_peq(const std::vector<std::string>& tseq, std::vector<unsigned long long>& peq);
void launch_edthr_filter(hipStream_t st, const ChunkDesc* chunks, int n_chunks, int T, int Lmax, int ed_thr,
                         const uint32_t* bases2, const uint32_t* nmask, const unsigned long long* peq,
                         const int32_t* tlen, const int32_t* end_vlane, const int32_t* end_off,
                         int32_t* dist, uint32_t* cendoff, uint32_t* crank, uint16_t* grank, int waves = 1,
                         uint16_t* kpos = nullptr, uint16_t* klist = nullptr, int32_t* nkept = nullptr,
                         int uniform_half = -1,    // 0 / 1: every template ends in the low / high half of word ceil(L/64)-1
                                                   // and all have that many words (sd_hw_dist_u); -1: general kernel
                         const int32_t* vlane0 = nullptr);   // first virtual lane of each template (narrow layout)
// --ed_thr with more than 128 templates: the chunk order split into W classes by ceil(kept templates / 128)
// --ed_thr on the tiled multi-wave layout: per chunk, the kept templates' lanes (sd_filter.hip: sd_tiled_place)
void launch_tiled_place(hipStream_t st, int n_chunks, int T, int P, int W, const uint16_t* klist, int32_t* nkept,
                        const int32_t* tlen, uint16_t* kpos, uint32_t* lane_t, int* overflow_flag = nullptr);
void launch_split_order(hipStream_t st, const int* order, int n, const int32_t* nkept, int* orders, int* counts,
                        int W);   // orders: [W][n] -- class w-1 = the chunks that need w waves, in the given order

// number of checkpoint rows of the batch; fills ChunkDesc::pad with each chunk's first checkpoint
int64_t fast_ckpt_rows_total(const FastPlan& plan, std::vector<ChunkDesc>& chunks);

void launch_fast_fill(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                      const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                      const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV,
                      uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order, int n_cu,
                      const uint32_t* cendoff, const uint32_t* crank, size_t min_lds = 0);

// variants with the start-term maximum in the first slots of a lane only (sd_fast_fl.hip); false = not covered
bool launch_fast_fill_fl(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                         int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                         const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                         int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                         const uint32_t* crank);

// the same for the biased-u16 cell format (sd_fast_fl_u16.hip, sd_fast_fl_long_u16.hip) and its full-floor kernels (sd_fast_u16.hip)
bool launch_fast_fill_fl_u16(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                             int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                             const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                             int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                             const uint32_t* crank);
bool launch_fast_fill_fl_long_u16(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                                  int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                                  const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                                  int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                                  const uint32_t* crank);
// the u16 kernels with ONE floor level for every row (scorings with a negative table value: sd_fast_fl_u16s.hip)
bool launch_fast_fill_fl_u16s(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                              int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                              const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                              int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                              const uint32_t* crank);
bool launch_fast_fill_fl_long_u16s(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                                   int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                                   const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                                   int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                                   const uint32_t* crank);
void launch_fast_fill_full_u16(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                               int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                               const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                               int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                               const uint32_t* crank);

bool launch_fast_fill_fl_i16(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                             int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                             const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                             int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                             const uint32_t* crank);
bool launch_fast_fill_fl_long(const FastPlan& plan, hipStream_t st, int grid, int nw, size_t lds, const ChunkDesc* chunks,
                              int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                              const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV, uint32_t* ckpt,
                              int32_t* ckbase, int* queue, const int* order, const uint32_t* cendoff,
                              const uint32_t* crank);

// wide variant (sd_fast_wide.hip), called by launch_fast_fill when plan.wide
void launch_fast_fill_wide(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                           const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                           const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt,
                           int32_t* ckbase, int* queue, const int* order, int n_cu,
                           const uint32_t* cendoff, const uint32_t* crank);

bool launch_fast_fill_wide_fl(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks,
                              int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                              const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                              int* queue, const int* order, const uint32_t* cendoff, const uint32_t* crank);

// multi-wave wide variant (sd_fast_wn.hip): more than 128 templates, W = plan.waves waves per chunk
void launch_fast_fill_wn(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                         const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                         const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                         int* queue, const int* order, int n_cu, const uint32_t* cendoff, const uint32_t* crank,
                         const int* n_ptr = nullptr);   // n_ptr: the number of chunks lives on the device (order = a class list)
// tiled multi-wave variant (sd_fast_wt.hip): templates over consecutive virtual lanes, W = plan.waves >= 1 waves per chunk
void launch_fast_fill_wt(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                         const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                         const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                         int* queue, const int* order, int n_cu, const uint32_t* cendoff, const uint32_t* crank,
                         const int* n_ptr = nullptr);   // n_ptr: the number of chunks lives on the device (order = a class list)
// --ed_thr on the tiled layout: one class of chunks, filled by wb waves holding their kept templates (lane_t: sd_tiled_place)
void launch_fast_fill_wt_compact(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, const uint32_t* bases2,
                                 const uint32_t* nmask, const uint32_t* lane_consts, ScoreArgs sc, int32_t* B,
                                 uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order_w, const int* n_ptr,
                                 int n_cu, const uint32_t* lane_t, const uint8_t* tcodes, const int32_t* toff,
                                 const int32_t* tlen, int wb);
// --ed_thr, more than 128 templates: one class of chunks, filled by wb waves holding their kept templates (sd_fast_wn_ck.hip)
void launch_fast_fill_wn_compact(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, const uint32_t* bases2,
                                 const uint32_t* nmask, const uint32_t* lane_consts, ScoreArgs sc, int32_t* B,
                                 uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order_w, const int* n_ptr,
                                 int n_cu, const uint16_t* klist, const uint8_t* tcodes, const int32_t* toff,
                                 const int32_t* tlen, int wb);   // wb: waves per chunk of this class

// the same two layouts with int16 cells / int8 table bytes (sd_fast_wn_i16.hip): plan.f16 == false
bool launch_fast_fill_wn_i16(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks, int n_chunks,
                             const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table, const uint32_t* lane_consts,
                             ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order,
                             const uint32_t* cendoff, const uint32_t* crank, const int* n_ptr);
bool launch_fast_fill_wt_i16(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks, int n_chunks,
                             const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table, const uint32_t* lane_consts,
                             ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order,
                             const uint32_t* cendoff, const uint32_t* crank, const int* n_ptr);

bool launch_fast_fill_wn_fl(const FastPlan& plan, hipStream_t st, int grid, size_t lds, const ChunkDesc* chunks,
                            int n_chunks, const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                            const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, uint32_t* ckpt, int32_t* ckbase,
                            int* queue, const int* order, const uint32_t* cendoff, const uint32_t* crank,
                            const int* n_ptr = nullptr);

void launch_fast_trace(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                       const uint32_t* bases2, const uint32_t* nmask, const uint32_t* slot_of,
                       const uint8_t* tcodes, const uint32_t* lane_consts, const int32_t* toff,
                       const int32_t* tlen, ScoreArgs sc, const int32_t* B, const int32_t* argV,
                       const uint32_t* ckpt, const int32_t* ckbase, DevRec* recs,
                       int32_t* rec_cnt, int* queue, const int* order, int n_cu,
                       const uint16_t* klist = nullptr, const uint16_t* kpos = nullptr, const int32_t* nkept = nullptr,
                       const uint32_t* tr2_tab = nullptr,   // device copy of FastPlan::tr2_tab: the packed two-block form where it applies
                       const uint32_t* lane_t = nullptr);   // compacted tiled chunks: the lane table (kpos = first lanes then;
                                                            // FastPlan::filter_only: every chunk is one, or skipped)

}  // namespace sd
