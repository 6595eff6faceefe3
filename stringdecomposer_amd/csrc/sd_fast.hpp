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

constexpr int FAST_R = 32;          // checkpoint / rebase interval (rows)
constexpr int FAST_MAX_SCAN = 6;    // scan steps available in the lane-constant block
constexpr int FAST_LANE_WORDS = 12; // dwords of per-lane constants

// per-lane constant block (dword index), every dword = packed {lo plane, hi plane} int16
enum FastLaneConst {
    FLC_INS0 = 0,     // ins for slot 0 (NEG for the first virtual lane of a template)
    FLC_STARTMASK,    // 0xffff where the virtual lane starts a template (or is idle)
    FLC_ENDOFF,       // (L-1)*del on the last virtual lane of a template, NEG elsewhere
    FLC_SCAN0,        // FLC_SCAN0+s: 0xffff where virtual lane v-2^s belongs to the same template
    FLC_TMPL = FLC_SCAN0 + FAST_MAX_SCAN,  // template index per plane (for the traceback: vlane -> template)
    FLC_START0,       // slot-0 adjustment for row 0 (del on start lanes, 0 elsewhere)
};

struct FastPlan {
    bool ok = false;
    int P = 0;            // slots per virtual lane
    int S = 0;            // inclusive-scan steps: 2^S >= Vmax-1
    int T = 0;
    int split = 0;        // templates [0,split) in the lo plane
    int Lmax = 0;
    int Qk = 0;           // traceback: template cells per lane = ceil(Lmax/64)
    std::vector<int32_t> vlane0;         // first virtual lane of template j
    std::vector<uint32_t> table;         // [5][P/4][64][4] packed (mm - del) values, NEG on padding
    std::vector<uint32_t> lane_consts;   // [64][FAST_LANE_WORDS]
    std::vector<uint16_t> slot_of;       // per template cell x=toff[j]+k: (slot << 7) | vlane
    std::vector<uint8_t> tcodes;         // per template cell: base code
};

// Builds the plan; returns false (with the reason) when the fast family cannot represent the
// input exactly (then the generic family is used).
bool fast_plan_build(const std::vector<std::string>& tseq, ScoreArgs sc, int max_rows,
                     FastPlan& plan, std::string& why);

// number of checkpoint rows of the batch; fills ChunkDesc::pad with each chunk's first checkpoint
int64_t fast_ckpt_rows_total(const FastPlan& plan, std::vector<ChunkDesc>& chunks);

void launch_fast_fill(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                      const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                      const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV,
                      uint32_t* ckpt, int32_t* ckbase);

void launch_fast_trace(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                       const uint32_t* bases2, const uint32_t* nmask, const uint16_t* slot_of,
                       const uint8_t* tcodes, const uint32_t* lane_consts, const int32_t* toff,
                       const int32_t* tlen, ScoreArgs sc, const int32_t* B, const int32_t* argV,
                       const uint32_t* ckpt, const int32_t* ckbase, DevRec* recs,
                       int32_t* rec_cnt);

}  // namespace sd
