// sd_device.hpp -- data structures shared between the host engine and the HIP kernels.
#pragma once

#include <cstdint>

namespace sd {

// One chunk of a read on the device.
struct ChunkDesc {
    uint32_t woff;     // word offset of its 2-bit bases (16 bases / uint32, base i at bits 2*(i&15))
    int32_t n;         // rows (bases) in the chunk
    int32_t noff;      // word offset of its N mask (32 bases / uint32), or -1 if the chunk has no N
    uint32_t pad;
    uint64_t row0;     // prefix sum of n over the batch (row index of row 0 of this chunk)
};

// scoring, kept as plain ints in kernel arguments
struct ScoreArgs {
    int32_t ins, del, mismatch, match;
    // fp16 cell formats only (sd_fast_dev.hpp: F16Guard): the magnitude every stored cell must stay below for the
    // arithmetic to be exact, and the device flag a wave raises when one does not (the host then repeats the batch
    // with integer cells)
    int32_t guard_lim = 0;
    int* guard_flag = nullptr;
    // fast fills: the stored cells are rebased on B every rebase_mask + 1 rows (FastPlan::rebase: 128, or 64 where
    // only the shorter period keeps a template set and scoring inside the fp16 range)
    int32_t rebase_mask = 127;
};

// Device-side record as emitted by the traceback kernels (emission order = reverse read order).
struct DevRec {
    int32_t tmpl, start, end, score;
};

constexpr int32_t NEG_INF32 = -0x3fffffff;

// per-cell metadata byte of the generic kernel
constexpr uint8_t CELL_CODE_MASK = 0x07;  // 0..3 = ACGT, 4 = N, 7 = padding (matches nothing)
constexpr uint8_t CELL_START = 0x08;      // k == 0
constexpr uint8_t CELL_END = 0x10;        // k == L-1

}  // namespace sd
