// sd_ident.hpp -- in-stream identities (sd_ident.hip): the identity kernel of sd_nw_kernel.hpp run on the compact
// records of a device batch, right behind the compaction, reading the 2-bit reads the DP already holds in HBM.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "sd_device.hpp"

namespace sd {

struct IdentArgs {
    const ChunkDesc* chunks;
    const uint32_t* bases2;
    const uint32_t* nmask;
    const DevRec* dense;            // compact records (chunk-local coordinates)
    const int32_t* rec_chunk;       // chunk of every compact record
    const int64_t* total;           // device: number of compact records (roff[C])
    const int64_t* rec_lo = nullptr;   // device (may be null = 0 / *total): this launch covers the records [*rec_lo, *rec_hi) --
    const int64_t* rec_hi = nullptr;   // the records of a range of chunks (entries of the batch's record offsets)
    int64_t rec_cap;                // records the outputs have room for
    int64_t dense_cap;              // records the compaction had room for (beyond it `dense` is not written)
    int T;                          // templates per record: 1 (own template, `own` maps the DP's template index) or all
    const int32_t* own;
    const unsigned long long* peq;  // [Tmask][5][K], top-aligned (nw_build_masks)
    const int32_t* tlen;
    int Tmask;
    int K;
    int homo;
    int short_max, cap_short, grid_short;   // first launch: segments of up to short_max symbols
    int cap_long, grid_long;                // second launch: the records the first one put on the list
    void* ck;                       // checkpoint workspace, max(grid * 256 * cap) * K * 16 B
    int* ckpos;                     // max(grid * 256 * cap) ints
    int* long_cnt;                  // zeroed by launch_ident
    int32_t* long_list;             // rec_cap entries
    uint32_t* out;                  // [record][T]: (dist << 16) | matches; IDENT_NONE where nothing was computed
    // Pruned homopolymer pass (launch_ident_pruned; round 6): the post-processing reports only the two best templates of the
    // homopolymer-compressed alignments (main.py:137-146), so every pair first gets its distance alone, the identity bounds
    // that follow from it pick the pairs that can be among a record's two best, and only those are aligned in full.
    uint32_t* cand_list = nullptr;  // rec_cap * T entries: output index (record * T + template) of a pair to align in full; a launch
                                    // over the records [r_lo, r_hi) uses the region that starts at r_lo * T (at most T per record)
    int* cand_cnt = nullptr;        // this launch's counter (the caller zeroes it: launch_ident_pruned does, or once per run)
    int grid_cand = 0;
    void* ck_cand = nullptr;        // checkpoint workspace of the candidate stage when it runs beside the next launch's kernels
    int* ckpos_cand = nullptr;      // (launch_ident_pruned_back on a stream of its own); null: ck / ckpos
};
constexpr uint32_t IDENT_NONE = 0xffffffffu;

void launch_ident(hipStream_t st, const IdentArgs& a);
// the same outputs for a.homo != 0, T >= 3 and a.cand_list set, computed as described at IdentArgs::cand_list: the words of
// the pairs that cannot be among a record's two best hold (dist << 16 | the FEWEST matches the distance allows) -- an
// identity strictly below the record's second best, so the stable sort of main.py:145 never sees them in its first two
void launch_ident_pruned(hipStream_t st, const IdentArgs& a);
// the same in two parts, for a caller that runs the candidate stage beside the next slice's kernels: front = distances, the
// long segments in full, bounds + selection (cand_cnt zeroed by the caller before); back = the full alignments of the list
// (any stream that waits for the front; workspace ck_cand / ckpos_cand).  False: this launch is not pruned (front did everything).
bool launch_ident_pruned_front(hipStream_t st, const IdentArgs& a);
void launch_ident_pruned_back(hipStream_t st, const IdentArgs& a);
// bytes of checkpoint workspace (ck) and ints of ckpos for the two launches of launch_ident
size_t ident_ck_lanes(const IdentArgs& a);

}  // namespace sd
