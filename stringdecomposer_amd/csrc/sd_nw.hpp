// sd_nw.hpp -- launch wrapper of the batched NW identity kernel (sd_nw.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace sd {

// columns per register-resident block of the identity kernel (sd_nw_kernel.hpp): S * K * 4 history registers;
// a pair of q columns needs ceil(q / S) - 1 checkpoint slots
__host__ __device__ constexpr int nw_block_cols(int K) { return K <= 1 ? 16 : K == 2 ? 12 : K == 3 ? 9 : K == 4 ? 6 : K <= 6 ? 4 : 3; }

// One lane per (segment, template) pair; K = 64-bit words per template (ceil(max tlen / 64): 1, 2, 3, 4, 6, 8).
// ck / ckpos: grid * 256 lanes x cap checkpoint slots x (K x 16 B + 4 B); peq: [T][5][K], top-aligned
// (nw_build_masks); seg_idx (optional): the segments of this launch.
void launch_nw_pairs(int K, hipStream_t st, int grid, const uint8_t* seq, const int64_t* seg_start,
                     const int32_t* seg_len, const int32_t* seg_idx, int64_t n_seg, int T, const int32_t* pair_tmpl,
                     const unsigned long long* peq, const int32_t* tlen, int homo, int cap, void* ck, int* ckpos,
                     int32_t* dist, int32_t* matches);
// templates of 513 .. 2048 bp (sd_nw_long.hip): a pair across K lanes, systolic over the columns.  plist: the pair ids of
// the launch; ck: grid * (block_threads / 64) * (64 / lpp) pair slots x cap x 5 x lpp dwords, lpp = 16 (K <= 16) or 32
void launch_nw_long(int K, hipStream_t st, int grid, int block_threads, const uint8_t* seq, const int64_t* seg_start,
                    const int32_t* seg_len, const int64_t* plist, int64_t n_pairs, int T, const int32_t* pair_tmpl,
                    const unsigned long long* peq, const int32_t* tlen, int homo, int qcap, int cap, void* ck,
                    int32_t* dist, int32_t* matches);
size_t nw_long_lds_bytes(int lpp, int qcap, int block_threads);
int nw_long_slots(int qmax, int K);
void nw_build_masks(const std::vector<std::string>& ts, int K, std::vector<unsigned long long>& peq,
                    std::vector<int32_t>& tl);

// Host driver (sd_nw.hip): identity of segments of a text (the concatenation of `spans`) against templates on
// the device; see the definition.
int nw_identity_device(const std::vector<std::pair<const char*, int64_t>>& spans, const int64_t* seg_start,
                       const int32_t* seg_len, int64_t n_seg, const std::vector<std::string>& tmpl,
                       const int32_t* pair_tmpl, bool homo, int device, int threads, int32_t* dist,
                       int32_t* matches);
// accumulated over the device identity calls of the process: preparation + staging, uploads, kernel, downloads
void nw_stage_seconds(double out[4]);

}  // namespace sd
