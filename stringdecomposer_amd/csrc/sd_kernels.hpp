// sd_kernels.hpp -- launch wrappers of the HIP kernels (defined in sd_generic.hip / sd_fast.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "sd_device.hpp"

namespace sd {

// generic int32 workgroup-per-chunk family (sd_generic.hip)
int generic_pick_q(int64_t sum_len);
void launch_generic_fill(int Q, int threads, int grid, hipStream_t st, const ChunkDesc* chunks,
                         int chunk_begin, const uint32_t* bases2, const uint32_t* nmask,
                         const uint8_t* tmeta, const int32_t* tend_kd, const int32_t* tend_j,
                         ScoreArgs sc, int rowBytes, uint8_t* ptr, uint64_t row0_base, int32_t* B,
                         int32_t* argB, const uint16_t* grank, int T, int n_tiles, int32_t* Estate);
void launch_generic_trace(int n_sub, hipStream_t st, const ChunkDesc* chunks, int chunk_begin,
                          const uint8_t* ptr, uint64_t row0_base, int rowBytes, const int32_t* B,
                          const int32_t* argB, const int32_t* toff, const int32_t* tlen,
                          DevRec* recs, int32_t* rec_cnt);
void launch_compact(hipStream_t st, const ChunkDesc* chunks, int n_chunks, const int32_t* cnt,
                    int64_t* roff, const DevRec* recs, DevRec* out, int64_t out_cap, bool scan,
                    int32_t* out_chunk = nullptr,
                    long long* scan_ws = nullptr,   // 520 zeroed words that persist between launches: offsets + compaction in
                    long long epoch = 0,            // one launch (sd_scan_compact); a value no earlier launch on scan_ws has used
                    long long* tickets = nullptr);  // host: tickets drawn from scan_ws so far (advanced by the launch)

}  // namespace sd
