// sd_ident.hip -- identities of the final TSV in-stream (stringdecomposer/main.py:29-60,107-150 computes them
// with python-edlib from the read text, one alignment per row, 2 T per row with --second-best).
//
// Round 2 ran the identity kernel as a separate step of the post-processing: the read text was uploaded a second
// time (the DP already holds it 2-bit packed), and every call paid ~10 ms of preparation, copies and
// synchronisation around a 1.4-ms kernel.  Here the kernel is one more launch of the batch's stream: right behind
// the compaction it takes the compact records of the batch (chunk-local start / end / template), reads the
// segment from the packed bases of the record's chunk, and leaves one word per (record, template) pair --
// (dist << 16) | matches -- that travels to the host with the records.  The seam merge (main.cpp:287-302) only
// drops records, so a surviving row's identity is the one computed for its chunk record.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "sd_ident.hpp"
#include "sd_nw_kernel.hpp"

namespace sd {

template <int K>
__global__ __launch_bounds__(256, 3) void sd_ident_pairs(IdentArgs a, int use_list, int cap) {
    extern __shared__ unsigned long long speq[];   // [Tmask][5][K] when it fits
    const int eq_lds = (size_t)a.Tmask * 5 * K * 8 <= 60 * 1024;
    if (eq_lds) {
        for (int idx = threadIdx.x; idx < a.Tmask * 5 * K; idx += blockDim.x) speq[idx] = a.peq[idx];
        __syncthreads();
    }
    // A batch with more records than the outputs or the compaction have room for is post-processed from the read text
    // (sd_engine.hip: ident_valid); its compact records are then incomplete -- the compaction skips the chunks that do
    // not fit -- and must not be touched.
    const int64_t n_tot = *a.total;
    if (n_tot > a.rec_cap || n_tot > a.dense_cap) return;
    const int64_t r_lo = a.rec_lo ? *a.rec_lo : 0;
    const int64_t n_all = (a.rec_hi ? *a.rec_hi : n_tot) - r_lo;
    const int64_t n_rec = use_list ? (int64_t)min(*a.long_cnt, (int)min(a.rec_cap, (int64_t)0x7fffffff)) : n_all;
    const int64_t n_pairs = n_rec * a.T;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t* ckl = reinterpret_cast<uint32_t*>(a.ck) + (size_t)blockIdx.x * (size_t)cap * K * 4 * 256 + threadIdx.x;
    int* ckp = a.ckpos + (size_t)blockIdx.x * (size_t)cap * 256 + threadIdx.x;
    for (int64_t p = gid; p < n_pairs; p += stride) {
        const int64_t xl = p / a.T;
        const int tq = (int)(p - xl * a.T);
        const int64_t x = use_list ? (int64_t)a.long_list[xl] : r_lo + xl;
        const DevRec rec = a.dense[x];
        const int t = a.own ? a.own[rec.tmpl] : tq;
        const int ql = rec.end - rec.start + 1;
        const int tl = a.tlen[t];
        const int64_t o = x * a.T + tq;
        if (ql <= 0 || tl <= 0) { a.out[o] = IDENT_NONE; continue; }   // cannot happen for a DP record; the host decides
        if (!use_list && ql > a.short_max) {
            if (tq == 0) {
                const int slot = atomicAdd(a.long_cnt, 1);
                a.long_list[slot] = (int32_t)x;
            }
            continue;
        }
        const ChunkDesc cd = a.chunks[a.rec_chunk[x]];
        NwQueryPacked q{a.bases2 + cd.woff, cd.noff >= 0 ? a.nmask + cd.noff : nullptr, rec.start};
        // two instantiations, so that the masks are read with ds_read from LDS (a pointer that may be either LDS or
        // global makes every mask read a FLAT load with 64-bit address arithmetic per column)
        int d = 0, m = 0;
        bool ok;
        if (eq_lds) ok = nw_pair<K>(q, ql, reinterpret_cast<const uint2*>(speq + (size_t)t * 5 * K), tl, a.homo != 0, ckl, ckp, (size_t)256, cap, d, m);
        else ok = nw_pair<K>(q, ql, reinterpret_cast<const uint2*>(a.peq + (size_t)t * 5 * K), tl, a.homo != 0, ckl, ckp, (size_t)256, cap, d, m);
        a.out[o] = (ok && d < 65536 && m < 65536) ? (((uint32_t)d << 16) | (uint32_t)m) : IDENT_NONE;
    }
}

size_t ident_ck_lanes(const IdentArgs& a) {
    return std::max((size_t)a.grid_short * 256 * (size_t)a.cap_short, (size_t)a.grid_long * 256 * (size_t)a.cap_long);
}

void launch_ident(hipStream_t st, const IdentArgs& a) {
    const size_t eq_bytes = (size_t)a.Tmask * 5 * (size_t)a.K * 8;
    const size_t lds = eq_bytes <= 60 * 1024 ? eq_bytes : 0;
    (void)hipMemsetAsync(a.long_cnt, 0, sizeof(int), st);
#define SD_ID(KK)                                                                                                  \
    {                                                                                                              \
        hipLaunchKernelGGL(sd_ident_pairs<KK>, dim3(a.grid_short), dim3(256), lds, st, a, 0, a.cap_short);          \
        hipLaunchKernelGGL(sd_ident_pairs<KK>, dim3(a.grid_long), dim3(256), lds, st, a, 1, a.cap_long);            \
    }
    switch (a.K) {
        case 1: SD_ID(1) break;
        case 2: SD_ID(2) break;
        case 3: SD_ID(3) break;
        case 4: SD_ID(4) break;
        case 6: SD_ID(6) break;
        default: SD_ID(8) break;
    }
#undef SD_ID
}

}  // namespace sd
