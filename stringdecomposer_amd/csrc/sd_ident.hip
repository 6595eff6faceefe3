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
    // use_list: 0 all pairs of the records [r_lo, r_hi) (long segments go on the list), 1 the records of the long list,
    // 2 the pairs of the candidate list (launch_ident_pruned)
    const int64_t n_rec = use_list == 1 ? (int64_t)min(*a.long_cnt, (int)min(a.rec_cap, (int64_t)0x7fffffff)) : n_all;
    const int64_t n_pairs = use_list == 2 ? (int64_t)*a.cand_cnt : n_rec * a.T;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    void* ckbase = (use_list == 2 && a.ck_cand) ? a.ck_cand : a.ck;
    int* ckpbase = (use_list == 2 && a.ckpos_cand) ? a.ckpos_cand : a.ckpos;
    uint32_t* ckl = reinterpret_cast<uint32_t*>(ckbase) + (size_t)blockIdx.x * (size_t)cap * K * 4 * 256 + threadIdx.x;
    int* ckp = ckpbase + (size_t)blockIdx.x * (size_t)cap * 256 + threadIdx.x;
    const uint32_t* clist = a.cand_list ? a.cand_list + r_lo * a.T : nullptr;   // this launch's region of the candidate list
    for (int64_t p = gid; p < n_pairs; p += stride) {
        const int64_t pp = use_list == 2 ? (int64_t)clist[p] : p;
        const int64_t xl = pp / a.T;
        const int tq = (int)(pp - xl * a.T);
        const int64_t x = use_list == 1 ? (int64_t)a.long_list[xl] : use_list == 2 ? xl : r_lo + xl;
        const DevRec rec = a.dense[x];
        const int t = a.own ? a.own[rec.tmpl] : tq;
        const int ql = rec.end - rec.start + 1;
        const int tl = a.tlen[t];
        const int64_t o = x * a.T + tq;
        if (ql <= 0 || tl <= 0) { a.out[o] = IDENT_NONE; continue; }   // cannot happen for a DP record; the host decides
        if (use_list == 0 && ql > a.short_max) {
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

// ---- pruned homopolymer pass ------------------------------------------------------------------------------------
// Identity of a pair = matches / columns with matches = c - d + nL, columns = c + nL (c kept query symbols, d the edit
// distance, nL the "left" moves of edlib's path: template symbols against a gap).  With tl template symbols:
// nL - nI = tl - c, and the mismatches d - nL - nI >= 0, so  max(0, tl - c) <= nL <= (d + tl - c) / 2, and the identity
// grows with nL: two bounds from the distance alone.
__device__ __forceinline__ void ident_bounds(int d, int c, int tl, double& lb, double& ub, int& m_lb) {
    const int nlo = max(0, tl - c);
    const int nhi = max(nlo, min(tl, (d + tl - c) / 2));
    m_lb = max(0, c - d + nlo);
    lb = (double)m_lb / (double)max(1, c + nlo);
    ub = (double)max(0, c - d + nhi) / (double)max(1, c + nhi);
}

// distance pass: one lane per pair, out = (d << 16) | c; long segments go on the long list as in sd_ident_pairs
template <int K>
__global__ __launch_bounds__(256, 4) void sd_ident_dist(IdentArgs a) {
    extern __shared__ unsigned long long speq[];
    const int eq_lds = (size_t)a.Tmask * 5 * K * 8 <= 60 * 1024;
    if (eq_lds) {
        for (int idx = threadIdx.x; idx < a.Tmask * 5 * K; idx += blockDim.x) speq[idx] = a.peq[idx];
        __syncthreads();
    }
    const int64_t n_tot = *a.total;
    if (n_tot > a.rec_cap || n_tot > a.dense_cap) return;
    const int64_t r_lo = a.rec_lo ? *a.rec_lo : 0;
    const int64_t n_pairs = ((a.rec_hi ? *a.rec_hi : n_tot) - r_lo) * a.T;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += stride) {
        const int64_t xl = p / a.T;
        const int tq = (int)(p - xl * a.T);
        const int64_t x = r_lo + xl;
        const DevRec rec = a.dense[x];
        const int ql = rec.end - rec.start + 1;
        const int tl = a.tlen[tq];
        const int64_t o = x * a.T + tq;
        if (ql <= 0 || tl <= 0) { a.out[o] = IDENT_NONE; continue; }
        if (ql > a.short_max) {
            if (tq == 0) {
                const int slot = atomicAdd(a.long_cnt, 1);
                a.long_list[slot] = (int32_t)x;
            }
            continue;
        }
        const ChunkDesc cd = a.chunks[a.rec_chunk[x]];
        NwQueryPacked q{a.bases2 + cd.woff, cd.noff >= 0 ? a.nmask + cd.noff : nullptr, rec.start};
        int d = 0, c = 0;
        if (eq_lds) nw_dist<K>(q, ql, reinterpret_cast<const uint2*>(speq + (size_t)tq * 5 * K), tl, a.homo != 0, d, c);
        else nw_dist<K>(q, ql, reinterpret_cast<const uint2*>(a.peq + (size_t)tq * 5 * K), tl, a.homo != 0, d, c);
        a.out[o] = (d < 65536 && c < 65536) ? (((uint32_t)d << 16) | (uint32_t)c) : IDENT_NONE;
    }
}

// selection: one wave per record.  L2 = the second largest lower bound of the record's T pairs; a pair whose upper bound
// does not reach it is strictly below the record's two best and keeps (d, fewest matches); the others go on the list.
// (A pair the distance pass could not encode -- IDENT_NONE -- is a candidate: the full kernel decides.)
__global__ __launch_bounds__(256) void sd_ident_select(IdentArgs a) {
    // A wave collects the candidates of the records it goes through in LDS and takes ONE range of the list for all of them:
    // with an atomicAdd per record and half-wave, 20 000 of them per launch met on one counter and the launch took 240 us
    // for 23 M instructions (profiles/r06_ident_pmc.json before this change).
    constexpr int BUF = 1024;
    __shared__ uint32_t cbuf[4][BUF];
    const int64_t n_tot = *a.total;
    if (n_tot > a.rec_cap || n_tot > a.dense_cap) return;
    const int64_t r_lo = a.rec_lo ? *a.rec_lo : 0;
    const int64_t n_rec = (a.rec_hi ? *a.rec_hi : n_tot) - r_lo;
    uint32_t* clist = a.cand_list + r_lo * a.T;
    const int lane = threadIdx.x & 63;
    uint32_t* buf = cbuf[threadIdx.x >> 6];
    int nbuf = 0;   // wave-uniform
    auto flush = [&]() {
        if (nbuf == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(a.cand_cnt, nbuf);
        base = __shfl(base, 0);
        for (int k = lane; k < nbuf; k += 64) clist[base + k] = buf[k];
        nbuf = 0;
    };
    const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool two = a.T <= 128;     // the common case (up to 64 monomers): a lane's two pairs stay in registers between the passes
    for (int64_t xl = wid; xl < n_rec; xl += nw) {
        const int64_t x = r_lo + xl;
        const DevRec rec = a.dense[x];
        const int ql = rec.end - rec.start + 1;
        if (ql <= 0 || ql > a.short_max) continue;   // (nothing computed / the long launch aligns every pair in full)
        if (nbuf + a.T > BUF) flush();
        // top two lower bounds over the T pairs (a duplicate of the maximum counts twice)
        double m1 = -1.0, m2 = -1.0;
        uint32_t w2[2] = {IDENT_NONE, IDENT_NONE};
        double ub2[2] = {2.0, 2.0};
        int ml2[2] = {0, 0};
        for (int t = lane, k = 0; t < a.T; t += 64, ++k) {
            const uint32_t w = a.out[x * a.T + t];
            double lb = -1.0, ub = 2.0;   // not encoded: its lower bound is unknown (it must not raise L2) and it is a candidate
            int ml = 0;
            if (w != IDENT_NONE) ident_bounds((int)(w >> 16), (int)(w & 0xffffu), a.tlen[t], lb, ub, ml);
            if (two && k < 2) { w2[k] = w; ub2[k] = ub; ml2[k] = ml; }
            if (lb > m1) { m2 = m1; m1 = lb; } else if (lb > m2) m2 = lb;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o1 = __shfl_xor(m1, off), o2 = __shfl_xor(m2, off);
            const double hi = fmax(m1, o1);
            const double lo = fmax(fmin(m1, o1), fmax(m2, o2));
            m1 = hi; m2 = lo;
        }
        const double L2 = m2 * (1.0 - 1e-12);
        const bool direct = a.T > BUF;   // (a record with more templates than the buffer holds: straight to the list)
        for (int t0 = 0, k = 0; t0 < a.T; t0 += 64, ++k) {
            const int t = t0 + lane;
            bool cand = false;
            if (t < a.T) {
                uint32_t w;
                double ub;
                int ml;
                if (two) { w = w2[k & 1]; ub = ub2[k & 1]; ml = ml2[k & 1]; }
                else {
                    w = a.out[x * a.T + t];
                    double lb = -1.0;
                    ub = 2.0; ml = 0;
                    if (w != IDENT_NONE) ident_bounds((int)(w >> 16), (int)(w & 0xffffu), a.tlen[t], lb, ub, ml);
                }
                cand = w == IDENT_NONE || ub >= L2;
                if (!cand) a.out[x * a.T + t] = (w & 0xffff0000u) | (uint32_t)ml;
            }
            const unsigned long long bm = __ballot(cand);
            if (bm) {
                const int n = __popcll(bm), at = __popcll(bm & ((1ull << lane) - 1ull));
                if (direct) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(a.cand_cnt, n);
                    base = __shfl(base, 0);
                    if (cand) clist[base + at] = (uint32_t)(x * a.T + t);
                } else {
                    if (cand) buf[nbuf + at] = (uint32_t)(x * a.T + t);
                    nbuf += n;
                }
            }
        }
    }
    flush();
}

size_t ident_ck_lanes(const IdentArgs& a) {
    return std::max(std::max((size_t)a.grid_short, a.ck_cand ? (size_t)0 : (size_t)a.grid_cand) * 256 * (size_t)a.cap_short,
                    (size_t)a.grid_long * 256 * (size_t)a.cap_long);
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

bool launch_ident_pruned_front(hipStream_t st, const IdentArgs& a) {
    if (!a.homo || a.T < 3 || !a.cand_list || !a.cand_cnt) { launch_ident(st, a); return false; }
    const size_t eq_bytes = (size_t)a.Tmask * 5 * (size_t)a.K * 8;
    const size_t lds = eq_bytes <= 60 * 1024 ? eq_bytes : 0;
    (void)hipMemsetAsync(a.long_cnt, 0, sizeof(int), st);
#define SD_IDP(KK)                                                                                                 \
    {                                                                                                              \
        hipLaunchKernelGGL(sd_ident_dist<KK>, dim3(a.grid_short), dim3(256), lds, st, a);                           \
        hipLaunchKernelGGL(sd_ident_pairs<KK>, dim3(a.grid_long), dim3(256), lds, st, a, 1, a.cap_long);            \
        hipLaunchKernelGGL(sd_ident_select, dim3(a.grid_short), dim3(256), 0, st, a);                               \
    }
    switch (a.K) {
        case 1: SD_IDP(1) break;
        case 2: SD_IDP(2) break;
        case 3: SD_IDP(3) break;
        case 4: SD_IDP(4) break;
        case 6: SD_IDP(6) break;
        default: SD_IDP(8) break;
    }
#undef SD_IDP
    return true;
}

void launch_ident_pruned_back(hipStream_t st, const IdentArgs& a) {
    const size_t eq_bytes = (size_t)a.Tmask * 5 * (size_t)a.K * 8;
    const size_t lds = eq_bytes <= 60 * 1024 ? eq_bytes : 0;
    switch (a.K) {
        case 1: hipLaunchKernelGGL(sd_ident_pairs<1>, dim3(a.grid_cand), dim3(256), lds, st, a, 2, a.cap_short); break;
        case 2: hipLaunchKernelGGL(sd_ident_pairs<2>, dim3(a.grid_cand), dim3(256), lds, st, a, 2, a.cap_short); break;
        case 3: hipLaunchKernelGGL(sd_ident_pairs<3>, dim3(a.grid_cand), dim3(256), lds, st, a, 2, a.cap_short); break;
        case 4: hipLaunchKernelGGL(sd_ident_pairs<4>, dim3(a.grid_cand), dim3(256), lds, st, a, 2, a.cap_short); break;
        case 6: hipLaunchKernelGGL(sd_ident_pairs<6>, dim3(a.grid_cand), dim3(256), lds, st, a, 2, a.cap_short); break;
        default: hipLaunchKernelGGL(sd_ident_pairs<8>, dim3(a.grid_cand), dim3(256), lds, st, a, 2, a.cap_short); break;
    }
}

void launch_ident_pruned(hipStream_t st, const IdentArgs& a) {
    if (a.cand_cnt && a.homo && a.T >= 3 && a.cand_list) (void)hipMemsetAsync(a.cand_cnt, 0, sizeof(int), st);
    if (launch_ident_pruned_front(st, a)) launch_ident_pruned_back(st, a);
}

}  // namespace sd
