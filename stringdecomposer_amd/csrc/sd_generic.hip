// sd_generic.hip -- generic (fallback) device path: any template set, int32 arithmetic, any scoring
// inside the INF bound.  One workgroup per chunk, the flattened template axis blocked over the
// threads (Q cells per thread, in registers; beyond 32 768 cells in tiles of 1024 x 32 cells with the
// previous row in HBM), 2-bit priority-encoded back-pointers written to HBM, pointer-walking
// traceback.  With the --ed_thr prefilter the B_i reduction skips dropped templates and breaks ties by
// the chunk's filtered order (per-chunk rank table).
//
// Replaces AlignPartClassicDP (reference stringdecomposer/src/main.cpp:151-270):
//   fill      main.cpp:171-216   -> sd_generic_fill
//   traceback main.cpp:217-269   -> sd_generic_trace
//
// The recurrence is evaluated in the shifted domain E[i][x] = D[i][x] - k*del (k = position of
// cell x inside its template), where it loses every dependence on k:
//   E[i][x] = max( B_i + mm,  E[i-1][x-1] + (mm - del),  E[i-1][x] + ins,  E[i][x-1] )
// (first term only for k == 0), so the in-row deletion chain is a plain segmented prefix maximum.
// The four equality tests of the traceback are invariant under the shift, so the 2-bit pointers
// are exactly the reference's choices, in its priority order DEL > INS > DIAG > START.
#include <hip/hip_runtime.h>

#include "sd_device.hpp"
#include "sd_kernels.hpp"

namespace sd {

namespace {
// Key of a template end in the B_i reduction: value in the high half; in the low half what breaks ties --
// the smallest template index (main.cpp:211-215, 230-236) or, with the --ed_thr prefilter, the smallest rank
// in the chunk's filtered order (main.cpp:141-147).  A dropped template (rank 0xffff) never takes part.
__device__ __forceinline__ long long end_key(int32_t val, int32_t j, const uint16_t* grank, size_t chunk_base) {
    if (grank) {
        const uint32_t rk = grank[chunk_base + (size_t)j];
        if (rk == 0xffffu) return (long long)NEG_INF32 * 4294967296LL;
        return (long long)val * 4294967296LL + (long long)(((0xffffu - rk) << 16) | (uint32_t)j);
    }
    return (long long)val * 4294967296LL + (long long)(0x7fffffff - j);
}
__device__ __forceinline__ int32_t key_tmpl(long long kb, bool ranked) {
    const uint32_t lo = (uint32_t)(kb & 0xffffffffLL);
    return ranked ? (int32_t)(lo & 0xffffu) : (int32_t)(0x7fffffff - (int32_t)lo);
}
}  // namespace

template <int Q>
__global__ __launch_bounds__(1024) void sd_generic_fill(
    const ChunkDesc* __restrict__ chunks, int chunk_begin, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint8_t* __restrict__ tmeta,
    const int32_t* __restrict__ tend_kd, const int32_t* __restrict__ tend_j, ScoreArgs sc,
    int rowBytes, uint8_t* __restrict__ ptr, uint64_t row0_base, int32_t* __restrict__ Bout,
    int32_t* __restrict__ argBout, const uint16_t* __restrict__ grank, int T) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, nw = blockDim.x >> 6;
    const int c = chunk_begin + blockIdx.x;
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    const int ins = sc.ins;
    const int mD = sc.match - sc.del, xD = sc.mismatch - sc.del;

    __shared__ int32_t waveV[16];
    __shared__ int32_t waveF[16];
    __shared__ long long waveKey[16];

    uint8_t meta[Q];
    bool anyStart = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        meta[q] = tmeta[t * Q + q];
        anyStart |= (meta[q] & CELL_START) != 0;
    }
    int32_t E[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) E[q] = NEG_INF32;

    uint8_t* prow = ptr + (cd.row0 - row0_base) * (uint64_t)rowBytes + (size_t)t * (Q / 4);
    const uint64_t boff = cd.row0 + (uint64_t)c;  // B / argB have n+1 entries per chunk

    int32_t pdEdge = NEG_INF32;  // final E of cell x0-1 in the previous row
    int32_t Bi = 0;
    uint32_t wbits = 0, nbits = 0;
    for (int i = 0; i < n; ++i) {
        if ((i & 15) == 0) wbits = bases2[cd.woff + (i >> 4)];
        int r = (wbits >> (2 * (i & 15))) & 3;
        if (cd.noff >= 0) {
            if ((i & 31) == 0) nbits = nmask[cd.noff + (i >> 5)];
            if ((nbits >> (i & 31)) & 1) r = 4;
        }
        const bool row0 = (i == 0);
        const int32_t Bd = Bi + sc.del;

        // pass A: local (in-thread) prefix maxima
        int32_t loc[Q];
        int32_t run = NEG_INF32, pd = pdEdge;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const bool st = (meta[q] & CELL_START) != 0;
            const int32_t mmd = ((meta[q] & CELL_CODE_MASK) == r) ? mD : xD;
            int32_t cand;
            if (row0) cand = st ? mmd + sc.del : mmd;            // main.cpp:171-182
            else if (st) cand = Bd + mmd;                        // k == 0: start term only
            else cand = max(max(pd, Bd) + mmd, E[q] + ins);      // start/diag, ins
            run = st ? cand : max(cand, run);                    // deletion chain
            loc[q] = run;
            pd = E[q];
        }
        // segmented inclusive max-scan of the thread totals over the workgroup
        int32_t v = run;
        int fl = anyStart ? 1 : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t v2 = __shfl_up(v, off);
            const int f2 = __shfl_up(fl, off);
            if (lane >= off) {
                if (!fl) v = max(v, v2);
                fl |= f2;
            }
        }
        if (lane == 63) { waveV[w] = v; waveF[w] = fl; }
        __syncthreads();
        int32_t cw = NEG_INF32;
        for (int w2 = 0; w2 < w; ++w2) cw = waveF[w2] ? waveV[w2] : max(cw, waveV[w2]);
        const int32_t S = fl ? v : max(v, cw);
        int32_t Sprev = __shfl_up(S, 1);
        if (lane == 0) Sprev = cw;  // final E of cell x0-1 in THIS row

        // pass B: apply the carry, derive the pointers, collect template-end values
        int32_t left = Sprev;
        pd = pdEdge;
        bool before = true;
        uint64_t codes = 0;
        long long best = (long long)NEG_INF32 * 4294967296LL;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const bool st = (meta[q] & CELL_START) != 0;
            if (st) before = false;
            const int32_t Ef = before ? max(loc[q], Sprev) : loc[q];
            const int32_t old = E[q];  // E[i-1][x]
            const int32_t mmd = ((meta[q] & CELL_CODE_MASK) == r) ? mD : xD;
            int pc;
            if (!st && Ef == left) pc = 0;                          // DEL   main.cpp:242
            else if (!row0 && Ef == old + ins) pc = 1;          // INS   main.cpp:245 (k==0 too)
            else if (!row0 && !st && Ef == pd + mmd) pc = 2;        // DIAG  main.cpp:249
            else pc = 3;                                            // START main.cpp:253 / STOP
            codes |= (uint64_t)pc << (2 * q);
            if (meta[q] & CELL_END) {
                const int x = t * Q + q;
                best = max(best, end_key(Ef + tend_kd[x], tend_j[x], grank, (size_t)c * T));
            }
            left = Ef;
            pd = old;
            E[q] = Ef;
        }
        pdEdge = Sprev;
        {
            uint8_t* p = prow + (size_t)i * rowBytes;
            if (Q == 4) *p = (uint8_t)codes;
            else if (Q == 8) *reinterpret_cast<uint16_t*>(p) = (uint16_t)codes;
            else if (Q == 16) *reinterpret_cast<uint32_t*>(p) = (uint32_t)codes;
            else *reinterpret_cast<uint64_t*>(p) = codes;
        }
        // B_{i+1} = max over template ends, first template on ties (main.cpp:184-186, 230-236)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off));
        if (lane == 0) waveKey[w] = best;
        __syncthreads();
        long long kb = waveKey[0];
        for (int w2 = 1; w2 < nw; ++w2) kb = max(kb, waveKey[w2]);
        Bi = (int32_t)(kb >> 32);
        if (t == 0) {
            Bout[boff + i + 1] = Bi;
            argBout[boff + i + 1] = key_tmpl(kb, grank != nullptr);
        }
    }
}

// Template sets beyond 32 768 cells (hundreds of monomers): the same row-synchronous sweep with the flattened
// template axis cut into tiles of 1024 threads x 32 cells that one workgroup processes in order, every row.
// The previous row lives in HBM / L2 (`Estate`, one row per resident chunk) instead of registers; the in-row
// deletion chain and the diagonal input cross a tile boundary through two carried values.
__global__ __launch_bounds__(1024) void sd_generic_fill_tiled(
    const ChunkDesc* __restrict__ chunks, int chunk_begin, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint8_t* __restrict__ tmeta,
    const int32_t* __restrict__ tend_kd, const int32_t* __restrict__ tend_j, ScoreArgs sc,
    int rowBytes, uint8_t* __restrict__ ptr, uint64_t row0_base, int32_t* __restrict__ Bout,
    int32_t* __restrict__ argBout, const uint16_t* __restrict__ grank, int T, int n_tiles,
    int32_t* __restrict__ Estate) {
    constexpr int Q = 32;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int c = chunk_begin + blockIdx.x;
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    const int ins = sc.ins;
    const int mD = sc.match - sc.del, xD = sc.mismatch - sc.del;
    const size_t cells = (size_t)n_tiles * 1024 * Q;
    int32_t* Est = Estate + (size_t)blockIdx.x * cells;

    __shared__ int32_t waveV[16];
    __shared__ int32_t waveF[16];
    __shared__ int32_t waveOld[16];   // previous-row value of each wave's last cell
    __shared__ long long waveKey[16];
    __shared__ int32_t tileS, tileOld;

    uint8_t* prow = ptr + (cd.row0 - row0_base) * (uint64_t)rowBytes;
    const uint64_t boff = cd.row0 + (uint64_t)c;
    int32_t Bi = 0;
    uint32_t wbits = 0, nbits = 0;
    for (int i = 0; i < n; ++i) {
        if ((i & 15) == 0) wbits = bases2[cd.woff + (i >> 4)];
        int r = (wbits >> (2 * (i & 15))) & 3;
        if (cd.noff >= 0) {
            if ((i & 31) == 0) nbits = nmask[cd.noff + (i >> 5)];
            if ((nbits >> (i & 31)) & 1) r = 4;
        }
        const bool row0 = (i == 0);
        const int32_t Bd = Bi + sc.del;
        long long best = (long long)NEG_INF32 * 4294967296LL;
        int32_t carryS = NEG_INF32;    // this row: final value of the last cell of the previous tile
        int32_t carryOld = NEG_INF32;  // previous row: value of the last cell of the previous tile
        for (int tile = 0; tile < n_tiles; ++tile) {
            const size_t x0 = ((size_t)tile * 1024 + (size_t)t) * Q;
            uint8_t meta[Q];
            int32_t E[Q];
            {
                const uint4* mp = reinterpret_cast<const uint4*>(tmeta + x0);
                const uint4 m0 = mp[0], m1 = mp[1];
                const uint32_t mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                for (int q = 0; q < Q; ++q) meta[q] = (uint8_t)(mw[q >> 2] >> (8 * (q & 3)));
                if (row0) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) E[q] = NEG_INF32;
                } else {
                    const int4* ep = reinterpret_cast<const int4*>(Est + x0);
#pragma unroll
                    for (int q4 = 0; q4 < Q / 4; ++q4) {
                        const int4 e = ep[q4];
                        E[4 * q4] = e.x; E[4 * q4 + 1] = e.y; E[4 * q4 + 2] = e.z; E[4 * q4 + 3] = e.w;
                    }
                }
            }
            bool anyStart = false;
#pragma unroll
            for (int q = 0; q < Q; ++q) anyStart |= (meta[q] & CELL_START) != 0;
            // previous-row value of cell x0-1 (diagonal input of the thread's first cell)
            if (lane == 63) waveOld[w] = E[Q - 1];
            __syncthreads();
            int32_t pdEdge = __shfl_up(E[Q - 1], 1);
            if (lane == 0) pdEdge = w == 0 ? carryOld : waveOld[w - 1];
            const int32_t lastOld = waveOld[15];

            int32_t loc[Q];
            int32_t run = NEG_INF32, pd = pdEdge;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool st = (meta[q] & CELL_START) != 0;
                const int32_t mmd = ((meta[q] & CELL_CODE_MASK) == r) ? mD : xD;
                int32_t cand;
                if (row0) cand = st ? mmd + sc.del : mmd;
                else if (st) cand = Bd + mmd;
                else cand = max(max(pd, Bd) + mmd, E[q] + ins);
                run = st ? cand : max(cand, run);
                loc[q] = run;
                pd = E[q];
            }
            int32_t v = run;
            int fl = anyStart ? 1 : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int32_t v2 = __shfl_up(v, off);
                const int f2 = __shfl_up(fl, off);
                if (lane >= off) {
                    if (!fl) v = max(v, v2);
                    fl |= f2;
                }
            }
            if (lane == 63) { waveV[w] = v; waveF[w] = fl; }
            __syncthreads();
            int32_t cw = carryS;
            for (int w2 = 0; w2 < w; ++w2) cw = waveF[w2] ? waveV[w2] : max(cw, waveV[w2]);
            const int32_t S = fl ? v : max(v, cw);
            int32_t Sprev = __shfl_up(S, 1);
            if (lane == 0) Sprev = cw;

            int32_t left = Sprev;
            pd = pdEdge;
            bool before = true;
            uint64_t codes = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool st = (meta[q] & CELL_START) != 0;
                if (st) before = false;
                const int32_t Ef = before ? max(loc[q], Sprev) : loc[q];
                const int32_t old = E[q];
                const int32_t mmd = ((meta[q] & CELL_CODE_MASK) == r) ? mD : xD;
                int pc;
                if (!st && Ef == left) pc = 0;
                else if (!row0 && Ef == old + ins) pc = 1;
                else if (!row0 && !st && Ef == pd + mmd) pc = 2;
                else pc = 3;
                codes |= (uint64_t)pc << (2 * q);
                if (meta[q] & CELL_END) {
                    const size_t x = x0 + q;
                    best = max(best, end_key(Ef + tend_kd[x], tend_j[x], grank, (size_t)c * T));
                }
                left = Ef;
                pd = old;
                E[q] = Ef;
            }
            {
                int4* ep = reinterpret_cast<int4*>(Est + x0);
#pragma unroll
                for (int q4 = 0; q4 < Q / 4; ++q4) ep[q4] = make_int4(E[4 * q4], E[4 * q4 + 1], E[4 * q4 + 2], E[4 * q4 + 3]);
                *reinterpret_cast<uint64_t*>(prow + (size_t)i * rowBytes + x0 / 4) = codes;
            }
            if (t == 1023) { tileS = S; tileOld = lastOld; }
            __syncthreads();
            carryS = tileS;
            carryOld = tileOld;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off));
        if (lane == 0) waveKey[w] = best;
        __syncthreads();
        long long kb = waveKey[0];
        for (int w2 = 1; w2 < 16; ++w2) kb = max(kb, waveKey[w2]);
        Bi = (int32_t)(kb >> 32);
        if (t == 0) {
            Bout[boff + i + 1] = Bi;
            argBout[boff + i + 1] = key_tmpl(kb, grank != nullptr);
        }
        __syncthreads();   // waveKey / tile carries are rewritten by the next row
    }
}

// Pointer-walking traceback (main.cpp:217-269).  One thread per chunk.
__global__ void sd_generic_trace(const ChunkDesc* __restrict__ chunks, int chunk_begin,
                                 int n_sub, const uint8_t* __restrict__ ptr, uint64_t row0_base,
                                 int rowBytes, const int32_t* __restrict__ B,
                                 const int32_t* __restrict__ argB, const int32_t* __restrict__ toff,
                                 const int32_t* __restrict__ tlen, DevRec* __restrict__ recs,
                                 int32_t* __restrict__ rec_cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sub) return;
    const int c = chunk_begin + s;
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    const uint8_t* p0 = ptr + (cd.row0 - row0_base) * (uint64_t)rowBytes;
    const int32_t* Bc = B + cd.row0 + (uint64_t)c;
    const int32_t* Ac = argB + cd.row0 + (uint64_t)c;
    DevRec* out = recs + cd.row0;
    int cnt = 0;
    int i = n - 1, j = Ac[n], k = tlen[j] - 1, open_end = i;
    while (true) {
        const int x = toff[j] + k;
        const int pc = (p0[(size_t)i * rowBytes + (x >> 2)] >> (2 * (x & 3))) & 3;
        if (pc == 0) { --k; }
        else if (pc == 1) { --i; }
        else if (pc == 2) { --i; --k; }
        else {
            DevRec r;
            r.tmpl = j;
            r.start = i;
            r.end = open_end;
            r.score = Bc[open_end + 1] - (i != 0 ? Bc[i] : 0);  // main.cpp:255 / 258-262
            out[cnt++] = r;
            if (i == 0) break;
            j = Ac[i];        // between-monomers hop, main.cpp:228-236
            i = i - 1;
            k = tlen[j] - 1;
            open_end = i;
        }
    }
    rec_cnt[c] = cnt;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
int generic_pick_q(int64_t sum_len) {
    const int qs[4] = {4, 8, 16, 32};
    for (int q : qs)
        if ((sum_len + q - 1) / q <= 1024) return q;
    return 32;   // tiled: ceil(sum_len / 32768) tiles of 1024 threads x 32 cells
}

void launch_generic_fill(int Q, int threads, int grid, hipStream_t st, const ChunkDesc* chunks,
                         int chunk_begin, const uint32_t* bases2, const uint32_t* nmask,
                         const uint8_t* tmeta, const int32_t* tend_kd, const int32_t* tend_j,
                         ScoreArgs sc, int rowBytes, uint8_t* ptr, uint64_t row0_base, int32_t* B,
                         int32_t* argB, const uint16_t* grank, int T, int n_tiles, int32_t* Estate) {
    if (n_tiles > 1) {
        hipLaunchKernelGGL(sd_generic_fill_tiled, dim3(grid), dim3(1024), 0, st, chunks, chunk_begin, bases2, nmask, tmeta,
                           tend_kd, tend_j, sc, rowBytes, ptr, row0_base, B, argB, grank, T, n_tiles, Estate);
        return;
    }
#define SD_LAUNCH(QQ)                                                                            \
    hipLaunchKernelGGL(sd_generic_fill<QQ>, dim3(grid), dim3(threads), 0, st, chunks, chunk_begin, \
                       bases2, nmask, tmeta, tend_kd, tend_j, sc, rowBytes, ptr, row0_base, B, argB, grank, T)
    switch (Q) {
        case 4: SD_LAUNCH(4); break;
        case 8: SD_LAUNCH(8); break;
        case 16: SD_LAUNCH(16); break;
        default: SD_LAUNCH(32); break;
    }
#undef SD_LAUNCH
}

void launch_generic_trace(int n_sub, hipStream_t st, const ChunkDesc* chunks, int chunk_begin,
                          const uint8_t* ptr, uint64_t row0_base, int rowBytes, const int32_t* B,
                          const int32_t* argB, const int32_t* toff, const int32_t* tlen,
                          DevRec* recs, int32_t* rec_cnt) {
    const int bs = 64;
    hipLaunchKernelGGL(sd_generic_trace, dim3((n_sub + bs - 1) / bs), dim3(bs), 0, st, chunks,
                       chunk_begin, n_sub, ptr, row0_base, rowBytes, B, argB, toff, tlen, recs,
                       rec_cnt);
}

// ---------------------------------------------------------------------------------------------
// record compaction, shared by both kernel families
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sd_scan_counts(const int32_t* __restrict__ cnt, int n,
                                                       int64_t* __restrict__ roff) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t base_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int b = 0; b < n; b += 1024) {
        const int idx = b + t;
        const int64_t v = idx < n ? cnt[idx] : 0;
        int64_t s = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t s2 = __shfl_up(s, off);
            if (lane >= off) s += s2;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        int64_t pre = base_s;
        for (int w2 = 0; w2 < w; ++w2) pre += wsum[w2];
        if (idx < n) roff[idx] = pre + s - v;
        __syncthreads();
        if (t == 1023) base_s = pre + s;
        __syncthreads();
    }
    if (t == 0) roff[n] = base_s;
}

__global__ void sd_compact(const ChunkDesc* __restrict__ chunks, const int32_t* __restrict__ cnt,
                           const int64_t* __restrict__ roff, const DevRec* __restrict__ recs,
                           DevRec* __restrict__ out, int64_t out_cap, int32_t* __restrict__ out_chunk) {
    const int c = blockIdx.x;
    const int k = cnt[c];
    if (roff[c] + k > out_cap) return;  // host re-runs the compaction with a larger buffer
    const DevRec* src = recs + chunks[c].row0;
    DevRec* dst = out + roff[c];
    for (int a = threadIdx.x; a < k; a += blockDim.x) dst[a] = src[k - 1 - a];  // reverse, main.cpp:268
    if (out_chunk)   // in-stream identities (sd_ident.hip): which chunk's bases a record's segment lies in
        for (int a = threadIdx.x; a < k; a += blockDim.x) out_chunk[roff[c] + a] = c;
}

// Offsets and compaction in ONE launch: a workgroup owns a contiguous range of chunks, publishes the record count of its
// range, adds up the counts of the ranges before it as they appear, scans its own chunks and copies their records.
// WHICH range a workgroup owns is decided by a ticket it draws when it starts (ws[2 * SC_NB], a counter that only grows;
// `ticket0` = its value before this launch), not by blockIdx: a workgroup then only ever waits for workgroups that have
// started -- on a multi-XCD part workgroups are dealt round-robin to XCDs that dispatch on their own, and this kernel
// shares the machine with the next batch's persistent fill, so "lower blockIdx runs first" is not a guarantee (rocPRIM's
// decoupled look-back orders its blocks the same way).  `ws`: words that persist between launches -- [b] the count of
// range b, [SC_NB + b] the launch (`epoch`) it belongs to -- so nothing has to be cleared.  Replaces the one-workgroup sd_scan_counts, which sat behind the
// next batch's fill on the lower-priority stream for milliseconds before its 16 us of work (round 3: 4 ms on average).
constexpr int SC_NB = 256, SC_T = 256;
__global__ __launch_bounds__(SC_T) void sd_scan_compact(const ChunkDesc* __restrict__ chunks, const int32_t* __restrict__ cnt,
                                                        int n, int64_t* __restrict__ roff, const DevRec* __restrict__ recs,
                                                        DevRec* __restrict__ out, int64_t out_cap,
                                                        int32_t* __restrict__ out_chunk, long long* __restrict__ ws,
                                                        long long epoch, long long ticket0) {
    __shared__ long long red[SC_T / 64];
    __shared__ long long run_s;
    __shared__ int b_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) b_s = (int)((long long)atomicAdd(reinterpret_cast<unsigned long long*>(ws + 2 * SC_NB), 1ull) - ticket0);
    __syncthreads();
    const int nb = (int)gridDim.x, b = b_s;
    const int per = (n + nb - 1) / nb;
    const int c0 = min(n, b * per), c1 = min(n, c0 + per);
    auto block_sum = [&](long long v) {   // valid in every thread
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        __syncthreads();
        if (lane == 0) red[w] = v;
        __syncthreads();
        long long r = 0;
        for (int x = 0; x < SC_T / 64; ++x) r += red[x];
        return r;
    };
    long long s = 0;
    for (int c = c0 + t; c < c1; c += SC_T) s += cnt[c];
    const long long mine = block_sum(s);
    if (t == 0) {
        __hip_atomic_store(ws + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + SC_NB + b, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long pre = 0;
    for (int x = t; x < b; x += SC_T) {
        while (__hip_atomic_load(ws + SC_NB + x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
        pre += __hip_atomic_load(ws + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long base = block_sum(pre);
    if (t == 0) run_s = base;
    __syncthreads();
    for (int cb = c0; cb < c1; cb += SC_T) {   // exclusive scan of the range, SC_T chunks at a time
        const int c = cb + t;
        const long long v = c < c1 ? cnt[c] : 0;
        long long sc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const long long s2 = __shfl_up(sc, off);
            if (lane >= off) sc += s2;
        }
        if (lane == 63) red[w] = sc;
        __syncthreads();
        long long p2 = run_s;
        for (int x = 0; x < w; ++x) p2 += red[x];
        if (c < c1) roff[c] = p2 + sc - v;
        __syncthreads();
        if (t == SC_T - 1) run_s = p2 + sc;
        __syncthreads();
    }
    if (b == nb - 1 && t == 0) roff[n] = base + mine;
    __threadfence_block();
    __syncthreads();
    // compaction of the range: one wave per chunk at a time
    for (int c = c0 + w; c < c1; c += SC_T / 64) {
        const int k = cnt[c];
        const long long off = roff[c];
        if (off + k > out_cap) continue;   // host re-runs the compaction with a larger buffer
        const DevRec* src = recs + chunks[c].row0;
        DevRec* dst = out + off;
        for (int a = lane; a < k; a += 64) dst[a] = src[k - 1 - a];   // reverse, main.cpp:268
        if (out_chunk)
            for (int a = lane; a < k; a += 64) out_chunk[off + a] = c;
    }
}

void launch_compact(hipStream_t st, const ChunkDesc* chunks, int n_chunks, const int32_t* cnt,
                    int64_t* roff, const DevRec* recs, DevRec* out, int64_t out_cap, bool scan, int32_t* out_chunk,
                    long long* scan_ws, long long epoch, long long* tickets) {
    if (scan && scan_ws != nullptr && tickets != nullptr && n_chunks > 0) {
        const int nb = std::max(1, std::min(SC_NB, (n_chunks + 31) / 32));
        hipLaunchKernelGGL(sd_scan_compact, dim3(nb), dim3(SC_T), 0, st, chunks, cnt, n_chunks, roff, recs, out, out_cap,
                           out_chunk, scan_ws, epoch, *tickets);
        *tickets += nb;
        return;
    }
    if (scan) hipLaunchKernelGGL(sd_scan_counts, dim3(1), dim3(1024), 0, st, cnt, n_chunks, roff);
    hipLaunchKernelGGL(sd_compact, dim3(n_chunks), dim3(64), 0, st, chunks, cnt, roff, recs, out,
                       out_cap, out_chunk);
}

}  // namespace sd
