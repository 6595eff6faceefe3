// sd_fast_dev.hpp -- device helpers shared by the fast-family kernels (sd_fast.hip, sd_fast_wide.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "sd_device.hpp"

#ifndef SD_USE_DPP
#define SD_USE_DPP 1
#endif
#ifndef SD_ACC_WRITELANE
#define SD_ACC_WRITELANE 1
#endif

namespace sd {

namespace {

constexpr uint32_t NEG2 = 0x80008000u;  // packed {-32768, -32768}
constexpr int NEG16 = -32768;

typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
__device__ __forceinline__ uint32_t pack2(int v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }

// value of lane (l - d); lanes l < d get 0 (DPP, bound_ctrl) or their own value (shuffle form): callers
// mask those lanes.  bound_ctrl spares the copy a tied `old` operand would need.
__device__ __forceinline__ uint32_t lane_up(uint32_t x, int d) {
#if SD_USE_DPP
    if (d == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
#endif
    return (uint32_t)__shfl_up((int)x, d);
}

// max over the wave, valid in every lane (shuffle form) / in lane 63 (DPP form) -> broadcast
__device__ __forceinline__ int wave_max(int v) {
#if SD_USE_DPP
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
#else
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// value of lane l-1; lane 0 gets INT_MIN, the identity of max: a following max(., y) folds with the move into ONE
// v_max_i32_dpp (no constant to materialise, no separate v_mov_b32_dpp).  Only for values that feed a max directly.
__device__ __forceinline__ int lane_up_min(int x) {
    return __builtin_amdgcn_update_dpp(INT_MIN, x, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

// value of lane l-1; lane 0 gets -inf
__device__ __forceinline__ int lane_up_neg(int x) {
    return __builtin_amdgcn_update_dpp(NEG_INF32, x, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

// inclusive prefix maximum over the 64 lanes (DPP row shifts + row broadcasts).  `old` = INT_MIN (the identity
// of max) lets the compiler fold every step
// into one v_max_i32_dpp.
__device__ __forceinline__ int wave_prefix_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
}

// Cell arithmetic of the fill kernels.  Three cell formats (CF):
//   0  packed int16 (saturating adds);
//   1  packed fp16 holding exact integers.  gfx950 has a packed three-input maximum only for fp16 (v_pk_maximum3_f16),
//      which folds the last two maxima of the cell update:  u = max(S[x-1], KB);  v = u + tbl;  S_new[x] = max3(S_new[x-1], v, S[x]);
//   2  (round 6) packed BIASED UNSIGNED 16-bit integers, U = S + U16_BIAS, every word inside 0x0000..0x7BFF: on those bit
//      patterns the fp16 maxima ARE unsigned integer maxima and v_cmp_eq_f16 is pattern equality (tools/ubench_u16max.hip,
//      exhaustive on the device; the packed traceback has used this since round 4), so the three-input maximum stays, the
//      add is a plain 32-bit add of a signed pair (hi * 65536 + lo: no carry crosses the halves while both sums stay in
//      0..65535) -- v_add_u32 issues in 2.3 cycles where v_pk_add_f16 takes 4.1 -- and the exact range is +-15 k instead of
//      fp16's +-2 k.  0 is "-inf": the identity of every maximum, what an AND with a lane mask leaves, what idle planes hold.
constexpr int CF_I16 = 0, CF_F16 = 1, CF_U16 = 2;
constexpr int U16_BIAS = 0x3E00;                  // centre of 0..0x7BFF
constexpr uint32_t U16_BIAS2 = 0x3E003E00u;
// a signed pair as ONE 32-bit addend: adding it to a packed pair of fields adds lo to the low and hi to the high field
__device__ __forceinline__ uint32_t spair(int lo, int hi) { return (uint32_t)(hi * 65536 + lo); }

template <int CF>
struct CellOps {
    static constexpr bool F16 = CF == CF_F16, U16 = CF == CF_U16;
    static constexpr uint32_t NEG = F16 ? 0xFC00FC00u : U16 ? 0u : NEG2;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t mx(uint32_t a, uint32_t b) {
        if constexpr (F16 || U16) {
            return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_bit_cast(h2, a),
                                                                              __builtin_bit_cast(h2, b)));
        } else {
            return pk_max(a, b);
        }
    }
    static __device__ __forceinline__ uint32_t mx3(uint32_t a, uint32_t b, uint32_t c) {
        return __builtin_bit_cast(
            uint32_t, __builtin_elementwise_maximum(
                          __builtin_elementwise_maximum(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b)),
                          __builtin_bit_cast(h2, c)));
    }
    // cell + offset (U16: the offset is a signed pair, spair / splat / from_i16x2)
    static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) {
        if constexpr (F16) {
            return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, a) + __builtin_bit_cast(h2, b));
        } else if constexpr (U16) {
            return a + b;
        } else {
            return pk_adds(a, b);
        }
    }
    static __device__ __forceinline__ uint32_t sub(uint32_t a, uint32_t b) {
        if constexpr (F16) return add(a, b ^ 0x80008000u);
        else if constexpr (U16) return a - b;
        else return pk_subs(a, b);
    }
    // {x, x} as an OFFSET (U16: a signed pair; a value needs U16_BIAS2 on top)
    static __device__ __forceinline__ uint32_t splat(int x) {
        if constexpr (F16) {
            const float f = (float)x;
            return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(f, f));
        } else if constexpr (U16) {
            return spair(x, x);
        } else {
            return pack2(x);
        }
    }
    // host-built packed int16 constant -> cell format (anything <= -30000 is "-inf"; U16: an offset, "-inf" -> `ninf`)
    static __device__ __forceinline__ uint32_t from_i16x2(uint32_t w, int ninf = 0) {
        if constexpr (F16) {
            const int lo = (int)(short)(w & 0xffffu), hi = (int)w >> 16;
            const float fl = lo <= -30000 ? -__builtin_inff() : (float)lo;
            const float fh = hi <= -30000 ? -__builtin_inff() : (float)hi;
            return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(fl, fh));
        } else if constexpr (U16) {
            const int lo = (int)(short)(w & 0xffffu), hi = (int)w >> 16;
            return spair(lo <= -30000 ? ninf : lo, hi <= -30000 ? ninf : hi);
        } else {
            return w;
        }
    }
    // the two cells of a word as (saturated) integers
    static __device__ __forceinline__ void to_int(uint32_t w, int& lo, int& hi) {
        if constexpr (F16) {
            uint32_t a, b;
            asm("v_cvt_i16_f16_e32 %0, %1" : "=v"(a) : "v"(w));
            asm("v_cvt_i16_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(b) : "v"(w));
            lo = (int)(short)(a & 0xffffu);
            hi = (int)(short)(b & 0xffffu);
        } else if constexpr (U16) {
            lo = (int)(w & 0xffffu) - U16_BIAS;
            hi = (int)(w >> 16) - U16_BIAS;
        } else {
            lo = (int)(short)(w & 0xffffu);
            hi = (int)w >> 16;
        }
    }
};

// Run-time guard of the fp16 cell formats.  The fills keep integers in packed fp16, exact while |value| < 2048;
// fast_plan_build proves a bound per (template set, scoring) -- and that proof was wrong once (round 1: fuzz seed
// 906).  So the kernels check the assumption itself.  Between two rebases every stored cell with an insertion move
// (all but the k = 0 cells, which follow the start term) only grows from row to row -- the insertion candidate of
// S'[x] is S[x] -- so a period's largest values are those of its last row and its smallest those of its first:
//   * after row 0 and after every rebase shift: no finite cell below -lim,
//   * before every rebase shift and after the last row: no cell above +lim.
// About P + 6 packed ops per 128 rows.  Idle planes (no template) hold -inf in every slot and are exempt.
template <int P>
struct F16Guard {
    uint32_t pos2 = 0, neg2 = 0;       // packed {+lim, +lim} / {-lim, -lim}: SGPRs (readfirstlane), not hoisted VGPR constants
    bool bad = false;                  // wave-uniform, sticky over the chunk
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t mn(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_minimum(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b)));
    }
    __device__ __forceinline__ void start(uint32_t, int lim_) {
        bad = false;
        pos2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)CellOps<true>::splat(lim_));
        neg2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)CellOps<true>::splat(-lim_));
    }
    // The per-lane lower limit {-lim, or -inf on a plane without a template} is rebuilt from the last slot at every
    // check (a plane is idle exactly when that slot is -inf, at any row) instead of living in a register: the fill's
    // 96 VGPRs are what lets two traceback waves share a SIMD with four fill waves in the overlapped stream mode.
    __device__ __forceinline__ void check_low(const uint32_t (&L)[P]) {
        uint32_t m = L[0];
#pragma unroll
        for (int s = 1; s < P; ++s) m = mn(m, L[s]);
        const uint32_t last = L[P - 1];
        const uint32_t nl = neg2;
        const uint32_t lo = (last & 0xffffu) == 0xFC00u ? 0xFC00u : (nl & 0xffffu);
        const uint32_t hi = (last >> 16) == 0xFC00u ? 0xFC000000u : (nl & 0xffff0000u);
        const uint32_t t = CellOps<true>::mx(m, lo | hi);
        bad = bad || __ballot(t != m) != 0ull;
    }
    __device__ __forceinline__ void check_high(const uint32_t (&L)[P]) {
        uint32_t m = L[0];
#pragma unroll
        for (int s = 1; s + 1 < P; s += 2) m = CellOps<true>::mx3(m, L[s], L[s + 1]);
        if ((P & 1) == 0) m = CellOps<true>::mx(m, L[P - 1]);
        const uint32_t t = mn(m, pos2);
        bad = bad || __ballot(t != m) != 0ull;
    }
    __device__ __forceinline__ void finish(int* flag) {
        if (bad && flag && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
    }
};

// The same guard for the biased-u16 cells (CellOps<CF_U16>): every real cell inside [BIAS - lim, BIAS + lim].  The plan
// leaves room below and above that window for every intermediate sum (FastPlan::u16_lim), so while the guard holds no sum
// ever left 0..0xFFFF -- no carry crossed the halves of a word -- and no maximum saw a pattern beyond 0x7BFF.  Idle planes
// hold 0 in every slot and are exempt (`pm`: the lane's plane mask).
template <int P>
struct U16Guard {
    uint32_t hi2 = 0, lo2 = 0;         // packed {BIAS + lim} / {BIAS - lim}: SGPRs
    bool bad = false;
    typedef unsigned short u2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t mn(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u2, a), __builtin_bit_cast(u2, b)));
    }
    static __device__ __forceinline__ uint32_t mxu(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u2, a), __builtin_bit_cast(u2, b)));
    }
    __device__ __forceinline__ void start(int lim_) {
        bad = false;
        const int l = lim_ < U16_BIAS ? lim_ : U16_BIAS;
        hi2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pack2(U16_BIAS + l));
        lo2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pack2(U16_BIAS - l));
    }
    __device__ __forceinline__ void check_low(const uint32_t (&L)[P], uint32_t pm) {
        uint32_t m = L[0];
#pragma unroll
        for (int s = 1; s < P; ++s) m = mn(m, L[s]);
        const uint32_t t = mxu(m, lo2 & pm);
        bad = bad || __ballot(t != m) != 0ull;
    }
    __device__ __forceinline__ void check_high(const uint32_t (&L)[P]) {
        uint32_t m = L[0];
#pragma unroll
        for (int s = 1; s < P; ++s) m = mxu(m, L[s]);
        const uint32_t t = mn(m, hi2);
        bad = bad || __ballot(t != m) != 0ull;
    }
    __device__ __forceinline__ void finish(int* flag) {
        if (bad && flag && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
    }
};

// acc[lane == slot] = value (wave-uniform value and slot): one v_mov + one v_cndmask with a
// scalar one-hot mask instead of v_mov + v_cmp + v_cndmask
__device__ __forceinline__ void acc_put(int& acc, int value, int slot) {
#if SD_ACC_WRITELANE
    // scalar value into the lane `slot`; the lane select goes through m0 (two SGPR operands would exceed the one
    // constant-bus read a gfx9 VALU instruction may make; SGPR + m0 is the form v_writelane allows)
    // (m0 is a reserved register the compiler never allocates; nothing else in these kernels uses it -- gfx9 LDS
    // instructions do not -- so the clobber only documents the write)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(acc) : "s"(value), "s"(slot) : "m0");
#pragma clang diagnostic pop
#else
    const unsigned long long m = 1ull << slot;
    int v = value;
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc) : "v"(v), "s"(m));
#endif
}

// Persistent waves: every wave pulls the next chunk from an atomic queue when it is done with its
// own, so the tail of a launch is balanced per SIMD instead of per workgroup (C2 fill 18.3 ->
// 17.7 ms).  `order` lists the chunks longest first.  (A static strided assignment was measured
// slower: 19.3 ms -- SIMDs do not run at identical speed.)
// Issue fairness among the waves of a SIMD.  The SIMD issues from its oldest ready wave first, so of four waves that
// fill equal chunks side by side the oldest runs ahead and the youngest is left with a good part of its chunk when the
// others are done -- alone on the SIMD it is latency bound (13 cycles per instruction against 4.1 shared), and every
// launch ends with that tail (one chunk time x ~0.25).  Every 32 rows a wave publishes the rows it still
// has to fill (one LDS word per wave of the workgroup) and takes the issue priority (s_setprio) of its rank among the
// workgroup's waves on the same SIMD: most rows left = highest priority.
struct FairShare {
    int* prog;
    int wave, lane;
    unsigned long long mates;   // lanes m < nw whose wave m runs on this wave's SIMD (this wave excluded)
    const int* queue;
    int n_chunks;
    __device__ __forceinline__ void init(int* area, int wave_, int nw, int lane_, const int* queue_, int n_chunks_) {
        prog = area; wave = wave_; lane = lane_; queue = queue_; n_chunks = n_chunks_;
        int* simd = area + 16;
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(hw));
        if (lane == 0) { simd[wave] = (int)hw; prog[wave] = 0x7fffffff; }
        __syncthreads();
        const int sm = lane < nw ? simd[lane] : -1;
        mates = __ballot(sm == (int)hw && lane != wave);
    }
    bool drained = false;   // the chunk queue was empty at the last update: the launch is in its last round
    __device__ __forceinline__ void update(int remaining) {
        if (lane == 0) __hip_atomic_store(prog + wave, remaining, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int v = __hip_atomic_load(prog + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // only while no wave can pull another chunk (the end of the launch): before that a wave that is done simply
        // takes the next chunk and priorities would only serialise the SIMD's waves
        const int head = __builtin_amdgcn_readfirstlane(__hip_atomic_load(queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        drained = head >= n_chunks;
        if (head < n_chunks) { __builtin_amdgcn_s_setprio(0); return; }
        const int rank = __popcll(__ballot(v < remaining) & mates);   // mates with fewer rows left
        if (rank == 0) __builtin_amdgcn_s_setprio(0);
        else if (rank == 1) __builtin_amdgcn_s_setprio(1);
        else if (rank == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
    __device__ __forceinline__ void leave() {
        if (lane == 0) __hip_atomic_store(prog + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

struct ChunkSched {
    int* queue;
    const int* order;
    int n_chunks;
    __device__ __forceinline__ void init(int* q, const int* o, int n) {
        queue = q; order = o; n_chunks = n;
    }
    int pos = 0;   // queue position of the chunk last handed out
    // wave-uniform next chunk index, -1 when the queue is empty
    __device__ __forceinline__ int next() {
        int q = 0;
        if ((threadIdx.x & 63) == 0) q = atomicAdd(queue, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        pos = q;
        return q < n_chunks ? order[q] : -1;
    }
    // true when no wave will find another chunk after this one: the launch's last round, in which the SIMDs empty out
    __device__ __forceinline__ bool last_round() const {
        return pos + (int)(gridDim.x * (blockDim.x >> 6)) >= n_chunks;
    }
};

struct ReadCursor {
    const uint32_t* w;   // 2-bit words of the chunk
    const uint32_t* nm;  // N mask words or nullptr
    __device__ __forceinline__ int code(int i) const {
        int r = (w[i >> 4] >> (2 * (i & 15))) & 3;
        if (nm && ((nm[i >> 5] >> (i & 31)) & 1)) r = 4;
        return r;
    }
};

// Sequential reader: one scalar load per 16 rows, issued 16 rows ahead of its first use.
struct ReadStream {
    const uint32_t* w;
    const uint32_t* nmp;  // N mask words; without a mask it aliases w so that every load is unconditional
    bool has_n;           // (conditional loads would put the wave-uniform words into VGPRs)
    int last;            // n - 1
    uint32_t cur, nxt;   // words holding rows [16k, 16k+16) and the following 16
    uint32_t ncur;
    int adv_at;          // next row after which the words move on (one scalar compare per row instead of two tests)
    __device__ __forceinline__ void init(const uint32_t* w_, const uint32_t* nm_, int n) {
        w = w_; has_n = nm_ != nullptr; nmp = has_n ? nm_ : w_; last = n - 1;
        cur = w[0];
        nxt = w[last >= 16 ? 1 : 0];
        ncur = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmp[0]);
        adv_at = 15 < last ? 15 : INT_MAX;
    }
    // code of row i; rows must be requested in non-decreasing order (clamped to the last row)
    __device__ __forceinline__ int code(int i) {
        i = i < last ? i : last;
        int r = (cur >> (2 * (i & 15))) & 3;
        r = (has_n && ((ncur >> (i & 31)) & 1)) ? 4 : r;  // scalar select, no branch
        return r;
    }
    // call after consuming row i
    __device__ __forceinline__ void advance(int i) {
        if (i == adv_at) {   // (i & 15) == 15 && i < last
            adv_at = i + 16 < last ? i + 16 : INT_MAX;
            cur = nxt;
            const int k = (i >> 4) + 2;
            nxt = w[(k << 4) <= last ? k : (last >> 4)];
            if ((i & 31) == 31) ncur = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmp[(i >> 5) + 1]);
        }
    }
};

}  // namespace

}  // namespace sd
