// sd_fast_wide_fill.hpp -- the wide fill kernel (one template per virtual lane), shared by sd_fast_wide.hip
// (every slot count and cell format) and sd_fast_wide_fl.hip (the variants that skip the start-term maximum
// behind the first FL slots; see sd_fast_fl.hip).  Described at the top of sd_fast_wide.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>

#include "sd_fast.hpp"
#include "sd_fast_dev.hpp"

namespace sd {

namespace {

__device__ __forceinline__ uint32_t add_b8(uint32_t u, uint32_t tb, int pair) {
    uint32_t v;
    if (pair == 0) {
        asm("v_add_u16_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "=v"(v) : "v"(u), "v"(tb));
        asm("v_add_u16_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1" : "+v"(v) : "v"(u), "v"(tb));
    } else {
        asm("v_add_u16_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2" : "=v"(v) : "v"(u), "v"(tb));
        asm("v_add_u16_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3" : "+v"(v) : "v"(u), "v"(tb));
    }
    return v;
}

// fp16 variant: the byte pair of a slot holds bf8 (E5M2) values; one gfx950 instruction turns it into
// the packed fp16 pair {lo plane, hi plane} (exact for the small integers of a score table, -inf pads)
__device__ __forceinline__ uint32_t cvt_bf8x2(uint32_t tb, int pair, uint32_t one_s) {
    uint32_t v;   // one_s: the bits of 1.0f in an SGPR (the scale operand; also what pins the slot skew)
    if (pair == 0) asm("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2" : "=v"(v) : "v"(tb), "s"(one_s));
    else asm("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(v) : "v"(tb), "s"(one_s));
    return v;
}

}  // namespace

template <int P, bool RANKED, bool F16, int FL = P>
__global__ __launch_bounds__(512, 2) void sd_fast_fill_wide(
    const ChunkDesc* __restrict__ chunks, int n_chunks, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ lane_consts, ScoreArgs sc, int32_t* __restrict__ Bout,
    uint32_t* __restrict__ ckpt, int32_t* __restrict__ ckbase, int* __restrict__ queue,
    const int* __restrict__ order, const uint32_t* __restrict__ cendoff,
    const uint32_t* __restrict__ crank) {
    static_assert(P % 16 == 0, "wide variant streams the table 16 slots at a time");
    constexpr int G = P / 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // [5][G][2 halves][64][4]
    constexpr int TBL = 5 * G * 512;
    for (int idx = threadIdx.x * 4; idx < TBL; idx += blockDim.x * 4)
        *reinterpret_cast<uint4*>(&lds[idx]) = *reinterpret_cast<const uint4*>(&table[idx]);
    __syncthreads();

    using CO = CellOps<F16>;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int lane = threadIdx.x & 63;
    (void)wave; (void)nw;
    // table value of a slot added to u: two SDWA byte adds (int8 table) or convert + packed add (bf8 table)
    auto add_tbl = [&](uint32_t u, uint32_t tb, int pair) {
        if constexpr (F16) return CO::add(u, cvt_bf8x2(tb, pair, 0x3f800000u));
        else return add_b8(u, tb, pair);
    };
    FairShare fair;   // issue fairness among the waves of a SIMD at the end of the launch (sd_fast_dev.hpp)
    fair.init(reinterpret_cast<int*>(lds + TBL), wave, nw, lane, queue, n_chunks);
    ChunkSched sched;
    sched.init(queue, order, n_chunks);
    for (int c = sched.next(); c >= 0; c = sched.next()) {
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    ReadStream rs;
    rs.init(bases2 + cd.woff, cd.noff >= 0 ? nmask + cd.noff : nullptr, n);

    const uint32_t* lc = lane_consts + lane * FAST_LANE_WORDS;
    const uint32_t endOffPlan = lc[FLC_ENDOFF];
    // --ed_thr: per-chunk end offsets (-inf for dropped templates) and tie-break ranks
    const uint32_t endOff = CO::from_i16x2(RANKED ? cendoff[(size_t)c * 64 + lane] : endOffPlan);
    const uint32_t rank2 = RANKED ? crank[(size_t)c * 64 + lane] : 0u;
    const uint32_t row0adj = CO::from_i16x2(lc[FLC_ROW0]);
    const uint32_t ins2 = CO::splat(sc.ins);

    int32_t* Bc = Bout + cd.row0 + (uint64_t)c;
    uint32_t* ck = ckpt + (uint64_t)cd.pad * (uint64_t)(P * 64) + lane;
    int32_t* ckb = ckbase + cd.pad;

    uint32_t L[P];
    uint32_t tbg[2][8];  // 16 slots per buffer: dword d holds slots 2d, 2d+1 as {lo,hi,lo,hi} bytes
    int base = 0, Brel = 0, tp = 0;
    int accBV = 0;

    // one address per row (table row of the read symbol + this lane's 16 bytes); the groups and halves are
    // immediate offsets of ds_read_b128.  `after` pins the loads behind the value it names.
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    using lds_u4 = __attribute__((address_space(3))) const u32x4_t;
    const uint32_t lds_lane = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)lds + lane * 16;
    uint32_t rowoff = 0;   // LDS byte address
    auto set_row = [&](int r) { rowoff = (uint32_t)(r * (G * 2048)) + lds_lane; };
    auto load_group = [&](int g, int buf, uint32_t& after) {
        asm volatile("" : "+v"(rowoff), "+v"(after));
        lds_u4* t = (lds_u4*)(uintptr_t)rowoff;
        const u32x4_t q0 = t[g * 128];
        const u32x4_t q1 = t[g * 128 + 64];
        tbg[buf][0] = q0.x; tbg[buf][1] = q0.y; tbg[buf][2] = q0.z; tbg[buf][3] = q0.w;
        tbg[buf][4] = q1.x; tbg[buf][5] = q1.y; tbg[buf][6] = q1.z; tbg[buf][7] = q1.w;
    };
    auto reduce_ends = [&](uint32_t Eend, int row) {
        const uint32_t val = CO::add(Eend, endOff);
        int lo, hi;
        CO::to_int(val, lo, hi);
        const int b = wave_max(max(lo, hi));
        unsigned long long mlo, mhi;
        if (RANKED) {
            mlo = __ballot(lo == b);
            mhi = __ballot(hi == b);
            if (__popcll(mlo) + __popcll(mhi) > 1) {   // wave-uniform, the rarer case: several ends tie
                // ties go to the first template of the chunk's filtered order (main.cpp:141-147)
                const int klo = lo == b ? (int)(rank2 & 0xffffu) : 0x7fff;
                const int khi = hi == b ? (int)(rank2 >> 16) : 0x7fff;
                const int kmin = -wave_max(-min(klo, khi));
                mlo = __ballot(klo == kmin);
                mhi = __ballot(khi == kmin);
            }
        } else {
            mlo = __ballot(lo == b);
            mhi = __ballot(hi == b);
        }
        const int v = mlo ? (__ffsll((long long)mlo) - 1) : (64 + __ffsll((long long)mhi) - 1);
        Brel = b + tp * sc.ins;
        const int slot = (row - 1) & 63;
        acc_put(accBV, (int)(((uint32_t)(base + Brel) << 7) | (uint32_t)v), slot);
        if (slot == 63 || row == n) {
            if (lane <= slot) Bc[row - slot + lane] = accBV;
        }
    };

    // ---- row 0 (main.cpp:171-182)
    {
        const int r0 = rs.code(0);
        rs.advance(0);
        uint32_t pin = 0;
        uint32_t run = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g == 0) set_row(r0);
            load_group(g, 0, pin);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int q = 16 * g + s;
                const uint32_t t16 = add_tbl(q == 0 ? row0adj : ins2, tbg[0][s >> 1], s & 1);
                run = q == 0 ? t16 : CO::mx(run, t16);
                L[q] = run;
            }
            pin = run;
        }
        reduce_ends(L[P - 1], 1);
    }
    F16Guard<P> guard;   // run-time check of the fp16 exact-integer range (sd_fast_dev.hpp)
    if constexpr (F16) {
        guard.start(L[P - 1], sc.guard_lim);
        guard.check_low(L);
    }
    int rnext = rs.code(1);  // read symbol of the next row; its group 0 is prefetched into tbg[0]
    rs.advance(1);
    set_row(rnext);
    load_group(0, 0, L[P - 1]);
    for (int i = 1; i < n; ++i) {
        if ((i & (FAST_R - 1)) == 0) {
            fair.update(n - i);
            if ((i & sc.rebase_mask) == 0) {   // FastPlan::rebase rows
                const uint32_t d2 = CO::splat(Brel - tp * sc.ins);
                base += Brel;
                Brel = 0;
                tp = 0;
                if constexpr (F16) guard.check_high(L);
#pragma unroll
                for (int s = 0; s < P; ++s) L[s] = CO::sub(L[s], d2);
                if constexpr (F16) guard.check_low(L);
            }
            const int q = (i / FAST_R) - 1;
#pragma unroll
            for (int s = 0; s < P; ++s) ck[(uint64_t)q * (P * 64) + s * 64] = L[s];
            if (lane == 0) ckb[q] = base + tp * sc.ins;
        }
        uint32_t KB = CO::splat(Brel + sc.del - tp * sc.ins);  // kept in a VGPR: see the per-step pin below
        uint32_t u_[P], v_[P], c_[P];
        uint32_t run = 0;
        if constexpr (F16) {
            // 4 ops per slot: u = max(S[x-1], KB); t = cvt(bf8 pair); v = u + t; S'[x] = max3(S'[x-1], v, S[x])
            uint32_t t_[P];
            uint32_t KBs = (uint32_t)__builtin_amdgcn_readfirstlane((int)KB);
            uint32_t one_s = 0x3f800000u;
#pragma unroll
            for (int s = 0; s < P + 4; ++s) {
                if (s >= 4) {
                    const int q = s - 4;
                    L[q] = q == 0 ? v_[0] : CO::mx3(L[q - 1], v_[q], L[q]);  // k == 0: start term only
                }
                if (s >= 2 && s - 2 < P) {
                    const int q = s - 2;
                    v_[q] = CO::add(u_[q], t_[q]);
                }
                if (s < P) {
                    const int q = s;
                    uint32_t u;
                    // behind slot FL the start term is dominated (FastPlan::floor_slots, see sd_fast_fl.hip)
                    if (q == 0) u = KB;
                    else if (q > FL) u = L[q - 1];
                    else asm("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(L[q - 1]), "s"(KBs));
                    u_[q] = u;
                    t_[q] = cvt_bf8x2(tbg[(q >> 4) & 1][(q & 15) >> 1], q & 1, one_s);
                    if ((q & 15) == 2 && (q >> 4) + 1 < G) load_group((q >> 4) + 1, ((q >> 4) + 1) & 1, t_[q]);
                }
                // pin the skew: this step's scalar operands are "redefined" here, so the compiler cannot batch
                // the u / table conversions of later slots first (that would need 2P registers).  Scalar pins:
                // an inline asm that defines a VGPR costs a wait state (s_nop) per step on gfx950.
                // The chain value of this step is only READ by the pin (it must exist by now).
                const uint32_t pinL = L[s >= 4 ? s - 4 : 0];
                asm volatile("" : "+s"(KBs), "+s"(one_s) : "v"(pinL));
                __builtin_amdgcn_sched_barrier(0);
            }
            (void)c_; (void)run;
        } else {
#pragma unroll
        for (int s = 0; s < P + 3; ++s) {
            if (s >= 3) {
                const int q = s - 3;
                run = q == 0 ? c_[0] : pk_max(run, c_[q]);
                L[q] = run;
            }
            if (s >= 2 && s - 2 < P) {
                const int q = s - 2;
                c_[q] = q == 0 ? v_[0] : pk_max(v_[q], L[q]);  // k == 0: start term only
            }
            if (s >= 1 && s - 1 < P) {
                const int q = s - 1;
                v_[q] = add_b8(u_[q], tbg[(q >> 4) & 1][(q & 15) >> 1], q & 1);
            }
            if (s < P) {
                const int q = s;
                // every virtual lane starts a template: slot 0 has no diagonal / chain input
                u_[q] = q == 0 ? KB : pk_max(L[q - 1], KB);
                // stream the next 16 slots of the table one group ahead of their first use
                if ((q & 15) == 2 && (q >> 4) + 1 < G) load_group((q >> 4) + 1, ((q >> 4) + 1) & 1, u_[q]);
            }
            // pin the skew: the next step's inputs (KB) become available only after this step's
            // chain update, so the compiler cannot batch all u/v first (that needs 2P registers)
            asm volatile("" : "+v"(KB), "+v"(run));
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        rnext = rs.code(i + 1);
        rs.advance(i + 1);
        set_row(rnext);
        load_group(0, 0, L[P - 1]);
        ++tp;
        reduce_ends(L[P - 1], i + 1);
    }
    if constexpr (F16) {
        guard.check_high(L);
        guard.finish(sc.guard_flag);
    }
    }  // chunk queue
    fair.leave();
}


}  // namespace sd
