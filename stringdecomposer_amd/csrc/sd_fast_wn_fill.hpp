// sd_fast_wn_fill.hpp -- the multi-wave wide fill kernel, shared by sd_fast_wn.hip and sd_fast_wn_fl.hip (the
// variant that skips the start-term maximum behind the first FL slots; see sd_fast_fl.hip).  Described at
// the top of sd_fast_wn.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>

#include "sd_fast.hpp"
#include "sd_fast_dev.hpp"

namespace sd {

namespace {
__device__ __forceinline__ uint32_t cvt_bf8_pair(uint32_t tb, int pair, uint32_t one_s) {
    uint32_t v;
    if (pair == 0) asm("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2" : "=v"(v) : "v"(tb), "s"(one_s));
    else asm("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(v) : "v"(tb), "s"(one_s));
    return v;
}
// integer cells: the table byte pair of a slot is int8; two SDWA adds sign-extend the bytes onto the 16-bit halves (the sums
// never start from the -32768 padding value: u is at least the row's finite start term, see the slot loop)
__device__ __forceinline__ uint32_t wn_add_b8(uint32_t u, uint32_t tb, int pair) {
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
}  // namespace

// F16 = false (sd_fast_wn_i16.hip, round 5): packed int16 cells and int8 table bytes instead of fp16 cells and bf8 bytes --
// 5.5 instead of 4.5 operations per slot (no packed three-input maximum for int16, two byte adds instead of convert + add)
// -- for scorings whose stored values leave the exact-integer range of fp16 (|S| up to 12 000 instead of 2 040) and as the
// form an engine repeats a batch in after its fp16 range guard tripped; until round 5 both cases fell to the generic
// family (50x slower per cell).  Same lane layout, checkpoints (int16 pairs) and B words; no --ed_thr compaction.
//
// COMPACT (--ed_thr with more than 128 templates, sd_fast_wn_ck.hip): a chunk is filled by as many waves as its
// KEPT templates need (blockDim.x / 64 = ceil(kept / 128), chosen per chunk class by the launcher), whose virtual
// lanes hold the kept templates in their filtered order (klist, written by sd_rank_keep) -- the reference's
// prefilter exists to cut the DP work of large monomer sets (main.cpp:128-149), and here it turns a chunk of W
// waves into a chunk of one, two, ...  The waves gather the codes of their templates into LDS when they take the
// chunk; (wave, lane) order = filtered order, so "smallest wave, then smallest virtual lane among equal ends" is
// already the reference's tie-break and the unranked reduction applies.  Same B words and checkpoint layout (the
// first waves of the W-wave layout).
//
// TILED (sd_fast_wt.hip): a template is tiled over V = ceil(L / P) consecutive virtual lanes of one plane of one wave,
// as in the narrow single-wave fills (sd_fast_fill.hpp) -- sets whose templates are longer than the widest lane (224
// slots) and too many or too long for one wave.  The in-row deletion chain then crosses lanes: the carry K (exclusive,
// template-segmented prefix maximum of the lane totals, Vmax - 1 DPP hops) is applied lazily exactly as there -- it
// joins the chain at slot 0 of the next row, it is the diagonal input of slot 0, and checkpoints / ends take
// max(value, K).
template <int P, bool RANKED, int FL = P, bool COMPACT = false, bool TILED = false, bool F16 = true>
__global__ __launch_bounds__(512, 2) void sd_fast_fill_wn(
    const ChunkDesc* __restrict__ chunks, int n_chunks, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint32_t* __restrict__ codes,
    const uint32_t* __restrict__ lane_consts, ScoreArgs sc, int W, uint32_t mb, uint32_t xb,
    int32_t* __restrict__ Bout, uint32_t* __restrict__ ckpt, int32_t* __restrict__ ckbase,
    int* __restrict__ queue, const int* __restrict__ order, const uint32_t* __restrict__ cendoff,
    const uint32_t* __restrict__ crank, const int* __restrict__ n_ptr, const uint16_t* __restrict__ klist,
    const uint8_t* __restrict__ tcodes, const int32_t* __restrict__ toff, const int32_t* __restrict__ tlen,
    int klist_stride, int Hx = 0,     // Hx (TILED): carry hops | 1-bp templates present << 8
    const uint32_t* __restrict__ lane_t = nullptr) {   // COMPACT && TILED: [chunk][W * 128] template | part << 16 of every virtual lane
    static_assert(P % 16 == 0, "the code table is streamed 16 slots at a time");
    static_assert(!(COMPACT && RANKED), "the compacted form needs no ranks");
    static_assert(F16 || (!COMPACT && FL == P), "integer cells: the plain W-wave kernels only");
    constexpr int G = P / 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // [W][G][2 halves][64 lanes][4 dwords] codes, then the exchange area
    if (n_ptr) n_chunks = *n_ptr;      // the size of a chunk class is known on the device only
    const int Wb = COMPACT ? (int)(blockDim.x >> 6) : W;   // waves of this workgroup; W stays the checkpoint stride
    const int TBL = Wb * G * 512;
    if constexpr (!COMPACT)
    for (int idx = threadIdx.x * 4; idx < TBL; idx += blockDim.x * 4)
        *reinterpret_cast<uint4*>(&lds[idx]) = *reinterpret_cast<const uint4*>(&codes[idx]);
    int32_t* xv = reinterpret_cast<int32_t*>(lds + TBL);   // [2 parities][8 waves] wave maxima
    int32_t* xa = xv + 16;                                 // [2][8] arg-max virtual lane of each wave
    int32_t* xr = xa + 16;                                 // [2][8] --ed_thr: smallest rank among each wave's maxima
    int32_t* xc = xr + 16;                                 // [1] chunk of the workgroup
    __syncthreads();

    using CO = CellOps<F16>;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    uint32_t* myc = lds + wave * (G * 512);
    const uint32_t* lc = lane_consts + (size_t)(wave * 64 + lane) * FAST_LANE_WORDS;
    const uint32_t endOffPlan = lc[FLC_ENDOFF];
    uint32_t row0adj = CO::from_i16x2(lc[FLC_ROW0]);   // (not const: the compacted tiled form derives the lane constants per chunk)
    const uint32_t ins2 = CO::splat(sc.ins);
    constexpr uint32_t NEGC = CO::NEG;
    uint32_t startMask = TILED ? lc[FLC_STARTMASK] : 0xffffffffu;
    uint32_t contMask = TILED ? lc[FLC_CONTMASK] : 0u;
    const int H = Hx & 0xff;
    // a 1-bp template ends in slot 0 (FLC_ONE): the pads behind a k = 0 cell keep their old value when the cell's falls
    // (no insertion move there, main.cpp:188-193); one wave-uniform branch per row of >= 96 slots
    const bool has_one = TILED && ((Hx >> 8) & 1);
    uint32_t oneMask = TILED ? lc[FLC_ONE] : 0u;
    // exclusive, template-segmented prefix maximum of the lane totals (both planes at once): H = Vmax - 1 hops
    auto excl_scan = [&](uint32_t a) {
        uint32_t inc = a;
        for (int h = 1; h < H; ++h) inc = CO::mx(a, bfi(contMask, lane_up(inc, 1), NEGC));
        return bfi(contMask, lane_up(inc, 1), NEGC);
    };

    for (;;) {
        if (threadIdx.x == 0) {
            const int q = atomicAdd(queue, 1);
            xc[0] = q < n_chunks ? order[q] : -1;
        }
        __syncthreads();
        const int c = __builtin_amdgcn_readfirstlane(xc[0]);
        __syncthreads();
        if (c < 0) break;
        const ChunkDesc cd = chunks[c];
        const int n = cd.n;
        ReadStream rs;
        rs.init(bases2 + cd.woff, cd.noff >= 0 ? nmask + cd.noff : nullptr, n);
        // --ed_thr: per-chunk end offsets (-inf for dropped templates) and tie-break ranks (main.cpp:141-147)
        const size_t cl = ((size_t)c * (size_t)W + (size_t)wave) * 64 + (size_t)lane;
        uint32_t endOffC = 0;
        if constexpr (COMPACT) {
            // this lane's two templates (lo plane: place 128 wave + lane of the filtered order, hi plane: + 64);
            // klist: [chunk][T], 0xffff behind the kept ones.  TILED: the placement kernel's lane table instead
            // (sd_tiled_place: template | part << 16 of every virtual lane, 0xffffffff = idle) -- the lane holds the cells
            // [part * P, part * P + P) of its template, and the lane constants of the carry follow from (part, parts).
            const int T = klist_stride;
            int tlo, thi, ulo = 0, uhi = 0;
            if constexpr (TILED) {
                const uint32_t* lt = lane_t + (size_t)c * (size_t)(W * 128) + 128 * wave + lane;
                const uint32_t elo_ = lt[0], ehi_ = lt[64];
                tlo = elo_ == 0xffffffffu ? 0xffff : (int)(elo_ & 0xffffu);
                thi = ehi_ == 0xffffffffu ? 0xffff : (int)(ehi_ & 0xffffu);
                ulo = tlo != 0xffff ? (int)(elo_ >> 16) : 0;
                uhi = thi != 0xffff ? (int)(ehi_ >> 16) : 0;
            } else {
                const uint16_t* kl = klist + (size_t)c * (size_t)T;
                const int plo = 128 * wave + lane, phi_ = plo + 64;
                tlo = plo < T ? kl[plo] : 0xffff;
                thi = phi_ < T ? kl[phi_] : 0xffff;
            }
            const int Llo = tlo != 0xffff ? tlen[tlo] : 0, Lhi = thi != 0xffff ? tlen[thi] : 0;
            const int klo = ulo * P, khi = uhi * P;   // first cell of the lane
            const uint8_t* clo = tcodes + (tlo != 0xffff ? toff[tlo] : 0) + klo;
            const uint8_t* chi = tcodes + (thi != 0xffff ? toff[thi] : 0) + khi;
            const int nlo = Llo - klo, nhi = Lhi - khi;   // cells from there on (more than P: the next lane's)
            for (int dw = 0; dw < G * 8; ++dw) {       // dword dw of the lane: slots 2 dw, 2 dw + 1 as {lo, hi, lo, hi}
                const int a = 2 * dw;
                const uint32_t b0 = a < nlo ? clo[a] : 7u, b1 = a < nhi ? chi[a] : 7u;
                const uint32_t b2 = a + 1 < nlo ? clo[a + 1] : 7u, b3 = a + 1 < nhi ? chi[a + 1] : 7u;
                myc[(dw >> 2) * 256 + lane * 4 + (dw & 3)] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
            // the end of a template sits in its last lane
            const bool endlo = Llo > 0 && nlo <= P, endhi = Lhi > 0 && nhi <= P;
            const int elo = endlo ? (Llo - 1) * sc.del : -32768, ehi = endhi ? (Lhi - 1) * sc.del : -32768;
            endOffC = ((uint32_t)elo & 0xffffu) | ((uint32_t)ehi << 16);
            if constexpr (TILED) {
                const uint32_t slo = (tlo == 0xffff || ulo == 0) ? 0xffffu : 0u, shi = (thi == 0xffff || uhi == 0) ? 0xffffu : 0u;
                startMask = slo | (shi << 16);
                contMask = ~startMask;
                oneMask = (Llo == 1 ? 0xffffu : 0u) | (Lhi == 1 ? 0xffff0000u : 0u);
                const int rlo = (tlo != 0xffff && ulo == 0) ? sc.ins + sc.del : sc.ins;
                const int rhi = (thi != 0xffff && uhi == 0) ? sc.ins + sc.del : sc.ins;
                row0adj = CO::from_i16x2(((uint32_t)rlo & 0xffffu) | ((uint32_t)rhi << 16));
            }
        }
        const uint32_t endOff = CO::from_i16x2(COMPACT ? endOffC : RANKED ? cendoff[cl] : endOffPlan);
        const uint32_t rank2 = RANKED ? crank[cl] : 0u;
        int32_t* Bc = Bout + cd.row0 + (uint64_t)c;
        uint32_t* ck = ckpt + ((uint64_t)cd.pad * (uint64_t)W + (uint64_t)wave) * (uint64_t)(P * 64) + lane;
        int32_t* ckb = ckbase + cd.pad;

        uint32_t L[P];
        uint32_t K = NEGC;   // TILED: the lane's carry
        uint32_t cg[2][8];  // 16 slots of codes per buffer: dword d = slots 2d, 2d+1 as {lo, hi, lo, hi} bytes
        int base = 0, Brel = 0, tp = 0;
        int accBV = 0;

        // one LDS address per lane; groups and halves are immediate offsets of ds_read_b128
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    using lds_u4 = __attribute__((address_space(3))) const u32x4_t;
        uint32_t cbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)myc + lane * 16;
        auto load_group = [&](int g, int buf, uint32_t& after) {
            asm volatile("" : "+v"(cbase), "+v"(after));
            lds_u4* t = (lds_u4*)(uintptr_t)cbase;
            const u32x4_t q0 = t[g * 128];
            const u32x4_t q1 = t[g * 128 + 64];
            cg[buf][0] = q0.x; cg[buf][1] = q0.y; cg[buf][2] = q0.z; cg[buf][3] = q0.w;
            cg[buf][4] = q1.x; cg[buf][5] = q1.y; cg[buf][6] = q1.z; cg[buf][7] = q1.w;
        };
        // the eight table bytes of a row symbol r: code 0..4 -> match / mismatch, 5..7 -> -inf (padding = 7)
        auto pool_of = [&](int r, uint32_t& plo, uint32_t& phi) {
            constexpr uint32_t PADS = F16 ? 0xFCFCFC00u : 0x80808000u;   // bf8 -inf / int8 -128
            uint32_t lo = xb * 0x01010101u, hi = PADS | xb;
            if (r < 4) lo = (lo & ~(0xffu << (8 * r))) | (mb << (8 * r));
            else hi = PADS | mb;
            plo = lo;
            phi = hi;
        };
        // B_{row} = max over ALL template ends of the chunk; arg = smallest (wave, virtual lane) attaining it
        auto reduce_ends = [&](uint32_t Eend, int row) {
            const uint32_t val = CO::add(Eend, endOff);
            int lo, hi;
            CO::to_int(val, lo, hi);
            const int bw = wave_max(max(lo, hi));
            unsigned long long mlo, mhi;
            int kw = 0;
            if (RANKED) {
                // among equal values the first template of the chunk's filtered order wins (smallest rank)
                const int klo = lo == bw ? (int)(rank2 & 0xffffu) : 0x7fff;
                const int khi = hi == bw ? (int)(rank2 >> 16) : 0x7fff;
                kw = -wave_max(-min(klo, khi));
                mlo = __ballot(klo == kw);
                mhi = __ballot(khi == kw);
            } else {
                mlo = __ballot(lo == bw);
                mhi = __ballot(hi == bw);
            }
            const int vw = mlo ? (__ffsll((long long)mlo) - 1) : (64 + __ffsll((long long)mhi) - 1);
            int b = bw, arg = vw;
            if (Wb > 1) {   // (one wave per chunk -- the compacted and the tiled forms can be -- needs no exchange)
                const int par = (row & 1) * 8;
                if (lane == 0) { xv[par + wave] = bw; xa[par + wave] = vw; xr[par + wave] = kw; }
                __syncthreads();
                int rk = xr[par];
                b = xv[par];
                arg = xa[par];
                for (int w2 = 1; w2 < Wb; ++w2) {
                    const int b2 = xv[par + w2], r2 = xr[par + w2];
                    if (b2 > b || (RANKED && b2 == b && r2 < rk)) { b = b2; rk = r2; arg = (w2 << 7) | xa[par + w2]; }
                }
            }
            b = __builtin_amdgcn_readfirstlane(b);
            arg = __builtin_amdgcn_readfirstlane(arg);
            Brel = b + tp * sc.ins;
            if (wave == 0) {
                const int slot = (row - 1) & 63;
                acc_put(accBV, (int)(((uint32_t)(base + Brel) << 10) | (uint32_t)arg), slot);
                if (slot == 63 || row == n) {
                    if (lane <= slot) Bc[row - slot + lane] = accBV;
                }
            }
        };

        // ---- row 0 (main.cpp:171-182)
        {
            const int r0 = rs.code(0);
            rs.advance(0);
            uint32_t plo, phi;
            pool_of(r0, plo, phi);
            uint32_t pin = 0;
            uint32_t run = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                load_group(g, 0, pin);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int q = 16 * g + s;
                    const uint32_t tb = __builtin_amdgcn_perm(phi, plo, cg[0][s >> 1]);
                    uint32_t t16;
                    if constexpr (F16) t16 = CO::add(q == 0 ? row0adj : ins2, cvt_bf8_pair(tb, s & 1, 0x3f800000u));
                    else t16 = wn_add_b8(q == 0 ? row0adj : ins2, tb, s & 1);
                    run = q == 0 ? t16 : CO::mx(run, t16);
                    L[q] = run;
                }
                pin = run;
            }
            if constexpr (TILED) {
                K = excl_scan(L[P - 1]);   // (row 0: the pads of a 1-bp lane equal its cell)
                reduce_ends(CO::mx(L[P - 1], K), 1);
            } else {
                reduce_ends(L[P - 1], 1);
            }
        }
        F16Guard<P> guard;   // run-time check of the fp16 exact-integer range (sd_fast_dev.hpp)
        if constexpr (F16) {
            guard.start(L[P - 1], sc.guard_lim);
            guard.check_low(L);
        }
        int rnext = rs.code(1);
        rs.advance(1);
        load_group(0, 0, L[P - 1]);
        for (int i = 1; i < n; ++i) {
            const int rcur = rnext;
            if ((i & (FAST_R - 1)) == 0) {
                if ((i & sc.rebase_mask) == 0) {   // FastPlan::rebase rows
                    const uint32_t d2 = CO::splat(Brel - tp * sc.ins);
                    base += Brel;
                    Brel = 0;
                    tp = 0;
                    if constexpr (F16) guard.check_high(L);
#pragma unroll
                    for (int s = 0; s < P; ++s) L[s] = CO::sub(L[s], d2);
                    if constexpr (TILED) K = bfi(startMask, NEGC, CO::sub(K, d2));
                    if constexpr (F16) guard.check_low(L);
                }
                const int q = (i / FAST_R) - 1;
                uint32_t* ckq = ck + (uint64_t)q * (uint64_t)W * (uint64_t)(P * 64);
#pragma unroll
                for (int s = 0; s < P; ++s) ckq[s * 64] = TILED ? CO::mx(L[s], K) : L[s];
                if (wave == 0 && lane == 0) ckb[q] = base + tp * sc.ins;
            }
            const uint32_t KB = CO::splat(Brel + sc.del - tp * sc.ins);
            uint32_t plo, phi;
            pool_of(rcur, plo, phi);
            if constexpr (!F16) {
                // integer cells, 5.5 ops per slot: [perm per 2 slots]; u = max(S[x-1], KB); v = u + int8 pair (2 SDWA adds);
                // c = max(v, S[x]); S'[x] = max(S'[x-1], c) -- software-pipelined over the slots
                uint32_t u_[P], v_[P], c_[P];
                uint32_t run = 0, tbw = 0;
                uint32_t KBx = KB, w0 = 0;
                if constexpr (TILED) {
                    KBx = CO::mx(K, KB);               // the carry is the diagonal input of slot 0 and joins the later floors
                    w0 = bfi(startMask, NEGC, L[0]);   // no insertion move at k = 0
                }
#pragma unroll
                for (int s = 0; s < P + 3; ++s) {
                    if (s >= 3) {
                        const int q = s - 3;
                        run = q == 0 ? c_[0] : pk_max(run, c_[q]);
                        L[q] = run;
                    }
                    if (s >= 2 && s - 2 < P) {
                        const int q = s - 2;
                        if (q == 0) c_[0] = TILED ? pk_max(pk_max(v_[0], w0), K) : v_[0];   // k == 0: start term only
                        else c_[q] = pk_max(v_[q], L[q]);
                    }
                    if (s >= 1 && s - 1 < P) {
                        const int q = s - 1;
                        if ((q & 1) == 0) tbw = __builtin_amdgcn_perm(phi, plo, cg[(q >> 4) & 1][(q & 15) >> 1]);
                        v_[q] = wn_add_b8(u_[q], tbw, q & 1);
                        if ((q & 15) == 2 && (q >> 4) + 1 < G) load_group((q >> 4) + 1, ((q >> 4) + 1) & 1, v_[q]);   // one group ahead
                    }
                    if (s < P) {
                        const int q = s;
                        u_[q] = q == 0 ? KBx : pk_max(L[q - 1], KBx);
                    }
                    asm volatile("" : "+v"(KBx), "+v"(run));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            uint32_t u_[P], v_[P], t_[P];
            uint32_t KBs = (uint32_t)__builtin_amdgcn_readfirstlane((int)KB);
            // TILED: slot 0's diagonal input is the true last slot of the lane below = the carry; KBv >= K serves the
            // later slots too (their old values are true up to the same K)
            uint32_t KBv = 0, w0 = 0;
            if constexpr (TILED) {
                KBv = CO::mx(K, KB);
                w0 = bfi(startMask, NEGC, L[0]);   // no insertion move at k = 0
            }
            uint32_t one_s = 0x3f800000u;
            uint32_t tbw = 0;
            // 4.5 ops per slot: [perm per 2 slots]; t = cvt(bf8 pair); u = max(S[x-1], KB); v = u + t;
            // S'[x] = max3(S'[x-1], v, S[x]), software-pipelined over the slots like the single-wave kernel
#pragma unroll
            for (int s = 0; s < P + 4; ++s) {
                if (s >= 4) {
                    const int q = s - 4;
                    if constexpr (TILED) L[q] = q == 0 ? CO::mx3(v_[0], w0, K) : CO::mx3(L[q - 1], v_[q], L[q]);
                    else
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
                    if (q == 0) u = TILED ? KBv : KB;
                    else if (q > FL) u = L[q - 1];
                    else if constexpr (TILED) u = CO::mx(L[q - 1], KBv);
                    else asm("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(L[q - 1]), "s"(KBs));
                    u_[q] = u;
                    if ((q & 1) == 0) tbw = __builtin_amdgcn_perm(phi, plo, cg[(q >> 4) & 1][(q & 15) >> 1]);
                    t_[q] = cvt_bf8_pair(tbw, q & 1, one_s);
                    if ((q & 15) == 2 && (q >> 4) + 1 < G) load_group((q >> 4) + 1, ((q >> 4) + 1) & 1, t_[q]);
                }
                const uint32_t pinL = L[s >= 4 ? s - 4 : 0];
                asm volatile("" : "+s"(KBs), "+s"(one_s) : "v"(pinL));
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            rnext = rs.code(i + 1);
            rs.advance(i + 1);
            load_group(0, 0, L[P - 1]);
            ++tp;
            if constexpr (TILED) {
                uint32_t a = L[P - 1];
                if (has_one) a = bfi(oneMask, L[0], a);   // (such a lane is a start lane: no carry to join)
                K = excl_scan(a);   // totals never decrease: the new carry replaces the old one
                reduce_ends(CO::mx(a, K), i + 1);
            } else {
                reduce_ends(L[P - 1], i + 1);
            }
        }
        if constexpr (F16) {
            guard.check_high(L);
            guard.finish(sc.guard_flag);
        }
        __syncthreads();   // the exchange area and xc are rewritten for the next chunk
    }
}

}  // namespace sd
