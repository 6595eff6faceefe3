// sd_fast_fill.hpp -- the fill kernel of the fast family (narrow layout), shared by the translation units that
// instantiate it: sd_fast.hip (every slot count P, cell format, --ed_thr variant) and sd_fast_fl.hip (the
// variants that skip the start-term maximum behind the first FL slots of a lane).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "sd_fast.hpp"
#include "sd_fast_dev.hpp"

namespace sd {

// ---------------------------------------------------------------------------------------------
// fill
// ---------------------------------------------------------------------------------------------
// Stored state of a wave after row i: S[x] = E[i][x] - base - tp*ins  (tp = rows since the last
// rebase), so that the insertion move costs nothing:
//   S_new[x] = max( max(S[x-1], B_i + del - tp*ins) + (mm - del - ins),  S[x],  S_new[x-1] )
// i.e. 4 packed ops per cell pair: u = max(pd, KB); v = u + tbl; cand = max(v, old); run = max(run, cand).
// waves (= chunks) per workgroup: 16 -- one workgroup per CU, so that FairShare sees all four waves of a SIMD -- when
// the launch fills the machine, 8 -- two workgroups per CU, 2 waves per SIMD each -- for smaller launches, which then
// spread over twice as many CUs (fill_block_waves)
#define SD_FILL_NW_MAX 16
//
// F16 variant: the same recurrence on packed fp16 (every value is an integer of magnitude < 2048,
// hence exact; -inf is the padding / "no predecessor" value).  gfx950 has v_pk_maximum3_f16, which
// folds the last two maxima:  u = max(S[x-1], KB);  v = u + tbl;  S_new[x] = max3(S_new[x-1], v, S[x])
// -- 3 packed ops per cell pair instead of 4.  fast_plan_build() enables it when the score range
// fits (FastPlan::f16); the checkpoints then hold fp16 pairs (the traceback converts them).
// U16 variant (CF = 2, round 6): the same three ops on biased unsigned 16-bit integers (CellOps<CF_U16>): the fp16 maxima
// are exact unsigned maxima on the patterns 0..0x7BFF, the add is a plain v_add_u32 of a signed pair.  "-inf" is 0, and
// nothing may be ADDED to it (a negative addend would borrow from the other half of the word), so:
//   * table values of pad slots are min(tmin, 0), tmin = the smallest real table value: a pad's candidate
//     max(S[x-1], KB) + pad then never exceeds what the chain already carries (S'[x-1] >= S'[0] >= KB + tmin; S'[x-1] >= S[x-1]),
//     so the pad still only forwards the template's last cell;
//   * idle planes (no template) hold 0 in every slot: their table values and end offsets are 0 and the row's start term
//     is ANDed with the lane's plane mask before it enters the slots (one v_and_b32 per row);
//   * a rebase shifts the real planes only (per-half add of the masked shift).
// ONE: the set has 1-bp templates (FLC_ONE lanes end at slot 0); instantiated for the full-floor kernels of
// sd_fast.hip only -- as a run-time branch in every kernel it cost the C2 fill 3 % (12.3 against 11.9 ms, same box)
// FLS > 0 (round 6, u16 cells): the floor level of a template set is the maximum over the five read symbols, and single symbols
// are often far below it (C2's set: A 16, C 15, G 10, T 6), so a ROW takes the floor in the first FL, FL - FLS, FL - 2 FLS or
// FL - 3 FLS slots by its read symbol (Hx bits 22..31: two bits per symbol, 0 = FL).  Three copies of the slot loop behind a scalar
// branch were built first and lost 13 %: the register allocator gives the three loops different registers and joins them
// with 27 moves per row.  Instead the floors are applied IN PLACE before the one slot loop, L[q-1] = max(L[q-1], KB), two
// scalar branches skipping the upper groups.  That also raises the "keep" operand of slot q-1 to KB, which changes nothing
// while every table value is >= 0: S'[q-1] >= S'[0] >= KB + tbl[0] >= KB anyway (the launchers check tmin >= 0; scorings
// whose mismatch costs more than a deletion plus an insertion take the one-level kernels).
template <int P, bool RANKED, int CF, int FL = P, bool ONE = false, int FLS = 0>
__global__ __launch_bounds__(SD_FILL_NW_MAX * 64, 4) void sd_fast_fill(
    const ChunkDesc* __restrict__ chunks, int n_chunks, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ lane_consts, ScoreArgs sc, int Hx, int32_t* __restrict__ Bout,
    int32_t* __restrict__ argV, uint32_t* __restrict__ ckpt, int32_t* __restrict__ ckbase,
    int* __restrict__ queue, const int* __restrict__ order, const uint32_t* __restrict__ cendoff,
    const uint32_t* __restrict__ crank) {
    constexpr int P4 = (P + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // [5][P4/4][64][4]
    constexpr int TBL = 5 * P4 * 64;
    constexpr bool F16 = CF == CF_F16, U16 = CF == CF_U16;
    using CO = CellOps<CF>;
    constexpr uint32_t NEGC = CO::NEG;
    for (int idx = threadIdx.x * 4; idx < TBL; idx += blockDim.x * 4) {
        uint4 q = *reinterpret_cast<const uint4*>(&table[idx]);
        if constexpr (F16) {
            q.x = CO::from_i16x2(q.x); q.y = CO::from_i16x2(q.y);
            q.z = CO::from_i16x2(q.z); q.w = CO::from_i16x2(q.w);
        }
        if constexpr (U16) {
            // signed pairs; pads: min(tmin, 0) in a plane that holds a template, 0 in an idle plane (see above)
            const uint32_t tm = lane_consts[((idx >> 2) & 63) * FAST_LANE_WORDS + FLC_TMPL];
            const int tpad = min(min(sc.match, sc.mismatch) - sc.del - sc.ins, 0);
            const int plo = (tm & 0xffffu) == 0xffffu ? 0 : tpad, phi = (tm >> 16) == 0xffffu ? 0 : tpad;
            auto cv = [&](uint32_t w) {
                const int lo = (int)(short)(w & 0xffffu), hi = (int)w >> 16;
                return spair(lo <= -30000 ? plo : lo, hi <= -30000 ? phi : hi);
            };
            q.x = cv(q.x); q.y = cv(q.y); q.z = cv(q.z); q.w = cv(q.w);
        }
        *reinterpret_cast<uint4*>(&lds[idx]) = q;
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int lane = threadIdx.x & 63;
    (void)wave; (void)nw;
    const int H = Hx & 0xff;                    // carry hops (FastPlan::Hx)
    const bool bperm_ok = (Hx >> 8) & 1;        // both planes segment alike and lane Hx >> 16 is idle in both
    // Issue fairness among the waves of a SIMD (FairShare, sd_fast_dev.hpp): the rows a wave still has to fill, one
    // word per wave behind the table
    FairShare fair;
    fair.init(reinterpret_cast<int*>(lds + TBL), wave, nw, lane, queue, n_chunks);
    ChunkSched sched;
    sched.init(queue, order, n_chunks);
    for (int c = sched.next(); c >= 0; c = sched.next()) {
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    ReadStream rs;
    rs.init(bases2 + cd.woff, cd.noff >= 0 ? nmask + cd.noff : nullptr, n);
    // The carry scan through ds_bpermute trades seven VALU instructions per row for three crossbar round trips: a gain
    // while four waves keep a SIMD's issue slots busy, a loss once the chunk queue is empty and the SIMDs empty out --
    // a wave's own latency is what is left then, and the DPP form takes over
    // (decided again every 32 rows from the state of the chunk queue, FairShare::update: the SIMDs stay full while
    // waves that finish still find chunks)
    bool bperm_scan = bperm_ok && (!fair.drained || ((Hx >> 9) & 1));

    const uint32_t* lc = lane_consts + lane * FAST_LANE_WORDS;
    const uint32_t startMask = lc[FLC_STARTMASK];
    const uint32_t contMask = lc[FLC_CONTMASK];
    const uint32_t cont2Mask = lc[FLC_CONT2];
    const uint32_t oneMask = ONE ? lc[FLC_ONE] : 0u;
    const uint32_t endOffPlan = lc[FLC_ENDOFF];
    // --ed_thr: per-chunk end offsets (-inf for dropped templates) and tie-break ranks
    // fp16, unranked: the whole B reduction stays in fp16 (no per-row int conversions); `del` is folded
    // into the end offsets so that the wave maximum is directly the start term B_i + del of the next row
    constexpr bool HRED = F16 || U16;   // (ranked too: sd_rank_keep writes the per-chunk end offsets on every lane of a template)
    // HRED row tail.  With a_l = max(last slot, K_l) the total of virtual lane l:
    //   * the end cell of a template is the maximum of a_l over ALL its lanes (prefix maximum along the
    //     template), so the B reduction takes every lane's total with its template's end offset
    //     (FLC_ENDALL) and does not need the scanned value of the end lane;
    //   * the carry K_l of the row (exclusive scan of a over the template's earlier lanes) is also the
    //     true last-slot value of lane l-1, i.e. the diagonal input of slot 0 in the next row, and
    //     KB = max(K, B+del) >= K: slot 0's u is KB itself;
    //   * totals never decrease from row to row in the stored domain (the insertion move is "keep"),
    //     so the new carry replaces the old one without a max.
    const uint32_t endOffRaw = pk_adds(RANKED ? cendoff[(size_t)c * 64 + lane] : lc[FLC_ENDALL], pack2(sc.del));
    const uint32_t endOff = HRED ? CO::from_i16x2(endOffRaw)
                                 : CO::from_i16x2(RANKED ? cendoff[(size_t)c * 64 + lane] : endOffPlan);
    const uint32_t rank2 = RANKED ? crank[(size_t)c * 64 + lane] : 0u;
    const uint32_t row0adj = CO::from_i16x2(lc[FLC_ROW0]);
    const uint32_t ins2 = CO::splat(sc.ins);
    // U16: which planes of this lane hold a template; which ends take part in the row maximum (an idle plane's and, with
    // --ed_thr, a dropped template's end must stay 0 = "-inf": the end offset is 0 there and the sum is masked)
    uint32_t planeMask = 0xffffffffu, endMask = 0xffffffffu, notStart = ~startMask;
    if constexpr (U16) {
        const uint32_t tm = lc[FLC_TMPL];
        planeMask = ((tm & 0xffffu) == 0xffffu ? 0u : 0xffffu) | ((tm >> 16) == 0xffffu ? 0u : 0xffff0000u);
        endMask = ((int)(short)(endOffRaw & 0xffffu) <= -30000 ? 0u : 0xffffu) | (((int)endOffRaw >> 16) <= -30000 ? 0u : 0xffff0000u);
        asm volatile("" : "+v"(planeMask), "+v"(endMask), "+v"(notStart));   // plain AND masks in registers
    }

    int32_t* Bc = Bout + cd.row0 + (uint64_t)c;
    uint32_t* ck = ckpt + (uint64_t)cd.pad * (uint64_t)(P * 64) + lane;
    int32_t* ckb = ckbase + cd.pad;

    uint32_t L[P];
    uint32_t tb[P4];
    uint32_t K = NEGC;
    int base = 0, Brel = 0, tp = 0;
    uint32_t bdel16 = 0;  // HRED: fp16 bits of (row maximum + del), relative like the cells
    int accBV = 0;  // (B << 7 | arg-max virtual lane) of the last <=64 rows, one row per lane
    int flush_at = min(n, 64);   // next row whose B word completes a group of 64 (or the chunk): one scalar compare per row

    // `after` pins the LDS reads behind the value it names (the last slot of the row being
    // finished): hoisted above the slot loop they would need a second register set + 35 copies
    // one LDS byte address per row (the lane's 16 bytes + the table of the read symbol): ONE VALU add; the slot groups
    // are immediate offsets of ds_read_b128
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    using lds_u4 = __attribute__((address_space(3))) const u32x4_t;
    const uint32_t lds_lane = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)lds + lane * 16;
    auto load_table = [&](int r, uint32_t& after) {
        uint32_t addr = lds_lane + (uint32_t)(r * (P4 * 256));
        asm volatile("" : "+v"(addr), "+v"(after));
        lds_u4* t = (lds_u4*)(uintptr_t)addr;
#pragma unroll
        for (int c4 = 0; c4 < P4 / 4; ++c4) {
            const u32x4_t q = t[c4 * 64];
            tb[4 * c4 + 0] = q.x; tb[4 * c4 + 1] = q.y; tb[4 * c4 + 2] = q.z; tb[4 * c4 + 3] = q.w;
        }
    };
    // exclusive, template-segmented prefix maximum over the virtual lanes (both planes at once):
    // H = Vmax-1 carry hops of one lane each (DPP wave_shr:1)
    // ds_bpermute byte addresses of the lanes one and two below, or of the idle lane (whose cells are -inf in both
    // planes) where that lane belongs to another template: the shifted operands of the scan then cost no VALU
    // instruction and need no mask (the LDS crossbar is otherwise idle in this kernel; the scan's latency hides
    // under the B reduction, which depends on the same lane totals only)
    const int idle4 = (Hx >> 16) << 2;
    int bp1 = (contMask & 0xffffu) ? (lane - 1) << 2 : idle4;
    int bp2 = (cont2Mask & 0xffffu) ? (lane - 2) << 2 : idle4;
    asm volatile("" : "+v"(bp1), "+v"(bp2));
    auto excl_scan = [&](uint32_t a) {
        uint32_t inc = a;
        if (bperm_scan) {   // wave-uniform
            inc = CO::mx(inc, (uint32_t)__builtin_amdgcn_ds_bpermute(bp1, (int)inc));
            inc = CO::mx(inc, (uint32_t)__builtin_amdgcn_ds_bpermute(bp2, (int)inc));
            return (uint32_t)__builtin_amdgcn_ds_bpermute(bp1, (int)inc);
        }
        if (H <= 4) {  // doubling: window of 4 previous lanes (the masks keep it inside the template;
                       // with H = 0 they are all zero and the result is -inf everywhere)
            inc = CO::mx(inc, bfi(contMask, lane_up(inc, 1), NEGC));
            inc = CO::mx(inc, bfi(cont2Mask, lane_up(lane_up(inc, 1), 1), NEGC));
        } else {
            for (int h = 1; h < H; ++h) inc = CO::mx(a, bfi(contMask, lane_up(inc, 1), NEGC));
        }
        return bfi(contMask, lane_up(inc, 1), NEGC);
    };
    // B_{row} (relative to base) = max over template ends; arg = smallest virtual lane attaining it
    auto reduce_ends = [&](uint32_t Eend, int row) {
        if constexpr (HRED) {
            uint32_t val = CO::add(Eend, endOff);
            if constexpr (U16 && RANKED) val &= endMask;   // (unranked: an idle plane's total and end offset are both 0)
            uint32_t m;
            asm("v_max_f16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                "s_nop 1\n\t"
                "v_max_f16_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_max_f16_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_max_f16_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_max_f16_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_max_f16_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_max_f16_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                : "=&v"(m) : "v"(val));
            const uint32_t b16 = (uint32_t)__builtin_amdgcn_readlane((int)m, 63) & 0xffffu;
            unsigned long long mlo, mhi;
            asm("v_cmp_eq_f16_e64 %0, %1, %2" : "=s"(mlo) : "v"(val), "s"(b16));
            asm("v_cmp_eq_f16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0" : "=s"(mhi) : "v"(val), "s"(b16));
            if constexpr (RANKED) {
                if (__popcll(mlo) + __popcll(mhi) > 1) {   // wave-uniform: several lanes tie (often lanes of one template)
                    // ties go to the first template of the chunk's filtered order (main.cpp:141-147)
                    const int klo = ((mlo >> lane) & 1ull) ? (int)(rank2 & 0xffffu) : 0x7fff;
                    const int khi = ((mhi >> lane) & 1ull) ? (int)(rank2 >> 16) : 0x7fff;
                    const int kmin = -wave_max(-min(klo, khi));
                    mlo = __ballot(klo == kmin);
                    mhi = __ballot(khi == kmin);
                }
            }
            // (some lane attains the maximum: ctz never sees two empty masks -- s_ff1 x 2, add, compare, select)
            const int v = mlo ? __builtin_ctzll(mlo) : 64 + __builtin_ctzll(mhi);
            bdel16 = b16;
            const int slot = (row - 1) & 63;
            acc_put(accBV, (int)((b16 << 7) | (uint32_t)v), slot);
            if (row == flush_at) {   // slot == 63 || row == n
                flush_at = min(n, row + 64);
                // 64 rows at once: fp16 -> int, B = base + (b + del) - del + tp_row * ins
                asm volatile("");   // keeps this a scalar branch: the lane test below is not evaluated on every row
                const uint32_t w = (uint32_t)accBV;
                const int bi = U16 ? (int)(w >> 7) - U16_BIAS : (int)(float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 7));
                const int tpl = tp - (slot - lane);
                const int Bv = base + bi - sc.del + tpl * sc.ins;
                if (lane <= slot) Bc[row - slot + lane] = (int)(((uint32_t)Bv << 7) | (w & 127u));
            }
            return;
        }
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
        if (row == flush_at) {   // slot == 63 || row == n
            flush_at = min(n, row + 64);
            if (lane <= slot) Bc[row - slot + lane] = accBV;
        }
    };

    // ---- row 0 (main.cpp:171-182): E[0][k] = max(E[0][k-1], mm_k - del), E[0][0] = mm_0
    uint32_t pin = 0;
    load_table(rs.code(0), pin);
    rs.advance(0);
    if constexpr (U16) {
        // values: BIAS + ..., idle planes 0
        L[0] = (tb[0] + row0adj + U16_BIAS2) & planeMask;
#pragma unroll
        for (int q = 1; q < P; ++q) L[q] = CO::mx(L[q - 1], (tb[q] + ins2 + U16_BIAS2) & planeMask);
    } else {
    L[0] = CO::add(tb[0], row0adj);
#pragma unroll
    for (int q = 1; q < P; ++q) L[q] = CO::mx(L[q - 1], CO::add(tb[q], ins2));
    }
    int rcur = rs.code(1);   // read symbol of the row the loop is about to fill
    load_table(rcur, L[P - 1]);
    rs.advance(1);
    std::conditional_t<U16, U16Guard<P>, F16Guard<P>> guard;
    if constexpr (F16) {
        guard.start(L[P - 1], sc.guard_lim);
        guard.check_low(L);
    }
    if constexpr (U16) {
        guard.start(sc.guard_lim);
        guard.check_low(L, planeMask);
    }
    // (a 1-bp template's lane ends at slot 0 in row 0 too: with fp16 cells its pads are -inf and the last slot IS slot 0's
    // value; the u16 format's pads are finite -- tpad + ins may exceed mm_0 -- so the end is taken where it is)
    const uint32_t a0 = ONE ? bfi(oneMask, L[0], L[P - 1]) : L[P - 1];
    if constexpr (HRED) reduce_ends(a0, 1);
    K = excl_scan(a0);
    uint32_t Eend = CO::mx(a0, K);
    if constexpr (!HRED) reduce_ends(Eend, 1);

    // rows in groups of FAST_R: the per-group work (fairness, scan form, rebase, checkpoint) sits between two inner loops
    // instead of behind a test in every row
    for (int i = 1; i < n;) {
        if ((i & (FAST_R - 1)) == 0) {
            fair.update(n - i);
            bperm_scan = bperm_ok && (!fair.drained || ((Hx >> 9) & 1));
            if ((i & (127 >> ((Hx >> 11) & 1))) == 0) {   // FastPlan::rebase rows: 128, or 64 (Hx bit 11; Hx is live here anyway)
                // rebase the int16 state on B_i and fold the row offset tp*ins back in
                if constexpr (F16)
                    Brel = __builtin_amdgcn_readfirstlane((int)(float)__builtin_bit_cast(_Float16, (unsigned short)bdel16)) -
                           sc.del + tp * sc.ins;
                if constexpr (U16) Brel = (int)bdel16 - U16_BIAS - sc.del + tp * sc.ins;
                const uint32_t d2 = U16 ? pack2(-(Brel - tp * sc.ins)) : CO::splat(F16 ? -(Brel - tp * sc.ins) : Brel - tp * sc.ins);
                base += Brel;
                Brel = 0;
                tp = 0;
                if constexpr (F16) bdel16 = __builtin_amdgcn_readfirstlane((int)(CO::splat(sc.del) & 0xffffu));
                if constexpr (U16) bdel16 = (uint32_t)(U16_BIAS + sc.del);
                if constexpr (U16) {
                    // per-half (wrapping) add of the shift, real planes only: idle planes and the carry of start lanes stay 0
                    typedef unsigned short u2_t __attribute__((ext_vector_type(2)));
                    auto shift = [](uint32_t x, uint32_t d) {
                        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u2_t, x) + __builtin_bit_cast(u2_t, d));
                    };
                    const uint32_t d2m = d2 & planeMask;
                    guard.check_high(L);
                    K = shift(K, d2m & notStart);
#pragma unroll
                    for (int s = 0; s < P; ++s) L[s] = shift(L[s], d2m);
                    guard.check_low(L, planeMask);
                } else if constexpr (F16) {
                    guard.check_high(L);
                    K = bfi(startMask, NEGC, CO::add(K, d2));
                    Eend = CO::add(Eend, d2);
#pragma unroll
                    for (int s = 0; s < P; ++s) L[s] = CO::add(L[s], d2);
                    guard.check_low(L);
                } else {
                K = bfi(startMask, NEG2, pk_subs(K, d2));
                Eend = pk_subs(Eend, d2);
#pragma unroll
                for (int s = 0; s < P; ++s) L[s] = pk_subs(L[s], d2);
                }
            }
            // checkpoint the (true) row i-1 for the traceback: E = ckbase + stored value
            const int q = (i / FAST_R) - 1;
#pragma unroll
            for (int s = 0; s < P; ++s) ck[(uint64_t)q * (P * 64) + s * 64] = CO::mx(L[s], K);
            if (lane == 0) ckb[q] = base + tp * sc.ins;
        }
        const int iend = min(n, (i | (FAST_R - 1)) + 1);
        for (; i < iend; ++i) {
        uint32_t KB;
        if constexpr (HRED) {
            // max(K, {b+del, b+del}): the scalar's low half feeds both lanes of the packed op
            asm("v_pk_max_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(KB) : "v"(K), "s"(bdel16));
            if constexpr (U16) KB &= planeMask;   // an idle plane takes no start term: its slots stay 0
        } else {
            KB = CO::mx(K, CO::splat(Brel + sc.del - tp * sc.ins));
        }
        // slot 0's diagonal input is the true last slot of the previous lane = this lane's carry K, and
        // KB >= K: its u is KB itself (all variants)
        const uint32_t w0 = U16 ? (L[0] & notStart) : bfi(startMask, NEGC, L[0]);
        uint32_t u_[P], v_[P], c_[P];
        uint32_t run = 0;
        // software-pipelined over the slots so that no packed op consumes the result of the
        // instruction right before it (gfx950 needs a wait state there)
        if constexpr (HRED) {
            if constexpr (FLS > 0) {
                // four levels: FL, FL - FLS, FL - 2 FLS, FL - 3 FLS (not below 1); the floors of a level's group are skipped
                // by the rows whose symbol needs fewer
                constexpr int FL3 = FL - 3 * FLS > 1 ? FL - 3 * FLS : 1;
                constexpr int FL2 = FL - 2 * FLS > FL3 ? FL - 2 * FLS : FL3;
                constexpr int FL1 = FL - FLS > FL2 ? FL - FLS : FL2;
                const int lv = (Hx >> (22 + 2 * rcur)) & 3;   // scalar: the level of this row's read symbol
#pragma unroll
                for (int q = 1; q <= FL3 && q < P; ++q) L[q - 1] = CO::mx(L[q - 1], KB);
                if (lv <= 2) {
#pragma unroll
                    for (int q = FL3 + 1; q <= FL2 && q < P; ++q) L[q - 1] = CO::mx(L[q - 1], KB);
                }
                if (lv <= 1) {
#pragma unroll
                    for (int q = FL2 + 1; q <= FL1 && q < P; ++q) L[q - 1] = CO::mx(L[q - 1], KB);
                }
                if (lv == 0) {
#pragma unroll
                    for (int q = FL1 + 1; q <= FL && q < P; ++q) L[q - 1] = CO::mx(L[q - 1], KB);
                }
            }
            const uint32_t w0f = FLS > 0 ? (U16 ? (L[0] & notStart) : bfi(startMask, NEGC, L[0])) : w0;   // (slot 0's keep operand, after the in-place floor)
#pragma unroll
            for (int s = 0; s < P + 4; ++s) {
                if (s >= 4) {
                    const int q = s - 4;
                    // slot 0 (HRED): the old carry K joins the chain here, so that the last slot is the
                    // lane total without a separate max(L[P-1], K) at the end of the row
                    L[q] = q == 0 ? CO::mx3(v_[0], w0f, K) : CO::mx3(L[q - 1], v_[q], L[q]);
                }
                if (s >= 2 && s - 2 < P) {
                    const int q = s - 2;
                    v_[q] = CO::add(u_[q], tb[q]);
                }
                if (s < P) {
                    const int q = s;
                    // The floor max(., KB) of the diagonal input matters only where the table value of the lane
                    // sets a new record along the slots (slot 1 and, typically, the first matching slot): everywhere
                    // else KB + tbl[q] is already below what the chain carries from an earlier slot.  The plan
                    // finds the last such slot over all lanes and symbols (FastPlan::floor_slots <= FL).
                    if (q == 0) u_[q] = KB;
                    else if (FLS == 0 && q <= FL) u_[q] = CO::mx(L[q - 1], KB);
                    else u_[q] = L[q - 1];
                }
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
                c_[q] = pk_max(v_[q], q == 0 ? w0 : L[q]);
            }
            if (s >= 1 && s - 1 < P) {
                const int q = s - 1;
                v_[q] = pk_adds(u_[q], tb[q]);
            }
            if (s < P) {
                const int q = s;
                u_[q] = q == 0 ? KB : q <= FL ? pk_max(L[q - 1], KB) : L[q - 1];   // see the fp16 loop
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        rcur = rs.code(i + 1);
        load_table(rcur, L[P - 1]);  // unconditional (clamped): keeps tb[] out of phi copies
        rs.advance(i + 1);
        uint32_t a = HRED ? L[P - 1] : CO::mx(L[P - 1], K);  // fp16 / u16: K already joined the chain
        // a 1-bp template ends in slot 0: the pads behind a k = 0 cell keep their old value when the cell's falls
        if constexpr (ONE) a = bfi(oneMask, L[0], a);   // (the lane is a start lane: no carry to join)
        ++tp;
        if constexpr (HRED) {
            // the scan's three crossbar trips run under the DPP chain of the B reduction (both need only `a`); the
            // reduction is ONE piece of code for both scan forms (its accumulator stays in one register)
            uint32_t t1 = 0;
            if (bperm_scan) t1 = (uint32_t)__builtin_amdgcn_ds_bpermute(bp1, (int)a);
            reduce_ends(a, i + 1);
            if (bperm_scan) {
                const uint32_t i1 = CO::mx(a, t1);
                const uint32_t i2 = CO::mx(i1, (uint32_t)__builtin_amdgcn_ds_bpermute(bp2, (int)i1));
                K = (uint32_t)__builtin_amdgcn_ds_bpermute(bp1, (int)i2);
            } else {
                K = excl_scan(a);
            }
        } else {
            K = excl_scan(a);           // totals never decrease: the new carry replaces the old one
            Eend = CO::mx(a, K);
            reduce_ends(Eend, i + 1);
        }
        }   // rows of the group
    }
    if constexpr (HRED) {
        guard.check_high(L);
        guard.finish(sc.guard_flag);
    }
    }  // chunk queue
    fair.leave();
}


}  // namespace sd
