// sd_fast_trace2.hip -- traceback of the fast family, second form: packed 16-bit recomputation, two 32-row blocks per wave.
//
// Same job as sd_fast_trace (sd_fast.hip): walk the reference's traceback (main.cpp:217-269) and, for every monomer
// instance (template j, end row e), recompute only template j's cells from the fill's checkpoints (every FAST_R = 32
// rows) to learn the moves.  What changes is how a wave spends its instructions:
//
//  * TWO blocks per step.  The rows of an instance are needed from the bottom up, but the cells of a block can be
//    recomputed from its checkpoint before the walk gets there: lanes 0..31 recompute block A (the block that holds the
//    walk's row i), lanes 32..63 block B (the 32 rows above it) in the same instructions, then the walk goes through
//    both.  The per-row cross-lane work (a prefix maximum over the lanes for the deletion chain, one row-info word, one
//    table address) serves two rows.  When the walk can be expected to end inside block A (k + 1 + margin <= rows of
//    A) the step runs A alone over A's rows only.
//  * TWO cells per lane-operation.  Cells are 16-bit words  U = 4*(E' - base) + tag + 0x4000  packed two to a VGPR
//    (low half / high half = two "planes": lane l of a half-wave owns the virtual lanes 2l and 2l+1, each QQ consecutive
//    template cells, 64 virtual lanes per half-wave).  E' = E - i*ins is the fill's row-shifted domain
//    (the insertion move is "keep"), base = the block's first start term, and the two low bits carry the reference's
//    traceback priority DEL 3 > INS 2 > DIAG 1 > START 0 (main.cpp:242-253) so that ONE maximum yields the value and,
//    among equal values, the move the reference's equality tests pick first (as in sd_fast_trace).  Every word stays
//    inside 0x0400..0x7BFF (fast_plan_build proves the range, FastPlan::tr2_ok), where gfx950's fp16 maxima are exact
//    UNSIGNED INTEGER maxima on the bit patterns (tools/ubench_u16max.hip, exhaustive): the cell update uses
//    v_pk_max_f16 / v_pk_maximum3_f16 on integers -- a three-input maximum does not exist for packed int16 -- the
//    table add is a plain 32-bit add of a signed pair, "| 3" tags the deletion candidate, and 0 is the identity of every
//    maximum (a DPP shift's zero fill, an AND with a lane mask).
//
// Per template and level QQ (cells per virtual lane, 1..QM) the host lays out (FastPlan::tr2_tab):
//    mt[5][QQ][32]     4*(mm - del - ins) - 1 of the lane's two cells of register q for the five read symbols, as a
//                      signed pair (hi * 65536 + lo); 0 on padding cells
//    ck[QQ][2][32]     where the cell sits in a checkpoint of the fill: slot * 64 + lane | plane << 31; ~0 = padding
// Applies to the layouts with one wave per chunk in the fill (narrow, and wide up to 128 templates) and templates of up to
// 64 * 4 = 256 bp; everything else keeps sd_fast_trace.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "sd_fast.hpp"
#include "sd_fast_dev.hpp"

namespace sd {

namespace {

constexpr int TR2_BIAS = 0x4000;
constexpr uint32_t TR2_TAGS = 0x00030003u;
constexpr int TR2_NWV = 1;   // waves per workgroup: 6.3 KB of LDS each, 26 fit a CU (6-7 per SIMD)

// u = max(pd, {b, b}) with b = the low half of w (the row's start term), both planes at once
__device__ __forceinline__ uint32_t tr2_floor(uint32_t pd, uint32_t w) {
    uint32_t u;
    asm("v_pk_max_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(u) : "v"(pd), "v"(w));
    return u;
}

// exclusive prefix maximum, over the lanes of each half-wave, of the lane totals max(lo plane, hi plane) of `Rl`
// (unsigned 16-bit values as fp16 patterns; 0 = identity); returns the carry word {lo plane: everything before this
// lane, hi plane: that joined with this lane's lo plane}
__device__ __forceinline__ uint32_t tr2_carry(uint32_t Rl) {
    uint32_t TT, e, tot;
    asm volatile(
        "v_max_f16_sdwa %0, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "s_nop 1\n\t"
        "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_max_f16_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f16_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f16_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f16_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_f16_e32 %2, %1, %0\n\t"
        "s_nop 1\n\t"
        "v_max_f16_dpp %1, %2, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_f16_e32 %2, %1, %3"
        : "=&v"(TT), "=&v"(e), "=&v"(tot)
        : "v"(Rl));
    // tot = max(e, lo plane of Rl): the hi plane's carry
    return __builtin_amdgcn_perm(tot, e, 0x05040100u);
}

}  // namespace

template <int QM>
__global__ __launch_bounds__(64 * TR2_NWV, 7) void sd_fast_trace_pk(
    const ChunkDesc* __restrict__ chunks, int n_chunks, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint32_t* __restrict__ lane_consts, const uint8_t* __restrict__ tcodes,
    const int32_t* __restrict__ toff, const int32_t* __restrict__ tlen, ScoreArgs sc, int P,
    const int32_t* __restrict__ B, const uint32_t* __restrict__ ckpt, const int32_t* __restrict__ ckbase,
    const uint32_t* __restrict__ tr2, DevRec* __restrict__ recs, int32_t* __restrict__ rec_cnt,
    int* __restrict__ queue, const int* __restrict__ order, int ckf16, int bshift, int margin, int xlim) {
    __shared__ uint8_t pt_all[TR2_NWV][2][FAST_R][64];      // 2-bit moves: [half][row][virtual lane] one byte = QQ cells
    __shared__ uint32_t mt_all[TR2_NWV][5 * QM * 32];       // table of the current (template, level)
    __shared__ uint32_t ri_all[TR2_NWV][64];                // row info of the two blocks: start term | table offset << 16
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int ll = lane & 31, hh = lane >> 5;
    uint8_t(*pt)[FAST_R][64] = pt_all[wave];
    uint32_t* mtw = mt_all[wave];
    uint32_t* ri = ri_all[wave];
    // lane masks: the first lane of a half-wave has no lane below it (diagonal input of its first cell), and its first
    // cell is k = 0, which has no insertion move in the fill (main.cpp:193-203)
    uint32_t pdmask = ll == 0 ? 0u : 0xffffffffu;
    uint32_t insmask = ll == 0 ? 0xffff0000u : 0xffffffffu;
    uint32_t two2 = 0x00020002u;
    asm volatile("" : "+v"(pdmask), "+v"(insmask), "+v"(two2));   // plain AND masks in registers (not selects on the lane test)
    constexpr int STRIDE = 224 * QM * (QM + 1) / 2;

    ChunkSched sched;
    sched.init(queue, order, n_chunks);
    for (int c = sched.next(); c >= 0; c = sched.next()) {
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    ReadCursor rc{bases2 + cd.woff, cd.noff >= 0 ? nmask + cd.noff : nullptr};
    const int32_t* BVc = B + cd.row0 + (uint64_t)c;   // (B << 7) | arg-max virtual lane
    const int bmask = (1 << bshift) - 1;
    auto Bof = [&](int r) { return BVc[r] >> bshift; };
    auto Vof = [&](int r) { return BVc[r] & bmask; };
    auto tmpl_of = [&](int v) {
        const uint32_t t = lane_consts[(v & 63) * FAST_LANE_WORDS + FLC_TMPL];
        return (int)(((v >> 6) & 1) ? (t >> 16) : (t & 0xffffu));
    };
    DevRec* out = recs + cd.row0;
    const int ins = sc.ins, del = sc.del;

    int cnt = 0;
    int e = n - 1;
    int j = tmpl_of(Vof(n));
    while (true) {
        const int Lj = tlen[j];
        const int x0 = toff[j];
        const uint32_t* tab = tr2 + (size_t)j * STRIDE;
        const int code0 = tcodes[x0];   // template cell k = 0
        struct Pos { int i, k; bool done; };
        int i = e, k = Lj - 1;
        bool stop_row0 = false;

        // one step at level QQ: recompute block A = rows [a0, i] (lanes 0..31) and, unless the walk is expected to end
        // inside A, block B = rows [a0 - 32, a0) (lanes 32..63); then walk.  Returns done = the instance start was reached.
        auto step = [&](auto qq_c, const int i_in, const int k_in) -> Pos {
            constexpr int QQ = decltype(qq_c)::value;
            constexpr int LOFF = 224 * (QQ - 1) * QQ / 2;
            const uint32_t* tck = tab + LOFF + 5 * QQ * 32;   // checkpoint map [QQ][2][32]
            int i = i_in, k = k_in;
            const int a0 = i & ~(FAST_R - 1);
            const int nA = i - a0 + 1;
            const bool hasB = a0 >= 2 * FAST_R && (k + 1 + margin > nA);
            const int aB = a0 - FAST_R;
            const int rs0 = a0 == 0 ? 1 : a0;
            const int baseA = Bof(rs0) + del - (rs0 - 1) * ins;
            const int baseB = hasB ? Bof(aB) + del - (aB - 1) * ins : 0;
            const int ab = hh ? aB : a0;
            const int mybase = hh ? baseB : baseA;
            const bool act = hh == 0 || hasB;
            // row info, one row per lane (lane ll of a half holds row ab + ll)
            bool badrow = false;
            {
                const int rl = ab + ll;
                const bool rvalid = act && rl >= 1 && rl <= (hh ? a0 - 1 : i);
                uint32_t w = 0;
                if (rvalid) {
                    const int bx = Bof(rl) + del - (rl - 1) * ins - mybase;   // the row's start term relative to the block's first
                    badrow = bx > xlim || bx < -xlim;
                    const int bd = 4 * bx + 1 + TR2_BIAS;
                    w = ((uint32_t)bd & 0xffffu) | ((uint32_t)(rc.code(rl) * (QQ * 128)) << 16);
                } else if (rl == 0 && act) {
                    // row 0 (main.cpp:171-182): E[0][k] = max(E[0][k-1], mm_k - del), E[0][0] = mm_0: the regular update with
                    // no row above and the start term "ins" (the k = 0 cell is adjusted in the row itself)
                    const int bd = 4 * (ins - mybase) + 1 + TR2_BIAS;
                    w = ((uint32_t)bd & 0xffffu) | ((uint32_t)(rc.code(0) * (QQ * 128)) << 16);
                }
                ri[lane] = w;
            }
            // the row above each block, from the fill's checkpoint (row ab - 1), or nothing above row 0
            uint32_t T[QQ];
#pragma unroll
            for (int q = 0; q < QQ; ++q) T[q] = 0;
            bool bad = badrow;   // run-time check of the plan's range proof: a word outside the exact range raises the guard flag
            if (a0 != 0 && act) {
                const int q0 = ab / FAST_R - 1;
                const int32_t cb = ckbase[cd.pad + q0];
                const uint32_t* ckq = ckpt + ((uint64_t)cd.pad + (uint64_t)q0) * (uint64_t)(P * 64);
                const int shift = (ab - 1) * ins + mybase - cb;
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    uint32_t pair = 0;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const uint32_t idx = tck[(q * 2 + pl) * 32 + ll];
                        uint32_t t16 = 0;
                        if (idx != 0xffffffffu) {
                            const uint32_t wv = ckq[idx & 0x7fffffffu];
                            const uint32_t hw = (idx >> 31) ? (wv >> 16) : (wv & 0xffffu);
                            // checkpoint cell formats: 0 int16, 1 fp16, 2 biased u16 (0 = "-inf")
                            const int val = ckf16 == 2 ? (hw ? (int)hw - U16_BIAS : -0x10000000)
                                            : ckf16 ? (int)(float)__builtin_bit_cast(_Float16, (unsigned short)hw) : (int)(short)hw;
                            const int X = max(val, -0x08000000) - shift;
                            bad = bad || X > xlim || X < -xlim;
                            t16 = (uint32_t)(4 * min(max(X, -xlim), xlim) + 2 + TR2_BIAS);
                        }
                        pair |= t16 << (16 * pl);
                    }
                    T[q] = pair;
                }
            }
            if (__ballot(bad) != 0ull && sc.guard_flag && lane == 0) atomicOr(sc.guard_flag, 2);

            uint8_t* prow = &pt[hh][0][2 * ll];
            const uint32_t* rip = ri + hh * 32;
            const uint32_t* mlane = mtw + ll;
            auto row = [&](auto row0_c, const int t, const uint32_t w, const uint32_t (&mm)[QQ]) {
                constexpr bool ROW0 = decltype(row0_c)::value;
                // diagonal inputs of the two first cells: lo plane <- hi plane of the lane below (row above), hi plane <- lo plane
                const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T[QQ - 1], 0x138 /*wave_shr:1*/, 0xf, 0xf, true) & pdmask;
                uint32_t pd = __builtin_amdgcn_alignbit(T[QQ - 1], sh, 16);
                uint32_t loc[QQ];
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    const uint32_t u = tr2_floor(pd, w);
                    uint32_t v = u + mm[q];
                    if (ROW0 && q == 0) v += (ll == 0 ? (uint32_t)(4 * del) : 0u);   // E[0][0] = mm_0, not mm_0 - del
                    if (q == 0) loc[0] = CellOps<true>::mx(v, T[0] & insmask);
                    else loc[q] = CellOps<true>::mx3(loc[q - 1] | TR2_TAGS, v, T[q]);
                    pd = T[q];
                }
                const uint32_t C = tr2_carry(loc[QQ - 1] | TR2_TAGS);
                uint32_t acc = 0;
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    const uint32_t Ef = CellOps<true>::mx(loc[q], C);
                    acc = (acc << 2) | (Ef & TR2_TAGS);
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(T[q]) : "v"(Ef), "s"(~TR2_TAGS), "v"(two2));   // (Ef & ~tags) | INS tag
                }
                // byte 0 = the lo plane's moves, byte 2 = the hi plane's: one 16-bit store {virtual lane 2l, 2l + 1}
                *reinterpret_cast<uint16_t*>(prow + t * 64) = (uint16_t)__builtin_amdgcn_perm(0u, acc, 0x0c0c0200u);
            };
            // the row loop, software-pipelined over the LDS reads: the table words of row t + 1 and the row info of row
            // t + 2 are in flight while row t computes (two dependent LDS round trips per row otherwise)
            auto table_of = [&](const uint32_t w, uint32_t (&mm)[QQ]) {
                const uint32_t* mrow = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(mlane) + (w >> 16));
#pragma unroll
                for (int q = 0; q < QQ; ++q) mm[q] = mrow[q * 32];
            };
            const int nIter = hasB ? FAST_R : nA;
            int t = 0;
            uint32_t w0 = rip[0];
            uint32_t mm0[QQ], mm1[QQ];
            table_of(w0, mm0);
            if (a0 == 0) {
                row(std::true_type(), 0, w0, mm0);
                w0 = rip[1];
                table_of(w0, mm0);
                t = 1;
            }
            uint32_t w1 = rip[(t + 1) & (FAST_R - 1)];   // (ri has 32 words per half: the index wraps to a valid word)
            // two rows per trip, the table registers ping-pong: at the top w0 / mm0 belong to row t, w1 to row t + 1
            for (; t + 1 < nIter; t += 2) {
                table_of(w1, mm1);
                const uint32_t w2 = rip[(t + 2) & (FAST_R - 1)];
                row(std::false_type(), t, w0, mm0);
                table_of(w2, mm0);
                const uint32_t w3 = rip[(t + 3) & (FAST_R - 1)];
                row(std::false_type(), t + 1, w1, mm1);
                w0 = w2;
                w1 = w3;
            }
            if (t < nIter) row(std::false_type(), t, w0, mm0);

            // walk (wave-uniform)
            const int low = hasB ? aB : a0;
            while (i >= low) {
                if (k == 0) return Pos{i, k, false};   // the k = 0 cell needs no recomputed cells: the caller's rule
                // Lane l looks at the cell l diagonal moves ahead, (i - l, k - l): the run of DIAG tags from lane 0 on is
                // taken in one go (reads of ~90 % identity are mostly diagonal runs), then the move that ends it.
                const int il = i - lane, kl = k - lane;
                const bool ok = il >= low && kl >= 1;
                int tg = 0;
                if (ok) {
                    const int hb = il < a0 ? 1 : 0;
                    const int tr = il - (hb ? aB : a0);
                    const int v = kl / QQ, q = kl - v * QQ;
                    tg = ((int)pt[hb][tr][v] >> (2 * (QQ - 1 - q))) & 3;
                }
                const unsigned long long dg = __ballot(ok && tg == 1);
                const int run = dg == ~0ull ? 64 : __builtin_ctzll(~dg);
                i -= run;
                k -= run;
                if (run == 64 || i < low || k < 1) continue;   // (the loop head deals with the block edge and with k = 0)
                const int te = __builtin_amdgcn_readlane(tg, run);
                if (te == 3) { --k; }
                else if (te == 2) { --i; }
                else { return Pos{i, k, true}; }
            }
            return Pos{i, k, false};
        };

        int cur = 0;
        while (true) {
            if (k == 0) {
                // The k = 0 cell holds the start term only (main.cpp:188-193), but the reference's traceback still tests
                // the insertion there (main.cpp:245): dp[i][j][0] == dp[i-1][j][0] + ins, both sides start terms -- a
                // matter of B and two read symbols, no block is recomputed for it (a 1-bp template never needs one).
                if (i == 0) { stop_row0 = true; break; }
                const int ci = rc.code(i), cp = rc.code(i - 1);
                const int lhs = Bof(i) + (ci == code0 ? sc.match : sc.mismatch);
                const int rhs = (i >= 2 ? Bof(i - 1) : 0) + (cp == code0 ? sc.match : sc.mismatch) + ins;
                if (lhs == rhs) { --i; continue; }
                break;   // START (stop_row0 stays false: i >= 1)
            }
            const int need = (k >> 6) + 1;
            Pos ps;
            auto go = [&](auto qq_c) {
                constexpr int QQ = decltype(qq_c)::value;
                if (cur != QQ) {
                    constexpr int LOFF = 224 * (QQ - 1) * QQ / 2;
                    for (int x = lane; x < 5 * QQ * 32; x += 64) mtw[x] = tab[LOFF + x];
                    cur = QQ;
                }
                ps = step(qq_c, i, k);
            };
            if (QM >= 4 && need >= 4) go(std::integral_constant<int, (QM >= 4 ? 4 : 1)>());
            else if (QM >= 3 && need == 3) go(std::integral_constant<int, (QM >= 3 ? 3 : 1)>());
            else if (QM >= 2 && need == 2) go(std::integral_constant<int, (QM >= 2 ? 2 : 1)>());
            else go(std::integral_constant<int, 1>());
            i = ps.i;
            k = ps.k;
            if (ps.done) { stop_row0 = (i == 0); break; }
        }
        if (lane == 0) {
            DevRec rec;
            rec.tmpl = j;
            rec.start = i;
            rec.end = e;
            rec.score = Bof(e + 1) - (stop_row0 ? 0 : Bof(i));  // main.cpp:255 / 258-262
            out[cnt] = rec;
        }
        ++cnt;
        if (stop_row0) break;
        j = tmpl_of(Vof(i));  // between-monomers hop, main.cpp:228-236
        e = i - 1;
    }
    if (lane == 0) rec_cnt[c] = cnt;
    }  // chunk queue
}

// host: the per-template tables of the kernel above (called by fast_plan_build for the narrow layout)
void fast_plan_trace2(const std::vector<std::string>& tseq, ScoreArgs sc, FastPlan& plan) {
    plan.tr2_ok = false;
    plan.tr2_tab.clear();
    if (plan.waves != 1 || plan.Lmax > 64 * 4) return;
    auto ab = [](int v) { return v < 0 ? -v : v; };
    const int maxabs = std::max(std::max(ab(sc.ins), ab(sc.del)), std::max(ab(sc.mismatch), ab(sc.match)));
    // Range of X = E' - base inside a pair of blocks.  A cell is at most (Lmax - 1)*|del| above the next row's B (the rest
    // of the template can be deleted) and at least one mismatch below the current row's B (the start term is a
    // candidate of every cell); B moves by at most G = max(0, smax - del) up and |ins| down per row and the row shift
    // adds |ins| per row; base is the B term of the block's first row, a block has 32 rows (+1 for the checkpoint row).
    const int smax = std::max(sc.match, sc.mismatch);
    const int64_t G = std::max(std::max(0, smax - sc.del), sc.ins);
    const int64_t R2 = (int64_t)(plan.Lmax + 1) * ab(sc.del) + 36 * (G + 2 * (int64_t)ab(sc.ins)) + 8 * (int64_t)maxabs + 16;
    if (4 * R2 + 16 > 15000) return;
    plan.tr2_bound = (int)R2;
    plan.tr2_xlim = (int)(15000 / 4 - 4);
    const int QM = (plan.Lmax + 63) / 64;
    plan.tr2_qm = QM;
    const int stride = 224 * QM * (QM + 1) / 2;
    const int T = (int)tseq.size();
    plan.tr2_tab.assign((size_t)T * (size_t)stride, 0u);
    size_t x0 = 0;
    for (int j = 0; j < T; ++j) {
        const std::string& s = tseq[(size_t)j];
        const int L = (int)s.size();
        uint32_t* tj = &plan.tr2_tab[(size_t)j * (size_t)stride];
        for (int QQ = 1; QQ <= QM; ++QQ) {
            uint32_t* mt = tj + 224 * (QQ - 1) * QQ / 2;
            uint32_t* ck = mt + 5 * QQ * 32;
            for (int ll = 0; ll < 32; ++ll)
                for (int q = 0; q < QQ; ++q) {
                    int kk[2];
                    for (int pl = 0; pl < 2; ++pl) {
                        const int k = (2 * ll + pl) * QQ + q;
                        kk[pl] = k < L ? k : -1;
                        uint32_t idx = 0xffffffffu;
                        if (kk[pl] >= 0) {
                            const uint32_t so = plan.slot_of[x0 + (size_t)k];   // (slot << 7) | virtual lane of the fill
                            idx = ((so >> 7) & 511u) * 64u + (so & 63u);
                            if ((so >> 6) & 1u) idx |= 0x80000000u;
                        }
                        ck[(q * 2 + pl) * 32 + ll] = idx;
                    }
                    for (int b = 0; b < 5; ++b) {
                        int val[2];
                        for (int pl = 0; pl < 2; ++pl) {
                            if (kk[pl] < 0) { val[pl] = 0; continue; }
                            const char ch = s[(size_t)kk[pl]];
                            const int cd = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
                            val[pl] = 4 * ((cd == b ? sc.match : sc.mismatch) - sc.del - sc.ins) - 1;
                        }
                        mt[(b * QQ + q) * 32 + ll] = (uint32_t)(val[1] * 65536 + val[0]);
                    }
                }
        }
        x0 += (size_t)L;
    }
    plan.tr2_ok = true;
}

bool launch_fast_trace2(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks, const uint32_t* bases2,
                        const uint32_t* nmask, const uint32_t* lane_consts, const uint8_t* tcodes, const int32_t* toff,
                        const int32_t* tlen, ScoreArgs sc, const int32_t* B, const uint32_t* ckpt, const int32_t* ckbase,
                        const uint32_t* tr2_tab, DevRec* recs, int32_t* rec_cnt, int* queue, const int* order, int n_cu) {
    if (!plan.tr2_ok || tr2_tab == nullptr) return false;
    int bpc = 26;
    if (const char* ev = getenv("SD_TRACE_BPC")) bpc = std::max(1, atoi(ev));  // developer knob
    int grid = std::min((n_chunks + TR2_NWV - 1) / TR2_NWV, bpc * n_cu);     // persistent: as many one-wave workgroups as the LDS of a CU holds
    if (const char* ev = getenv("SD_TRACE_GRID")) grid = std::max(1, atoi(ev));
    int margin = 3;
    if (const char* ev = getenv("SD_TRACE_MARGIN")) margin = atoi(ev);        // developer knob: when block B is skipped
#define SD_TRACE2(QQ)                                                                                              \
    hipLaunchKernelGGL(sd_fast_trace_pk<QQ>, dim3(grid), dim3(64 * TR2_NWV), 0, st, chunks, n_chunks, bases2, nmask, \
                       lane_consts, tcodes, toff, tlen, sc, plan.P, B, ckpt, ckbase, tr2_tab, recs, rec_cnt, queue,  \
                       order, plan.u16 ? 2 : plan.f16 ? 1 : 0, plan.bshift, margin, plan.tr2_xlim)
    switch (plan.tr2_qm) {
        case 1: SD_TRACE2(1); break;
        case 2: SD_TRACE2(2); break;
        case 3: SD_TRACE2(3); break;
        default: SD_TRACE2(4); break;
    }
#undef SD_TRACE2
    return true;
}

}  // namespace sd
