// sd_fast.hip -- fast device path of the DP fill + traceback for gfx950 (MI355X, wave64).
//
// Replaces AlignPartClassicDP (reference stringdecomposer/src/main.cpp:151-270) for scorings and
// template sets that fit the packed-int16 lane layout described in sd_fast.hpp.
//
//  sd_fast_fill   persistent waves, one chunk per wave at a time (atomic chunk queue).  Row-synchronous
//                 sweep (rows are strictly sequential because B_i = max_j dp[i-1][j][L_j-1] feeds
//                 every cell of row i, main.cpp:184-193); all template cells of a row are processed
//                 two per VALU lane-op with v_pk_*_i16, in the domain S = D - k*del - base - tp*ins
//                 where the in-row deletion chain is a prefix maximum and the insertion move is
//                 "keep" (see the comment above the kernel).  Per cell pair: 4 packed ops
//                     u = max(S[x-1], KB);  v = u + tbl;  c = max(v, S[x]);  run = max(run, c)
//                 The cross-lane carry K of the chain is applied lazily (true value = max(local, K)).
//                 The (mm - del - ins) table of all templates lives in LDS, indexed by the read base.
//                 Written to HBM per row: one word (B_i << 7 | arg-max virtual lane); every FAST_R
//                 rows a checkpoint of the row (values relative to a moving base).
//                 No per-cell back-pointers are stored: the traceback recomputes them.
//  sd_fast_trace  persistent waves, one chunk per wave at a time.  Walks the reference's traceback
//                 (main.cpp:217-269); for every monomer instance it recomputes only that
//                 template's cells, block by block from the checkpoints, derives the reference's
//                 priority-encoded moves (DEL > INS > DIAG > START) into LDS and follows them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "sd_fast.hpp"
#include "sd_fast_dev.hpp"
#include "sd_fast_fill.hpp"

namespace sd {

// ---------------------------------------------------------------------------------------------
// traceback with per-template recomputation
// ---------------------------------------------------------------------------------------------
#ifndef SD_TRACE_MINW
#define SD_TRACE_MINW 7   // VGPR budget 72: 62 registers at QK = 3 without spills -> 8 waves per SIMD
#endif
#ifndef SD_TRACE_ADAPT
#define SD_TRACE_ADAPT 1
#endif
// waves per traceback workgroup: the per-wave LDS (moves + table) is 4.5 KB up to QK = 8, 36 KB at QK = 32
constexpr int trace_waves_per_group(int QK) { return QK <= 8 ? 4 : 1; }

template <int QK>
__global__ __launch_bounds__(64 * trace_waves_per_group(QK), (QK <= 4 ? SD_TRACE_MINW : QK <= 8 ? 2 : 1)) void sd_fast_trace(
    const ChunkDesc* __restrict__ chunks, int n_chunks, const uint32_t* __restrict__ bases2,
    const uint32_t* __restrict__ nmask, const uint32_t* __restrict__ slot_of,
    const uint8_t* __restrict__ tcodes, const uint32_t* __restrict__ lane_consts,
    const int32_t* __restrict__ toff, const int32_t* __restrict__ tlen, ScoreArgs sc, int P,
    const int32_t* __restrict__ B, const int32_t* __restrict__ argV,
    const uint32_t* __restrict__ ckpt, const int32_t* __restrict__ ckbase,
    DevRec* __restrict__ recs, int32_t* __restrict__ rec_cnt, int* __restrict__ queue,
    const int* __restrict__ order, int ckf16, int bshift, int W, const uint16_t* __restrict__ klist,
    const uint16_t* __restrict__ kpos, const int32_t* __restrict__ nkept, int T,
    const uint32_t* __restrict__ lane_t) {   // compacted TILED chunks (sd_tiled_place): lane table; kpos = first lane of a template
    // QK = ceil(Lmax / 64) cells per lane: 1..8 for templates of up to 512 bp; 16 / 32 for the long ones (up to 2048 bp:
    // a handful of templates, each over up to 32 virtual lanes of the fill)
    constexpr int QP = QK <= 4 ? 4 : QK <= 8 ? 8 : QK <= 16 ? 16 : 32;   // cells per lane rounded up to whole 8-byte loads
    constexpr int NWV = trace_waves_per_group(QK);
    using pt_t = std::conditional_t<(QK <= 4), uint8_t, std::conditional_t<(QK <= 8), uint16_t,
                 std::conditional_t<(QK <= 16), uint32_t, unsigned long long>>>;
    __shared__ pt_t pt_all[NWV][FAST_R][64];      // 2-bit moves of the lane's cells
    __shared__ int16_t mt_all[NWV][5][64][QP];    // (mm - del) of the lane's cells for the 5 read symbols
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    pt_t(*pt)[64] = pt_all[wave];
    int16_t(*mt)[64][QP] = mt_all[wave];
    ChunkSched sched;
    sched.init(queue, order, n_chunks);
    for (int c = sched.next(); c >= 0; c = sched.next()) {
    const ChunkDesc cd = chunks[c];
    const int n = cd.n;
    ReadCursor rc{bases2 + cd.woff, cd.noff >= 0 ? nmask + cd.noff : nullptr};
    const int32_t* BVc = B + cd.row0 + (uint64_t)c;  // (B << 7) | arg-max virtual lane
    const int bmask = (1 << bshift) - 1;   // arg-max field: (wave << 7) | virtual lane
    auto Bof = [&](int r) { return BVc[r] >> bshift; };
    auto Vof = [&](int r) { return BVc[r] & bmask; };
    DevRec* out = recs + cd.row0;
    const int ins = sc.ins, del = sc.del;
    const int mD = sc.match - sc.del, xD = sc.mismatch - sc.del;

    // --ed_thr with more than 128 templates: a chunk filled in the compacted form (sd_fast_wn_ck.hip) has its kept
    // templates, in their filtered order, in the virtual lanes of its first ceil(kept / 128) waves
    // (lane_t with T < 0: FastPlan::filter_only -- every chunk is compacted, up to all W waves; a chunk whose kept
    // templates did not fit was not filled, count -1: no records, the host repeats the batch)
    if (klist != nullptr && nkept[c] < 0) {
        if (lane == 0) rec_cnt[c] = 0;
        continue;
    }
    const bool cmp = klist != nullptr && (T < 0 || nkept[c] <= 128 * (W - 1));
    const int Tk = T < 0 ? -T : T;
    const uint16_t* klc = cmp ? klist + (size_t)c * (size_t)Tk : nullptr;
    const uint16_t* kpc = cmp ? kpos + (size_t)c * (size_t)Tk : nullptr;
    const uint32_t* ltc = (cmp && lane_t) ? lane_t + (size_t)c * (size_t)(W * 128) : nullptr;
    auto tmpl_of = [&](int v) {
        if (ltc) return (int)(ltc[v] & 0xffffu);   // tiled: any lane of the template names it
        if (cmp) return (int)klc[v];   // v = (wave << 7) | virtual lane = the place in the filtered order
        const uint32_t t = lane_consts[(((v >> 7) << 6) | (v & 63)) * FAST_LANE_WORDS + FLC_TMPL];
        return (int)(((v >> 6) & 1) ? (t >> 16) : (t & 0xffffu));
    };

    int cnt = 0;
    int e = n - 1;
    int j = tmpl_of(Vof(n));
    while (true) {
        const int Lj = tlen[j];
        const int x0 = toff[j];
        struct Cells { int code[QK]; int slot[QK]; };   // of the lane's cells, passed by value: the
        struct Pos { int i, k; bool done; };             // lambdas must not capture mutable state
        Cells cells;
        int i = e, k = Lj - 1;
        bool stop_row0 = false;
        // The path only ever moves towards smaller k, so a block entered at cell k needs the cells
        // 0..k and nothing beyond: the lanes are re-dealt with fewer cells each (QQ = QK, ~QK/2, ~QK/4)
        // as the walk approaches the start of the template -- the recurrence of a cell never looks at
        // larger k, so this is exact.  Lane l owns the cells [l*QQ, l*QQ + QQ).
        auto setup = [&](auto qq_c) -> Cells {
            constexpr int QQ = decltype(qq_c)::value;
            Cells cl;
#pragma unroll
            for (int q = 0; q < QK; ++q) { cl.code[q] = 7; cl.slot[q] = 0; }
#pragma unroll
            for (int q = 0; q < QQ; ++q) {
                const int kk = lane * QQ + q;
                cl.code[q] = kk < Lj ? (int)tcodes[x0 + kk] : 7;
                // compacted chunk: one lane per template, slot = cell -- or, tiled, lane = first lane + cell / P
                const int gl = ltc ? (int)kpc[j] + kk / P : (cmp ? (int)kpc[j] : 0);
                cl.slot[q] = kk < Lj ? (cmp ? (((gl >> 7) << 16) | ((ltc ? kk % P : kk) << 7) | (gl & 127))
                                                 : (int)slot_of[x0 + kk]) : 0;
#pragma unroll
                for (int b = 0; b < 5; ++b) mt[b][lane][q] = (int16_t)(4 * ((cl.code[q] == b ? mD : xD) - ins) - 1);
            }
            return cl;
        };
        // one block of rows [a, i]: recompute, derive the moves, walk; returns true when the instance
        // start (START move) was reached
        // Cells are kept as T = 4*E + 3, and every candidate of the cell update carries the reference's
        // traceback priority in its two low bits -- DEL 3 > INS 2 > DIAG 1 > START 0 (main.cpp:242-253) -- so
        // that ONE maximum yields both the value and, among equal values, the move the reference's equality
        // tests would pick first; no compares.  Rows are kept in the fill's row-shifted domain E' = E - i*ins, where
        // the insertion move is "keep", and a finished row is stored with the INSERTION tag, T = 4*E' + 2, so that
        // the insertion candidate of the next row is the stored value itself.  With Bd1 = 4*(B_i + del - (i-1)*ins) + 1
        // and mm4 = 4*(mm - del - ins) - 1:
        //     diag / start:  max(T[k-1], Bd1) + mm4   -> 4*(E'[k-1] + mm - del - ins) + 1  or  4*(start term) + 0
        //     insertion:     T[k]                     -> 4*E'[k] + 2
        //     deletion:      (left cell's result) | 3
        auto block = [&](auto qq_c, const Cells cl, const int i_in, const int k_in) -> Pos {
            constexpr int QQ = decltype(qq_c)::value;
            constexpr int QQP = QQ <= 4 ? 4 : QQ <= 8 ? 8 : QQ <= 16 ? 16 : 32;
            int i = i_in, k = k_in;
            const int* code = cl.code;
            const int* slot = cl.slot;
            const int a = i & ~(FAST_R - 1);  // first row of the block
            int32_t T[QQ];
            int rstart;
            if (a == 0) {
                // row 0, main.cpp:171-182 (moves: DEL where the cell equals its left neighbour, else stop)
                const int r = rc.code(0);
                int32_t run = NEG_INF32;
                int32_t loc[QQ];
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    const int kk = lane * QQ + q;
                    const int32_t mmd = code[q] == r ? mD : xD;
                    const int32_t cand = kk == 0 ? mmd + del : mmd;
                    run = kk == 0 ? cand : max(run, cand);
                    loc[q] = run;
                }
                const int32_t X = lane_up_neg(wave_prefix_max(run));
                int32_t left = X;
                unsigned long long bits = 0;
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    const int32_t Ef = max(loc[q], X);
                    const int tg = Ef == left ? 3 : 0;  // k == 0: left = -inf
                    bits |= (unsigned long long)tg << (2 * q);
                    left = Ef;
                    T[q] = 4 * Ef + 2;
                }
                pt[0][lane] = (pt_t)bits;
                rstart = 1;
            } else {
                const int q0 = a / FAST_R - 1;
                const int32_t cb = ckbase[cd.pad + q0];
                const uint32_t* ckq = ckpt + ((uint64_t)cd.pad + q0) * (uint64_t)W * (uint64_t)(P * 64);
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    const int v = slot[q] & 127, s = (slot[q] >> 7) & 511, wv_ = slot[q] >> 16;
                    const uint32_t wv = ckq[(wv_ * P + s) * 64 + (v & 63)];
                    const uint32_t hw = (v >> 6) ? (wv >> 16) : (wv & 0xffffu);
                    // checkpoint cell formats (FastPlan): 0 int16, 1 fp16, 2 biased u16 (0 = "-inf")
                    const int32_t Ev = ckf16 == 2 ? (hw ? cb + (int)hw - U16_BIAS : -0x10000000)
                                       : cb + (ckf16 ? (int)(float)__builtin_bit_cast(_Float16, (unsigned short)hw)
                                                     : (int)(short)hw);
                    // row a-1, shifted by (a-1)*ins; padding cells hold the fill's "-inf": keep them far below, no overflow
                    T[q] = 4 * (max(Ev, -0x08000000) - (a - 1) * ins) + 2;
                }
                rstart = a;
            }
            // per-row scalars of the block, one row per lane (read once, then v_readlane per row)
            const int rl = a + lane;
            const bool rvalid = rl >= 1 && rl <= i;
            // both in one word -- (start term << 3) | read symbol -- so that a row costs ONE v_readlane (the split is
            // scalar work); |start term| < 2^26 for every scoring the fast family takes
            const int vBR = rvalid ? (((4 * (Bof(rl) + del - (rl - 1) * ins) + 1) << 3) | rc.code(rl)) : 0;
            for (int r_i = rstart; r_i <= i; ++r_i) {
                const int br = __builtin_amdgcn_readlane(vBR, r_i - a);
                const int r = br & 7;
                const int32_t Bd2 = br >> 3;
                const int32_t pdEdge = lane_up_min(T[QQ - 1]);   // lane 0: identity of the max with Bd2 below
                int32_t loc[QQ];
                int32_t pd = pdEdge;
                int32_t mm4[QQP];
#pragma unroll
                for (int h = 0; h < QQP / 4; ++h) {
                    const uint2 mrow = *reinterpret_cast<const uint2*>(&mt[r][lane][4 * h]);  // 4 x int16
                    mm4[4 * h + 0] = (int)(short)(mrow.x & 0xffffu); mm4[4 * h + 1] = (int)mrow.x >> 16;
                    mm4[4 * h + 2] = (int)(short)(mrow.y & 0xffffu); mm4[4 * h + 3] = (int)mrow.y >> 16;
                }
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    // diag (tag 1) / start (tag 0).  q == 0: pd comes from the lane below through DPP; with the start term
                    // in a VGPR the move and the max are ONE v_max_i32_dpp (lane 0 keeps the start term: max(identity, .))
                    int32_t Bq = Bd2;
                    if (q == 0) asm volatile("" : "+v"(Bq));
                    const int32_t v = max(pd, Bq) + mm4[q];
                    const int32_t w = T[q];                    // insertion (tag 2): the stored value itself
                    if (q == 0) {
                        // k == 0 (lane 0): the fill never takes the insertion there, but the reference's
                        // traceback tests it (main.cpp:245): the value stays v; the move is INS iff w == v
                        const int32_t c = max(v, w);
                        loc[0] = (lane == 0 && w > (v | 3)) ? v : c;
                    } else {
                        loc[q] = max(max(loc[q - 1] | 3, v), w);
                    }
                    pd = T[q];
                }
                const int32_t X = lane_up_min(wave_prefix_max(loc[QQ - 1] | 3));   // only ever an operand of max
                uint32_t bits = 0, bitsHi = 0;   // the tags of up to 16 / of the cells 16.. of the lane
#pragma unroll
                for (int q = 0; q < QQ; ++q) {
                    const int32_t Ef = max(loc[q], X);
                    if (q < 16) bits = __builtin_amdgcn_alignbit((uint32_t)Ef, bits, 2);   // tag into the top, oldest cell lowest
                    else bitsHi = __builtin_amdgcn_alignbit((uint32_t)Ef, bitsHi, 2);
                    asm("v_and_or_b32 %0, %1, -4, 2" : "=v"(T[q]) : "v"(Ef));   // (Ef & ~3) | 2 in one op
                }
                if constexpr (QQ <= 16) {
                    pt[r_i - a][lane] = (pt_t)(bits >> (32 - 2 * QQ));
                } else {
                    pt[r_i - a][lane] = (pt_t)((unsigned long long)bits | ((unsigned long long)(bitsHi >> (64 - 2 * QQ)) << 32));
                }
            }
            // walk inside the block (wave-uniform).  Lane l looks at the cell l diagonal moves ahead, (i - l, k - l): the
            // run of DIAG tags from lane 0 on is taken in one LDS round trip (reads of ~90 % identity are mostly diagonal
            // runs), then the move that ends it.
            while (i >= a) {
                const int il = i - lane, kl = k - lane;
                const bool ok = il >= a && kl >= 0;
                int tg = 0;
                if (ok) tg = (int)((pt[il - a][kl / QQ] >> (2 * (kl % QQ))) & 3);
                const unsigned long long dg = __ballot(ok && tg == 1);
                const int run = dg == ~0ull ? 64 : __builtin_ctzll(~dg);
                i -= run;
                k -= run;
                if (run == 64 || i < a) continue;
                // (k >= 0 here: the k = 0 cell has no diagonal candidate, its tag is never DIAG)
                const int te = __builtin_amdgcn_readlane(tg, run);
                if (te == 3) { --k; }
                else if (te == 2) { --i; }
                else { return Pos{i, k, true}; }
            }
            return Pos{i, k, false};
        };
        constexpr int Q1 = (QK + 1) / 2, Q2 = (QK + 3) / 4;
        int cur = 0;
        while (true) {
            const int need = (k >> 6) + 1;
            const int lvl = (SD_TRACE_ADAPT && need <= Q2) ? Q2 : (SD_TRACE_ADAPT && need <= Q1) ? Q1 : QK;
            Pos ps;
            if (lvl == QK) {
                if (cur != QK) { cells = setup(std::integral_constant<int, QK>()); cur = QK; }
                ps = block(std::integral_constant<int, QK>(), cells, i, k);
            } else if (lvl == Q1) {
                if (cur != Q1) { cells = setup(std::integral_constant<int, Q1>()); cur = Q1; }
                ps = block(std::integral_constant<int, Q1>(), cells, i, k);
            } else {
                if (cur != Q2) { cells = setup(std::integral_constant<int, Q2>()); cur = Q2; }
                ps = block(std::integral_constant<int, Q2>(), cells, i, k);
            }
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

// ---------------------------------------------------------------------------------------------
// host: layout plan
// ---------------------------------------------------------------------------------------------
static int code_of(char ch) {
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

bool fast_plan_build(const std::vector<std::string>& tseq, ScoreArgs sc, int max_rows,
                     FastPlan& plan, std::string& why, bool allow_f16, bool allow_tr2, bool filter_only_ok, bool allow_u16) {
    (void)max_rows;
    plan = FastPlan();
    const int T = (int)tseq.size();
    int Lmax = 0, Lmin = INT_MAX;
    for (const std::string& s : tseq) {
        Lmax = std::max(Lmax, (int)s.size());
        Lmin = std::min(Lmin, (int)s.size());
    }
    if (T <= 0) { why = "no templates"; return false; }
    // A 1-bp template: its one cell is a k = 0 cell (start term only, no insertion move, main.cpp:188-193), whose stored
    // value may FALL from one row to the next, so the pad slots behind it -- which keep their old value like any cell
    // with an insertion move -- cannot stand for it at the lane's last slot.  The narrow and the tiled fills take such a
    // lane's end from slot 0 (FLC_ONE); the plain wide layouts do not know the form (such sets take the tiled one).
    const bool has1 = Lmin < 2;
    // Gap scores.  A positive INSERTION score is taken since round 6: nothing in the kernels depends on its sign -- the stored
    // domain S = E - base - tp*ins makes the insertion move "keep" whatever ins is, so lane totals still never decrease
    // between rebases (what the lazy carry and the range guard rely on), and the range bound below charges B's growth per
    // row with max(0, smax - del, ins).  A positive DELETION score is not: deleting a whole template then GAINS
    // (L - 1) * del, so B can grow by that much in ONE row (start a template, delete all of it: main.cpp:187-207 allows it)
    // -- two orders of magnitude beyond what 16-bit cells hold over a 128-row period; such scorings keep the generic family.
    if (sc.del > 0) { why = "positive deletion score"; return false; }
    if (T > 65535) { why = "too many templates"; return false; }
    auto ab = [](int v) { return v < 0 ? -v : v; };
    const int maxabs = std::max(std::max(ab(sc.ins), ab(sc.del)), std::max(ab(sc.mismatch), ab(sc.match)));
    // Range of a stored cell S = E - base - tp*ins between two rebases (tp <= FAST_REBASE rows, base = B at the
    // last rebase; base = 0 before the first one).  From any cell (i, j, k) the rest of the template can be
    // deleted, so D[i][j][k] <= D[i][j][L-1] + (L-1-k)*|del| <= B_{i+1} + (L-1-k)*|del|, i.e.
    //     E[i][x] = D + k*|del| <= B_{i+1} + (Lmax-1)*|del|                      (whatever the scores);
    // B grows by at most G = max(0, smax - del) per row (turn the last aligned row of the best path into a
    // deletion) and falls by at most |ins| per row, and before the first rebase B_1 >= -(Lmax*|del| + max|score|).
    //     |S| <= (Lmax-1)*|del| + (REBASE+1)*G + REBASE*|ins|  (+ a few single scores for the intermediate sums)
    // (Round 1 charged (REBASE+1)*max(match, mismatch, 0) for the growth of B instead of (REBASE+1)*G: with a
    // large |del|, long templates and non-positive match scores -- e.g. 0,-4,-4,-1 on 480-bp monomers -- the
    // fp16 cells left the exact-integer range; found by tools/fuzz_gpu.py seed 906.)
    const int smax = std::max(sc.match, sc.mismatch);
    const int64_t G = std::max(std::max(0, smax - sc.del), sc.ins);   // (an insertion at the end cell: B_{i+1} >= B_i + ins, and no more through it)
    auto ub_of = [&](int R) {
        return (int64_t)(Lmax - 1) * ab(sc.del) + (int64_t)(R + 1) * G + (int64_t)R * ab(sc.ins) + 8 * (int64_t)maxabs + 8;
    };
    // The rebase period is 128 rows; where only a shorter one keeps the cells inside the exact-integer range of fp16
    // (long templates, large gap scores) the plan takes 64 -- P + 6 packed ops per period against the fourth op
    // per cell pair of the integer cells, and the multi-wave layouts have no integer form at all.
    // (Not 32: the fp16 fills turn the row maxima into B words 64 rows at a time with the base of that moment.)
    int rebase = FAST_REBASE;
    if (ub_of(rebase) > 2040 && allow_f16 && ub_of(64) <= 2040) rebase = 64;
    const int64_t ub = ub_of(rebase);
    // (the biased-u16 cells of the narrow layout reach a little further than saturating int16: checked again once the layout is known)
    const int64_t u16_room = 15800 - (int64_t)(Lmax + 1) * ab(sc.del) - 8 * (int64_t)maxabs;
    const bool u16_fits = allow_f16 && allow_u16 && ub_of(FAST_REBASE) <= u16_room;
    if (ub > 12000 && !u16_fits) { why = "scores too large for int16 cells"; return false; }
    plan.range_bound = (int)ub;
    plan.rebase = rebase;
    if (Lmax > 64 * 32) { why = "template longer than 2048 bp"; return false; }

    // narrow layout: the smallest slot counts P whose lanes fit the two planes; the first three are candidates,
    // the one with the fewest cell ops per row wins (2P + the FL level its lanes allow, see below)
    int P = 0, split = 0;
    std::vector<std::pair<int, int>> cand;   // (P, split)
    for (int p : FAST_P_LIST) {
        int used = 0, s = 0;
        while (s < T && used + ((int)tseq[s].size() + p - 1) / p <= 64) {
            used += ((int)tseq[s].size() + p - 1) / p;
            ++s;
        }
        int used2 = 0;
        for (int j = s; j < T; ++j) used2 += ((int)tseq[j].size() + p - 1) / p;
        if (used2 <= 64) {
            cand.emplace_back(p, s);
            if (cand.size() == 3) break;
        }
    }
    if (!cand.empty()) { P = cand[0].first; split = cand[0].second; }
    bool wide = false;
    std::vector<int> tiled_v0;   // tiled multi-wave layout: first global virtual lane of each template
    if (P == 0) {
        // wide layout: one template per virtual lane
        if (T <= 1024 && !has1)   // (a 1-bp template ends its lane at slot 0: the narrow and the tiled fills know that form)
            for (int p : FAST_WIDE_P_LIST)
                if (p >= Lmax) { P = p; break; }
        if (P != 0) {
            // (rounds 1-4 also refused (3 Lmax + 2 REBASE + 4) * max|score| > 8000 here, a bound from before the range proof
            // above: `ub` <= 12 000 covers the wide layouts' stored cells as it covers the narrow ones -- same recurrence, same
            // stored domain -- and the int8 table values are checked below)
            split = std::min(T, 64);
            if (T > 128) plan.waves = (T + 127) / 128;   // multi-wave wide layout: wave w holds templates [128 w, 128 w + 128)
        } else {
            // tiled multi-wave layout (sd_fast_wt.hip): a template over ceil(L / P) consecutive virtual lanes of one
            // plane of one wave, templates in file order (so that "smallest wave, then smallest virtual lane" among
            // equal ends is the reference's first template, main.cpp:184-186); the slot count with the least SIMD time
            // per row among those that fit eight waves and the CU's LDS (P * 128 bytes of codes per wave): a wave
            // spends 4.5 P + ~60 instructions on a row, W waves per chunk, and how well that runs depends on the waves
            // a CU holds (two per SIMD at the kernel's register count; five to seven leave SIMDs half empty or, in one
            // lock-stepped workgroup, uneven) -- efficiencies measured on three sets (tools/scratch/tiled_p.py,
            // profiles/r04_tiled_layout.txt: the order of the five slot counts follows this figure)
            int bestW = 0;
            int64_t best_cost = 0;
            const int force_p = getenv("SD_TILED_P") ? atoi(getenv("SD_TILED_P")) : 0;   // developer knob
            for (int p : FAST_TILED_P_LIST) {
                if (force_p && p != force_p) continue;
                std::vector<int> v0((size_t)T, 0);
                int cur = 0;
                for (int j = 0; j < T; ++j) {
                    const int V = ((int)tseq[(size_t)j].size() + p - 1) / p;
                    if (V > 64 - (cur & 63)) cur = (cur + 63) & ~63;
                    v0[(size_t)j] = cur;
                    cur += V;
                }
                const int Wp = (cur + 127) / 128;
                if (Wp > 8 || (size_t)Wp * (size_t)(p / 16) * 2048 + 256 > (size_t)160 * 1024) continue;
                const size_t ldsb = (size_t)Wp * (size_t)(p / 16) * 2048 + 256;
                const int per_cu = std::max(1, std::min(8 / Wp, (int)((size_t)160 * 1024 / ldsb)));
                static const int eff_pct[9] = {0, 21, 43, 64, 85, 72, 80, 87, 100};   // by waves per CU
                const int64_t cost = (int64_t)Wp * (9 * p / 2 + 60) * 100 / eff_pct[std::min(8, per_cu * Wp)];
                if (P == 0 || cost < best_cost) { P = p; bestW = Wp; best_cost = cost; tiled_v0.swap(v0); }
            }
            if (P == 0 && filter_only_ok) {
                // --ed_thr: the set as a whole does not fit eight waves, a chunk's KEPT templates mostly do (the reason the
                // reference has the prefilter, main.cpp:128-149).  No layout of the whole set: every chunk is filled in the
                // compacted form; slot count by lanes x instructions per lane over the set, among those whose eight waves
                // fit the LDS.
                int64_t bc = 0;
                for (int p : {96, 128}) {
                    int64_t cst = 0;
                    for (int j = 0; j < T; ++j) cst += (int64_t)(((int)tseq[(size_t)j].size() + p - 1) / p) * (9 * p / 2 + 60);
                    if (P == 0 || cst < bc) { P = p; bc = cst; }
                }
                bestW = 8;
                tiled_v0.assign((size_t)T, 0);
                plan.filter_only = true;
            }
            if (P == 0) { why = T > 1024 ? "more than 1024 templates, too many cells for the tiled layout" : "template set too large for the tiled layout"; return false; }
            plan.waves = bestW;
            plan.tiled = true;
            split = 0;
        }
        wide = true;
    }
    const int W = plan.waves;

    plan.wide = wide;
    {
        // Narrow layout, round 6: biased unsigned 16-bit cells (CellOps<CF_U16>): U = S + 0x3E00 must stay inside
        // 0..0x7BFF together with every intermediate sum -- the row maximum takes the cells with up to (Lmax + 1) * |del|
        // taken off (end offsets), a table value or a few single scores ride on top -- so the guard window is what is
        // left of +-15 800 after those, and the proven bound of the full 128-row period must fit it.  Where it does the
        // format replaces both fp16 (+-2040) and int16 cells; the rebase period goes back to 128 rows.
        const int64_t room = u16_room;
        const int64_t ub128 = ub_of(FAST_REBASE);
        plan.u16 = !wide && u16_fits;
        if (!plan.u16 && ub > 12000) { why = "scores too large for int16 cells"; return false; }
        if (plan.u16) {
            plan.u16_lim = (int)room;
            plan.rebase = FAST_REBASE;
            plan.range_bound = (int)ub128;
        }
        // fp16 cells are exact while every value stays an integer below 2048 in magnitude
        plan.f16 = !wide && !plan.u16 && ub <= 2040 && allow_f16;   // (SD_FLAG_NO_F16 / SD_FILL_CELLS=i16 arrive as allow_f16 = false)
    }
    plan.P = P;
    plan.P4 = (P + 3) & ~3;
    // Lanes of a template: V = ceil(L / P) virtual lanes of at most P cells each.  A lane may hold fewer cells
    // than P (its tail slots are transparent pads, as the last lane of a template always had), so the first cell
    // of every lane but the first is a choice: it is taken where the lane meets all four bases early, because the
    // last slot whose diagonal input needs the floor (see the slot loop of sd_fast_fill) -- per lane and read
    // symbol, the slots q >= 1 where the table value exceeds every earlier one of the lane; slot 0 always takes
    // the start term / the carry, its candidate KB + tbl[0] is in the chain from there on -- decides how many
    // slots keep the three-op form (FastPlan::floor_slots).  A small DP per template minimises the latest such
    // slot over its lanes; the first lane starts at cell 0 whatever it costs.
    auto lane_record = [&](const std::string& s, int b0, int b1) {
        int rec = 0;
        for (int b = 0; b < 5; ++b) {
            int run = code_of(s[(size_t)b0]) == b ? sc.match : sc.mismatch;
            for (int q = 1; b0 + q < b1; ++q) {
                const int val = code_of(s[(size_t)(b0 + q)]) == b ? sc.match : sc.mismatch;
                if (val > run) { run = val; rec = std::max(rec, q); }
            }
        }
        return rec;
    };
    using Bounds = std::vector<std::vector<int>>;   // [j][u] = first cell of lane u of template j; [j][V] = L
    auto layout_for = [&](int P, Bounds& bnd) {
        int fs = 1;
        bnd.assign((size_t)T, std::vector<int>());
        for (int j = 0; j < T; ++j) {
            const std::string& s = tseq[(size_t)j];
            const int L = (int)s.size(), V = (L + P - 1) / P;
            std::vector<int>& bj = bnd[(size_t)j];
            bj.assign((size_t)V + 1, 0);
            for (int u = 0; u <= V; ++u) bj[(size_t)u] = std::min(L, u * P);
            if (!wide && V > 1 && !getenv("SD_PLAN_UNIFORM_LANES")) {
                // start of lane u in [lo(u), hi(u)]: the lanes before hold at most u*P cells, the lanes from u on
                // at most (V-u)*P
                // (the first lane keeps at least two cells: cell 0 has no insertion move, its value may fall from one
                // row to the next, and a pad behind it would keep the old, larger value -- the reason 1-bp templates
                // are not taken either)
                auto lo = [&](int u) { return std::max(u + 1, L - (V - u) * P); };
                auto hi = [&](int u) { return std::min(u * P, L - (V - u)); };
                const int INF = 1 << 30;
                std::vector<std::vector<int>> cost((size_t)V), from((size_t)V);
                cost[0].assign(1, 0);
                from[0].assign(1, 0);
                for (int u = 1; u < V; ++u) {
                    const int n = hi(u) - lo(u) + 1;
                    cost[(size_t)u].assign((size_t)n, INF);
                    from[(size_t)u].assign((size_t)n, -1);
                    for (int x = 0; x < n; ++x) {
                        const int st = lo(u) + x;
                        const int pl = u == 1 ? 0 : lo(u - 1), ph = u == 1 ? 0 : hi(u - 1);
                        for (int pv = ph; pv >= pl; --pv) {    // fullest previous lane first: ties keep the uniform layout
                            if (pv >= st || st - pv > P) continue;
                            const int c0 = cost[(size_t)u - 1][(size_t)(pv - pl)];
                            if (c0 >= INF) continue;
                            const int c = std::max(c0, lane_record(s, pv, st));
                            if (c < cost[(size_t)u][(size_t)x]) { cost[(size_t)u][(size_t)x] = c; from[(size_t)u][(size_t)x] = pv; }
                        }
                    }
                }
                int best = INF, bst = -1;
                for (int st = hi(V - 1); st >= lo(V - 1); --st) {
                    const int c0 = cost[(size_t)V - 1][(size_t)(st - lo(V - 1))];
                    if (c0 >= INF || L - st > P) continue;
                    const int c = std::max(c0, lane_record(s, st, L));
                    if (c < best) { best = c; bst = st; }
                }
                if (bst >= 0) {
                    int st = bst;
                    for (int u = V - 1; u >= 1; --u) {
                        bj[(size_t)u] = st;
                        st = from[(size_t)u][(size_t)(st - lo(u))];
                    }
                }
            }
            for (int u = 0; u < V; ++u)
                fs = std::max(fs, lane_record(s, bj[(size_t)u], bj[(size_t)u + 1]));
        }
        return fs;
    };
    // cell ops per row of a candidate: 2 per slot + 1 per slot that keeps the start-term maximum (the FL level
    // the launchers of sd_fast_fl*.hip would pick; 3 per slot where no variant exists)
    auto cell_ops = [](int P, int fs) {
        if (P >= 30 && P <= 40) { for (int c : {12, 16, 20, 24, 28}) if (fs <= c && c + 2 < P) return 2 * P + c; }
        else if (P > 40) { for (int c : {16, 24, 32}) if (fs <= c) return 2 * P + c; }
        return 3 * P;
    };
    Bounds bnd;
    plan.floor_slots = 1;
    if (wide || cand.size() <= 1 || getenv("SD_PLAN_UNIFORM_LANES")) {
        plan.floor_slots = layout_for(P, bnd);
    } else {
        int best = 1 << 30;
        for (const auto& c : cand) {
            Bounds b2;
            const int fs = layout_for(c.first, b2);
            const int ops = cell_ops(c.first, fs);
            if (ops < best) { best = ops; P = c.first; split = c.second; plan.floor_slots = fs; bnd.swap(b2); }
        }
        plan.P = P;
        plan.P4 = (P + 3) & ~3;
    }
    // the floor level every read symbol needs on its own (round 6): per lane and symbol b the last slot q >= 1 whose table
    // value exceeds every earlier one of the lane -- floor_slots is the maximum over the five symbols; the narrow u16 fills
    // pick one of three unrolled slot loops per ROW by the row's symbol (sd_fast_fill: FLS)
    for (int b = 0; b < 5; ++b) plan.floor_sym[b] = 1;
    plan.table_nonneg = std::min(sc.match, sc.mismatch) - sc.del - sc.ins >= 0;
    if (!plan.filter_only)
        for (int j = 0; j < T; ++j) {
            const std::string& sq = tseq[(size_t)j];
            const std::vector<int>& bj = bnd[(size_t)j];
            for (size_t u = 0; u + 1 < bj.size(); ++u)
                for (int b = 0; b < 5; ++b) {
                    int run = code_of(sq[(size_t)bj[u]]) == b ? sc.match : sc.mismatch;
                    for (int q = 1; bj[u] + q < bj[u + 1]; ++q) {
                        const int val = code_of(sq[(size_t)(bj[u] + q)]) == b ? sc.match : sc.mismatch;
                        if (val > run) { run = val; plan.floor_sym[b] = std::max(plan.floor_sym[b], q); }
                    }
                }
        }
    if (getenv("SD_PLAN_DEBUG"))
        std::fprintf(stderr, "[sd plan] P = %d: floor slots %d; per read symbol A C G T N: %d %d %d %d %d\n", P, plan.floor_slots,
                     plan.floor_sym[0], plan.floor_sym[1], plan.floor_sym[2], plan.floor_sym[3], plan.floor_sym[4]);
    plan.T = T;
    plan.split = split;
    plan.Lmax = Lmax;
    plan.Qk = (Lmax + 63) / 64;
    if (plan.Qk == 5) plan.Qk = 6;
    if (plan.Qk == 7) plan.Qk = 8;
    if (plan.Qk > 8) plan.Qk = plan.Qk <= 16 ? 16 : 32;
    plan.vlane0.assign((size_t)T, 0);
    int Vmax = 1;
    if (plan.tiled) {
        for (int j = 0; j < T; ++j) {
            plan.vlane0[(size_t)j] = tiled_v0[(size_t)j];
            Vmax = std::max(Vmax, ((int)tseq[(size_t)j].size() + P - 1) / P);
        }
        plan.bshift = 10;
    } else if (W > 1) {
        for (int j = 0; j < T; ++j) plan.vlane0[(size_t)j] = j;   // global virtual lane = (wave << 7) | (plane << 6) | lane
        plan.bshift = 10;
    } else {
        int v = 0;
        for (int j = 0; j < T; ++j) {
            if (j == split) v = 64;
            plan.vlane0[(size_t)j] = v;
            const int V = ((int)tseq[(size_t)j].size() + P - 1) / P;
            Vmax = std::max(Vmax, V);
            v += V;
        }
    }
    plan.H = Vmax - 1;
    plan.end_vlane.assign((size_t)T, 0);
    plan.end_off.assign((size_t)T, 0);
    for (int j = 0; j < T; ++j) {
        const int L = (int)tseq[(size_t)j].size();
        plan.end_vlane[(size_t)j] = plan.filter_only ? -1 : plan.vlane0[(size_t)j] + (L + P - 1) / P - 1;   // (no static lanes)
        plan.end_off[(size_t)j] = (L - 1) * sc.del;
    }

    // per virtual lane: owner template and index inside it
    std::vector<int> owner((size_t)128 * W, -1), uidx((size_t)128 * W, 0), nv((size_t)128 * W, 0);
    for (int j = 0; j < T; ++j) {
        const int V = ((int)tseq[(size_t)j].size() + P - 1) / P;
        for (int u = 0; u < V; ++u) {
            owner[(size_t)plan.vlane0[(size_t)j] + u] = j;
            uidx[(size_t)plan.vlane0[(size_t)j] + u] = u;
            nv[(size_t)plan.vlane0[(size_t)j] + u] = V;
        }
    }
    auto put = [](uint32_t& w, int plane, int val) {
        const uint32_t h = (uint32_t)val & 0xffffu;
        w = plane ? ((w & 0x0000ffffu) | (h << 16)) : ((w & 0xffff0000u) | h);
    };
    plan.lane_consts.assign((size_t)64 * W * FAST_LANE_WORDS, 0u);
    for (int v = 0; v < 128 * W; ++v) {
        const int plane = (v >> 6) & 1, lane = ((v >> 7) << 6) | (v & 63);   // lane index incl. the wave
        uint32_t* lc = &plan.lane_consts[(size_t)lane * FAST_LANE_WORDS];
        const int j = owner[(size_t)v];
        const bool start = j < 0 || uidx[(size_t)v] == 0;
        const bool last = j >= 0 && uidx[(size_t)v] == nv[(size_t)v] - 1;
        put(lc[FLC_STARTMASK], plane, start ? 0xffff : 0);
        put(lc[FLC_CONTMASK], plane, start ? 0 : 0xffff);
        put(lc[FLC_ENDOFF], plane, last ? ((int)tseq[(size_t)j].size() - 1) * sc.del : NEG16);
        put(lc[FLC_ROW0], plane, (j >= 0 && uidx[(size_t)v] == 0) ? sc.ins + sc.del : sc.ins);
        put(lc[FLC_TMPL], plane, j >= 0 ? j : 0xffff);
        put(lc[FLC_CONT2], plane, (j >= 0 && uidx[(size_t)v] >= 2) ? 0xffff : 0);
        put(lc[FLC_ENDALL], plane, j >= 0 ? ((int)tseq[(size_t)j].size() - 1) * sc.del : NEG16);
        put(lc[FLC_ONE], plane, (j >= 0 && tseq[(size_t)j].size() == 1) ? 0xffff : 0);
    }
    // The carry scan of the narrow fills can take its shifted operands from ds_bpermute (no VALU work, and a start lane
    // simply reads an idle lane's -inf instead of being masked) when both planes have the same segment structure --
    // every lane starts / continues a template in both planes alike -- and some lane is idle in both: Hx = H | 1 << 8 |
    // idle lane << 16 (C2: 24 templates of five lanes, 12 per plane).
    plan.Hx = plan.H;
    if (!wide && W == 1 && plan.H >= 1 && plan.H <= 4 && !getenv("SD_FILL_DPP_SCAN")) {
        bool same = true;
        int idle = -1;
        for (int l = 0; l < 64; ++l) {
            const uint32_t c1 = plan.lane_consts[(size_t)l * FAST_LANE_WORDS + FLC_CONTMASK];
            const uint32_t c2 = plan.lane_consts[(size_t)l * FAST_LANE_WORDS + FLC_CONT2];
            if ((c1 & 0xffffu) != (c1 >> 16) || (c2 & 0xffffu) != (c2 >> 16)) same = false;
            if (owner[(size_t)l] < 0 && owner[(size_t)64 + l] < 0) idle = l;
        }
        if (same && idle >= 0) plan.Hx = plan.H | (1 << 8) | (idle << 16) | (getenv("SD_FILL_BPERM_TAIL") ? 1 << 9 : 0);   // bit 9 (developer A/B): in the last round too
    }
    if (has1) plan.Hx |= 1 << 10;
    if (plan.rebase == 64) plan.Hx |= 1 << 11;
    if (wide) {
        // int8 table [5][G][2 halves][64 lanes][4 dwords]; dword d of half h of group g holds slots
        // 16g+8h+2d, +1 as bytes {lo plane, hi plane, lo plane, hi plane}; -128 = transparent padding
        const int G = P / 16;
        const int xd = sc.mismatch - sc.del - sc.ins, md = sc.match - sc.del - sc.ins;
        if (xd < -127 || xd > 127 || md < -127 || md > 127) { why = "scores too large for the int8 table"; return false; }
        // fp16 cells (4 ops per slot instead of 5): the table bytes are bf8 (E5M2: sign, 5 exponent and
        // 2 mantissa bits), which one gfx950 instruction expands to a packed fp16 pair; usable when both
        // table values are exactly representable (|v| = 1.mm * 2^e) and the score range fits fp16
        auto bf8_of = [](int v, bool& ok) -> int {
            if (v == 0) return 0;
            const int a = v < 0 ? -v : v;
            int e = 0;
            while ((a >> (e + 1)) != 0) ++e;
            const int m4 = (a << 2) >> e;   // 4 * a / 2^e, in [4, 8)
            if (((m4 << e) >> 2) != a || (e > 2 && (a & ((1 << (e - 2)) - 1)) != 0)) ok = false;
            return (v < 0 ? 0x80 : 0) | ((e + 15) << 2) | (m4 & 3);
        };
        bool bf_ok = true;
        const int xb = bf8_of(xd, bf_ok), mb = bf8_of(md, bf_ok);
        plan.f16 = bf_ok && ub <= 2040 && allow_f16;
        if (W > 1 || plan.tiled) {
            // multi-wave wide layout (sd_fast_wn.hip): LDS holds the template base codes, [wave][G][2 halves][64
            // lanes][4 dwords], bytes as above; code 7 = padding; the two bf8 table bytes travel as kernel arguments
            // fp16 cells / bf8 bytes where the set and scoring allow; else int16 cells / int8 bytes (sd_fast_wn_i16.hip, round
            // 5: 5.5 instead of 4.5 ops per slot -- until then such sets fell to the generic family).  The compacted
            // --ed_thr forms exist for fp16 cells only: a set that has no layout of its own needs them.
            if (!plan.f16 && plan.filter_only) { why = "a set beyond eight waves needs bf8-exact table values and the fp16 score range"; return false; }
            plan.bf8_match = (uint32_t)(plan.f16 ? mb : md) & 0xffu;
            plan.bf8_mismatch = (uint32_t)(plan.f16 ? xb : xd) & 0xffu;
            plan.table.assign((size_t)W * G * 512, 0x07070707u);
            int64_t sumLw = 0;
            for (const std::string& s : tseq) sumLw += (int64_t)s.size();
            plan.slot_of.assign((size_t)sumLw, 0);
            plan.tcodes.assign((size_t)sumLw, 0);
            int64_t xw = 0;
            for (int j = 0; j < T; ++j) {
                const std::string& s = tseq[(size_t)j];
                int u = 0;
                for (int k = 0; k < (int)s.size(); ++k, ++xw) {
                    // (one lane per template unless tiled: bnd[j] = {0, L})
                    while (k >= bnd[(size_t)j][(size_t)u + 1]) ++u;
                    const int v = plan.vlane0[(size_t)j] + u, slot = k - bnd[(size_t)j][(size_t)u];
                    const int wv = v >> 7, plane = (v >> 6) & 1, lane = v & 63;
                    const int cd = code_of(s[(size_t)k]);
                    plan.tcodes[(size_t)xw] = (uint8_t)cd;
                    plan.slot_of[(size_t)xw] = ((uint32_t)wv << 16) | ((uint32_t)slot << 7) | (uint32_t)(v & 127);
                    const int g = slot / 16, s16 = slot & 15, h = s16 >> 3, d = (s16 & 7) >> 1, odd = s16 & 1;
                    uint32_t& w = plan.table[((((size_t)wv * G + g) * 2 + h) * 64 + lane) * 4 + d];
                    const int sh = 16 * odd + 8 * plane;
                    w = (w & ~(0xffu << sh)) | ((uint32_t)cd << sh);
                }
            }
            plan.ok = true;
            return true;
        }
        plan.table.assign((size_t)5 * G * 512, plan.f16 ? 0xFCFCFCFCu : 0x80808080u);
        auto putb = [&](int grp, int v, int slot16, int val) {
            if (plan.f16) val = val == md ? mb : xb;
            const int plane = v >> 6, lane = v & 63, h = slot16 >> 3, d = (slot16 & 7) >> 1, odd = slot16 & 1;
            uint32_t& w = plan.table[(((size_t)grp * 2 + h) * 64 + lane) * 4 + d];
            const int sh = 16 * odd + 8 * plane;
            w = (w & ~(0xffu << sh)) | (((uint32_t)val & 0xffu) << sh);
        };
        int64_t sumLw = 0;
        for (const std::string& s : tseq) sumLw += (int64_t)s.size();
        plan.slot_of.assign((size_t)sumLw, 0);
        plan.tcodes.assign((size_t)sumLw, 0);
        int64_t xw = 0;
        for (int j = 0; j < T; ++j) {
            const std::string& s = tseq[(size_t)j];
            const int v = plan.vlane0[(size_t)j];
            for (int k = 0; k < (int)s.size(); ++k, ++xw) {
                const int cd = code_of(s[(size_t)k]);
                plan.tcodes[(size_t)xw] = (uint8_t)cd;
                plan.slot_of[(size_t)xw] = (uint32_t)((k << 7) | v);
                for (int b = 0; b < 5; ++b) putb(b * G + k / 16, v, k & 15, cd == b ? md : xd);
            }
        }
        if (allow_tr2) fast_plan_trace2(tseq, sc, plan);   // (one wave per chunk: the same checkpoint layout as the narrow fills)
        plan.ok = true;
        return true;
    }
    // LDS table image [5][P4/4][64][4]: (mm - del - ins) per template cell, NEG on padding / idle
    const int P4 = plan.P4;
    plan.table.assign((size_t)5 * P4 * 64, NEG2);
    int64_t sumL = 0;
    for (const std::string& s : tseq) sumL += (int64_t)s.size();
    plan.slot_of.assign((size_t)sumL, 0);
    plan.tcodes.assign((size_t)sumL, 0);
    int64_t x = 0;
    for (int j = 0; j < T; ++j) {
        const std::string& s = tseq[(size_t)j];
        int u = 0;
        for (int k = 0; k < (int)s.size(); ++k, ++x) {
            while (k >= bnd[(size_t)j][(size_t)u + 1]) ++u;
            const int v = plan.vlane0[(size_t)j] + u, slot = k - bnd[(size_t)j][(size_t)u];
            const int plane = v >> 6, lane = v & 63;
            const int cd = code_of(s[(size_t)k]);
            plan.tcodes[(size_t)x] = (uint8_t)cd;
            plan.slot_of[(size_t)x] = (uint32_t)((slot << 7) | v);
            for (int b = 0; b < 5; ++b) {
                const int val = (cd == b ? sc.match : sc.mismatch) - sc.del - sc.ins;
                uint32_t& w = plan.table[(((size_t)b * (P4 / 4) + slot / 4) * 64 + lane) * 4 + (slot & 3)];
                put(w, plane, val);
            }
        }
    }
    if (allow_tr2) fast_plan_trace2(tseq, sc, plan);
    plan.ok = true;
    return true;
}

int64_t fast_ckpt_rows_total(const FastPlan& plan, std::vector<ChunkDesc>& chunks) {
    (void)plan;
    int64_t tot = 0;
    for (ChunkDesc& cd : chunks) {
        cd.pad = (uint32_t)tot;
        tot += (cd.n - 1) / FAST_R;
    }
    return tot;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_fast_fill(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                      const uint32_t* bases2, const uint32_t* nmask, const uint32_t* table,
                      const uint32_t* lane_consts, ScoreArgs sc, int32_t* B, int32_t* argV,
                      uint32_t* ckpt, int32_t* ckbase, int* queue, const int* order, int n_cu,
                      const uint32_t* cendoff, const uint32_t* crank, size_t min_lds) {
    // 16 waves per CU either way (4 per SIMD): one workgroup of 16 when the launch fills the machine, else workgroups of 8
    int nw = n_chunks >= SD_FILL_NW_MAX * n_cu ? SD_FILL_NW_MAX : 8;
    if (const char* ev = getenv("SD_FILL_NW")) nw = atoi(ev) == 16 ? 16 : 8;   // developer knob
    const int NW = nw;
    int grid = std::min((n_chunks + NW - 1) / NW, (16 / NW) * n_cu);  // persistent
    if (const char* ev = getenv("SD_FILL_GRID")) grid = std::max(1, atoi(ev));   // developer knob
    // `queue` points at a zeroed work-queue head that no earlier launch has used (sd_engine hands out a fresh
    // one per run): no memset kernel sits between the launches of a stream
    if (plan.tiled) {
        launch_fast_fill_wt(plan, st, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt, ckbase, queue,
                            order, n_cu, cendoff, crank);
        return;
    }
    if (plan.wide && plan.waves > 1) {
        launch_fast_fill_wn(plan, st, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt, ckbase, queue,
                            order, n_cu, cendoff, crank);
        return;
    }
    if (plan.wide) {
        launch_fast_fill_wide(plan, st, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, ckpt,
                              ckbase, queue, order, n_cu, cendoff, crank);
        return;
    }
    // min_lds (pipeline mode 2): ask for at least this much, so that a third workgroup never fits a CU
    const size_t lds = std::max((size_t)5 * plan.P4 * 64 * sizeof(uint32_t) + 128, min_lds);   // + FairShare's words
    const bool ranked = cendoff != nullptr;
#define SD_FILL_K(PP, RK, HF)                                                                        \
    {                                                                                                \
        if (has1) {                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill<PP, RK, HF, PP, true>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            hipLaunchKernelGGL((sd_fast_fill<PP, RK, HF, PP, true>), dim3(grid), dim3(nw * 64), lds, st, chunks, \
                               n_chunks, bases2, nmask, table, lane_consts, sc, plan.Hx, B, argV, ckpt, \
                               ckbase, queue, order, cendoff, crank);                                \
        } else {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_fast_fill<PP, RK, HF>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            hipLaunchKernelGGL((sd_fast_fill<PP, RK, HF>), dim3(grid), dim3(nw * 64), lds, st, chunks, \
                               n_chunks, bases2, nmask, table, lane_consts, sc, plan.Hx, B, argV, ckpt, \
                               ckbase, queue, order, cendoff, crank);                                \
        }                                                                                            \
    }
#define SD_FILL(PP)                                                                                  \
    case PP:                                                                                         \
        if (plan.f16) {                                                                              \
            if (ranked) SD_FILL_K(PP, true, true) else SD_FILL_K(PP, false, true)                    \
        } else {                                                                                     \
            if (ranked) SD_FILL_K(PP, true, false) else SD_FILL_K(PP, false, false)                  \
        }                                                                                            \
        break;
    // fp16 cells, 150-200 bp monomers: the variants that skip the dominated start-term maxima (sd_fast_fl.hip);
    // FastPlan::full_floor (SD_FLAG_FULL_FLOOR) keeps the full kernel (developer A/B and the parity test of the two)
    const bool has1 = ((plan.Hx >> 10) & 1) != 0;   // 1-bp templates: the full-floor kernels carry the FLC_ONE form
    if (!plan.full_floor && !has1 &&
        (plan.u16 ? (launch_fast_fill_fl_u16(plan, st, grid, nw, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B,
                                             argV, ckpt, ckbase, queue, order, cendoff, crank) ||
                     launch_fast_fill_fl_u16s(plan, st, grid, nw, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B,
                                              argV, ckpt, ckbase, queue, order, cendoff, crank))
         : plan.f16 ? launch_fast_fill_fl(plan, st, grid, nw, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B,
                                          argV, ckpt, ckbase, queue, order, cendoff, crank)
                    : launch_fast_fill_fl_i16(plan, st, grid, nw, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc,
                                              B, argV, ckpt, ckbase, queue, order, cendoff, crank)))
        return;
    if (plan.u16) {   // the full-floor kernels of the biased-u16 cell format live in their own unit (sd_fast_u16.hip)
        launch_fast_fill_full_u16(plan, st, grid, nw, lds, chunks, n_chunks, bases2, nmask, table, lane_consts, sc, B, argV, ckpt,
                                  ckbase, queue, order, cendoff, crank);
        return;
    }
    switch (plan.P) {
        SD_FILL(4) SD_FILL(8) SD_FILL(12) SD_FILL(16) SD_FILL(20) SD_FILL(24) SD_FILL(28) SD_FILL(30)
        SD_FILL(31) SD_FILL(32) SD_FILL(33) SD_FILL(34) SD_FILL(35) SD_FILL(36) SD_FILL(37) SD_FILL(38)
        SD_FILL(39) SD_FILL(40) SD_FILL(42) SD_FILL(44) SD_FILL(46) SD_FILL(48) SD_FILL(52) SD_FILL(56)
        SD_FILL(60) SD_FILL(64)
        default: break;
    }
#undef SD_FILL
#undef SD_FILL_K
}

void launch_fast_trace(const FastPlan& plan, hipStream_t st, const ChunkDesc* chunks, int n_chunks,
                       const uint32_t* bases2, const uint32_t* nmask, const uint32_t* slot_of,
                       const uint8_t* tcodes, const uint32_t* lane_consts, const int32_t* toff,
                       const int32_t* tlen, ScoreArgs sc, const int32_t* B, const int32_t* argV,
                       const uint32_t* ckpt, const int32_t* ckbase, DevRec* recs,
                       int32_t* rec_cnt, int* queue, const int* order, int n_cu, const uint16_t* klist,
                       const uint16_t* kpos, const int32_t* nkept, const uint32_t* tr2_tab, const uint32_t* lane_t) {
    // the packed two-block form where the plan has it (one wave per chunk in the fill, templates <= 256 bp, 16-bit tagged range)
    if (klist == nullptr && launch_fast_trace2(plan, st, chunks, n_chunks, bases2, nmask, lane_consts, tcodes, toff, tlen, sc,
                                               B, ckpt, ckbase, tr2_tab, recs, rec_cnt, queue, order, n_cu))
        return;
    int bpc = 8;
    if (const char* ev = getenv("SD_TRACE_BPC")) bpc = std::max(1, atoi(ev));  // developer knob
    int grid = std::min((n_chunks + 3) / 4, bpc * n_cu);  // persistent: 8 workgroups of 4 waves per CU
    // long templates: one wave per workgroup; 16 cells per lane: 239 registers and 18 KB of LDS, two waves per SIMD (5.2 ->
    // 3.4 ms on five 1-kb monomers); 32 cells per lane: 412 registers, one
    if (plan.Qk > 8) grid = std::min(n_chunks, (plan.Qk == 16 ? 8 : 4) * n_cu);
    if (const char* ev = getenv("SD_TRACE_GRID")) grid = std::max(1, atoi(ev));   // developer knob
#define SD_TRACE(QQ)                                                                              \
    hipLaunchKernelGGL(sd_fast_trace<QQ>, dim3(grid), dim3(64 * trace_waves_per_group(QQ)), 0, st, chunks, n_chunks, bases2, \
                       nmask, slot_of, tcodes, lane_consts, toff, tlen, sc, plan.P, B, argV, ckpt, \
                       ckbase, recs, rec_cnt, queue, order, plan.u16 ? 2 : plan.f16 ? 1 : 0, plan.bshift, plan.waves, klist, kpos, \
                       nkept, plan.filter_only ? -plan.T : plan.T, lane_t)
    switch (plan.Qk) {
        case 1: SD_TRACE(1); break;
        case 2: SD_TRACE(2); break;
        case 3: SD_TRACE(3); break;
        case 4: SD_TRACE(4); break;
        case 6: SD_TRACE(6); break;
        case 8: SD_TRACE(8); break;
        case 16: SD_TRACE(16); break;
        default: SD_TRACE(32); break;
    }
#undef SD_TRACE
}

}  // namespace sd
