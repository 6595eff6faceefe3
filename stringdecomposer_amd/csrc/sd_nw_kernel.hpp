// sd_nw_kernel.hpp -- the identity kernel (device side), shared by sd_nw.hip (segments of an ASCII text) and
// sd_ident.hip (in-stream: the records of a device batch against the 2-bit reads that are already resident).
//
// What it computes (stringdecomposer/main.py:29-60, `edist` + `aai` through python-edlib): for a pair (read
// segment q, template t) the unit-cost global alignment that edlib's traceback picks -- priority
// up ('I', consume a query symbol) > left ('D', consume a target symbol) > diagonal, walking from the
// bottom-right corner (edlib.cpp:945-1150) -- and of it only two numbers: the edit distance and the number of
// '=' columns (matches = |q| - dist + #left moves); identity = matches / (dist + matches) * 100.
//
// One lane per pair, Myers' bit-vector recurrence (J. ACM 46(3) 1999, block form of Hyyro 2003) with the bit
// rows along the TEMPLATE and one column per query symbol.  Round 2 wrote the two delta vectors of every
// column to HBM (16 B per word and column: ~8 KB per 171 x 171 pair written and read back, 13 KB measured with
// the sector granularity) and was HBM bound.  This kernel keeps the walk's history in REGISTERS:
//   pass 1  forward over all columns; every S columns the column state {Pv, Mv} (K x 16 B) goes to a
//           checkpoint area -- ~1 KB per 171 x 171 pair;
//   pass 2  blocks of S columns from the last to the first: reload the state entering the block, recompute its
//           S columns keeping {Ph before the shift, Pv after the column} of each in registers (the block loop is
//           fully unrolled, so "column q of the block" is a fixed set of registers), then walk through the
//           block's columns in reverse, again statically unrolled -- the only dynamic choice left is which
//           32-bit half of a column's vectors holds the walk's row.
// Twice the arithmetic, 1/6 of the memory traffic, and the arithmetic is cheap: vectors as 32-bit halves,
// the three-input boolean steps as single v_bitop3_b32.
//
// Templates are TOP-ALIGNED in their K words (template symbol k at bit 64K - L + k; the bits below are padding
// rows whose initial vertical deltas are 0 and whose match masks are empty: such rows keep D[c][row] = c, i.e.
// they all behave like row 0 of the true problem), so the carry out of a word is always its bit 63 and the
// score delta of a column is bit 63 of the last word for every lane, whatever the template length.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "sd_nw.hpp"

namespace sd {


template <int K>
struct NwState {
    uint32_t PvL[K], PvH[K], MvL[K], MvH[K];
};

// One column: state -> state, returns the horizontal delta of the template's last row (+1 / 0 / -1).
// eq = the K match-mask words of the column's symbol; PhU* (optional) receive Ph before its shift.
template <int K, bool HIST>
__device__ __forceinline__ int nw_column(NwState<K>& s, const uint2* __restrict__ eq, uint32_t* PhUL, uint32_t* PhUH) {
    constexpr uint32_t TT_XH = (0xF0 ^ 0xCC) | 0xAA;               // (a ^ b) | c
    constexpr uint32_t TT_ORN = 0xF0 | (~(0xCC | 0xAA) & 0xFF);    // a | ~(b | c)
    uint32_t cP = 1, cM = 0;   // global alignment: D[c][0] - D[c-1][0] = +1 enters row 1
#pragma unroll
    for (int b = 0; b < K; ++b) {
        const uint2 e = eq[b];
        uint32_t EqL = e.x;
        const uint32_t EqH = e.y;
        const uint32_t pvL = s.PvL[b], pvH = s.PvH[b], mvL = s.MvL[b], mvH = s.MvH[b];
        const uint32_t XvL = EqL | mvL, XvH = EqH | mvH;
        EqL |= cM;
        const unsigned long long sum = (((unsigned long long)(EqH & pvH) << 32) | (EqL & pvL)) +
                                       (((unsigned long long)pvH << 32) | pvL);
        const uint32_t XhL = __builtin_amdgcn_bitop3_b32((uint32_t)sum, pvL, EqL, TT_XH);
        const uint32_t XhH = __builtin_amdgcn_bitop3_b32((uint32_t)(sum >> 32), pvH, EqH, TT_XH);
        uint32_t PhL = __builtin_amdgcn_bitop3_b32(mvL, XhL, pvL, TT_ORN);
        uint32_t PhH = __builtin_amdgcn_bitop3_b32(mvH, XhH, pvH, TT_ORN);
        uint32_t MhL = pvL & XhL, MhH = pvH & XhH;
        if (HIST) { PhUL[b] = PhL; PhUH[b] = PhH; }
        const uint32_t oP = PhH >> 31, oM = MhH >> 31;
        PhH = __builtin_amdgcn_alignbit(PhH, PhL, 31);
        MhH = __builtin_amdgcn_alignbit(MhH, MhL, 31);
        PhL = (PhL << 1) | cP;
        MhL = (MhL << 1) | cM;
        s.PvL[b] = __builtin_amdgcn_bitop3_b32(MhL, XvL, PhL, TT_ORN);
        s.PvH[b] = __builtin_amdgcn_bitop3_b32(MhH, XvH, PhH, TT_ORN);
        s.MvL[b] = PhL & XvL;
        s.MvH[b] = PhH & XvH;
        cP = oP;
        cM = oM;
    }
    return (int)cP - (int)cM;
}

// half `idx` of a vector of 2K halves {L[0], H[0], L[1], H[1], ...} (idx < 2K)
template <int K>
__device__ __forceinline__ uint32_t nw_pick(const uint32_t* L, const uint32_t* H, int idx) {
    const bool odd = idx & 1;
    const int w = idx >> 1;
    uint32_t v = odd ? H[0] : L[0];
#pragma unroll
    for (int b = 1; b < K; ++b) {
        const uint32_t x = odd ? H[b] : L[b];
        v = w == b ? x : v;
    }
    return v;
}

// Query symbols of a pair, read a dword at a time (4 ASCII symbols / 16 packed bases) and kept until the walk
// leaves the dword.  ASCII: bytes of a text (A, C, G, T, N).  Packed: 2 bit per base + an optional 1-bit N mask, as
// the DP kernels read them (sd_device.hpp: ChunkDesc::woff / noff).
struct NwQueryAscii {
    const uint8_t* text;    // whole text (readable up to the next multiple of 4 past its end)
    int64_t q0;             // first symbol of the segment
    int64_t wat = -1;
    uint32_t word = 0;
    __device__ __forceinline__ int code(int i) {
        const int64_t pos = q0 + i, a = pos >> 2;
        if (a != wat) { word = *reinterpret_cast<const uint32_t*>(text + (a << 2)); wat = a; }
        const uint32_t ch = (word >> (8 * (int)(pos & 3))) & 0xffu;
        return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
    }
};
struct NwQueryPacked {
    const uint32_t* w;      // bases of the chunk, 16 per dword
    const uint32_t* nm;     // N mask of the chunk (1 bit per base) or nullptr
    int off;                // first base of the segment inside the chunk
    int wat = -1, nat = -1;
    uint32_t word = 0, nword = 0;
    __device__ __forceinline__ int code(int i) {
        const int x = off + i, a = x >> 4;
        if (a != wat) { word = w[a]; wat = a; }
        int r = (int)((word >> (2 * (x & 15))) & 3u);
        if (nm) {
            const int na = x >> 5;
            if (na != nat) { nword = nm[na]; nat = na; }
            if ((nword >> (x & 31)) & 1u) r = 4;
        }
        return r;
    }
};

// The pair: ql query symbols (before compression), template masks eqt ([5][K] words, LDS or global), template
// length tl (1 <= tl <= 64 K), homo = homopolymer-compress the query on the fly (main.py:87-92; the templates are
// compressed by the host).  ck / ckpos: this lane's checkpoint slots, element stride `ckstride` in dwords: component
// x (PvL, PvH, MvL, MvH) of word b of slot s at ck[((s * K + b) * 4 + x) * ckstride] -- planar dwords, so that a
// wave's store is 256 contiguous bytes AND the four components need not sit in consecutive registers (as one 16-byte
// store per word they cost a dozen register moves per column to keep them there); ckpos[s * ckstride]; `cap` slots.
// Returns false when the pair needs more than `cap` checkpoints (nothing computed).
template <int K, class Query>
__device__ __forceinline__ bool nw_pair(Query& q, int ql, const uint2* __restrict__ eqt, int tl, bool homo,
                                        uint32_t* __restrict__ ck, int* __restrict__ ckpos, size_t ckstride, int cap,
                                        int& dist_out, int& matches_out) {
    constexpr int S = nw_block_cols(K);
    const int pad = 64 * K - tl;           // padding rows below the template
    auto init = [&](NwState<K>& s) {
#pragma unroll
        for (int b = 0; b < K; ++b) {
            // Pv = ~0 << pad over the K words
            const int lo = pad - 64 * b;   // bits below `lo` of this word are padding
            const unsigned long long m = lo <= 0 ? ~0ull : lo >= 64 ? 0ull : (~0ull << lo);
            s.PvL[b] = (uint32_t)m; s.PvH[b] = (uint32_t)(m >> 32);
            s.MvL[b] = 0u; s.MvH[b] = 0u;
        }
    };
    // ---- pass 1: forward, checkpoints of the state ENTERING columns S, 2S, ...
    NwState<K> st;
    init(st);
    // One trip per block of S columns, the columns statically unrolled: the state is renamed from column to column
    // instead of being copied back into loop-carried registers (a dozen v_mov per column in a column-per-trip loop).
    // A block starts at its first kept symbol; its later columns are executed unconditionally -- past the end of the
    // query with a dummy symbol and without counting -- which can only happen in the last block, whose exit state
    // nobody reads.
    int score = tl, c = 0, prev = -1, slot = -1, i = 0;
    auto next_kept = [&](int& r) -> bool {     // consumes symbols up to and including the next kept one
        while (i < ql) {
            r = q.code(i++);
            const bool skip = homo && r == prev;
            prev = r;
            if (!skip) return true;
        }
        return false;
    };
    while (true) {
        int r0 = 0;
        if (!next_kept(r0)) break;
        if (c > 0) {   // c is a multiple of S: checkpoint of the state entering column c
            ++slot;
            if (slot >= cap) return false;
#pragma unroll
            for (int b = 0; b < K; ++b) {
                uint32_t* w = ck + ((size_t)slot * K + b) * 4 * ckstride;
                w[0] = st.PvL[b]; w[ckstride] = st.PvH[b]; w[2 * ckstride] = st.MvL[b]; w[3 * ckstride] = st.MvH[b];
            }
            if (homo) ckpos[(size_t)slot * ckstride] = i - 1;   // where the block's first symbol is (plain: column c is symbol c)
        }
        score += nw_column<K, false>(st, eqt + r0 * K, nullptr, nullptr);
        ++c;
#pragma unroll
        for (int x = 1; x < S; ++x) {
            int r = 0;
            const bool got = next_kept(r);
            const int h = nw_column<K, false>(st, eqt + (got ? r : 0) * K, nullptr, nullptr);
            score += got ? h : 0;
            c += got ? 1 : 0;
        }
    }
    // ---- pass 2: blocks from the last to the first; walk with edlib's priority up > left > diagonal
    int row = tl, nL = 0;
    for (int blk = (c - 1) / S; blk >= 0 && row > 0; --blk) {
        int i;
        if (blk == 0) {
            init(st);
            i = 0;
        } else {
#pragma unroll
            for (int b = 0; b < K; ++b) {
                const uint32_t* w = ck + ((size_t)(blk - 1) * K + b) * 4 * ckstride;
                st.PvL[b] = w[0]; st.PvH[b] = w[ckstride]; st.MvL[b] = w[2 * ckstride]; st.MvH[b] = w[3 * ckstride];
            }
            i = homo ? ckpos[(size_t)(blk - 1) * ckstride] : blk * S;
        }
        const int ncol = min(S, c - blk * S);
        // All S columns of the block are computed, also those behind the last column of the pair (the final block is
        // usually partial): the walk below skips them, and nothing after the block reads the state -- an unconditional
        // column keeps "Pv after column x" in ONE set of registers that is both the history entry and the input of
        // column x + 1 (a predicated one costs a register copy per vector and column).
        uint32_t hPhL[S][K], hPhH[S][K], hPvL[S][K], hPvH[S][K];
        int pv = i > 0 ? q.code(i - 1) : -1;
#pragma unroll
        for (int x = 0; x < S; ++x) {
            int r = 0;
            if (x < ncol) {
                do {   // next kept symbol (the first symbol of a block is kept by construction of pass 1)
                    r = q.code(i++);
                    const bool skip = homo && r == pv;
                    pv = r;
                    if (!skip) break;
                } while (true);
            }
            NwState<K> in;
#pragma unroll
            for (int b2 = 0; b2 < K; ++b2) {
                in.PvL[b2] = x == 0 ? st.PvL[b2] : hPvL[x > 0 ? x - 1 : 0][b2];
                in.PvH[b2] = x == 0 ? st.PvH[b2] : hPvH[x > 0 ? x - 1 : 0][b2];
                in.MvL[b2] = st.MvL[b2];
                in.MvH[b2] = st.MvH[b2];
            }
            (void)nw_column<K, true>(in, eqt + r * K, hPhL[x], hPhH[x]);
#pragma unroll
            for (int b2 = 0; b2 < K; ++b2) {
                hPvL[x][b2] = in.PvL[b2]; hPvH[x][b2] = in.PvH[b2];
                st.MvL[b2] = in.MvL[b2]; st.MvH[b2] = in.MvH[b2];
            }
        }
#pragma unroll
        for (int x = S - 1; x >= 0; --x) {
            bool live = x < ncol && row > 0;
            while (live) {
                const int bit = pad + row - 1;
                const uint32_t ph = nw_pick<K>(hPhL[x], hPhH[x], bit >> 5);
                const uint32_t pw = nw_pick<K>(hPvL[x], hPvH[x], bit >> 5);
                const uint32_t up = (ph >> (bit & 31)) & 1u;
                const uint32_t lf = (pw >> (bit & 31)) & 1u & ~up;
                row -= (int)(1u - up);   // left or diagonal: one template symbol consumed
                nL += (int)lf;
                live = lf && row > 0;    // a left move stays in this column
            }
        }
    }
    nL += row;   // column 0 reached with template symbols left: all "left" moves
    dist_out = score;
    matches_out = c - score + nL;
    return true;
}

// Distance only: pass 1 of nw_pair without its checkpoints -- the edit distance and the number of columns (kept query
// symbols) of the pair.  Half the work of the full pair; what the bounds of the pruned homopolymer pass are made of
// (sd_ident.hip: sd_ident_dist).
template <int K, class Query>
__device__ __forceinline__ void nw_dist(Query& q, int ql, const uint2* __restrict__ eqt, int tl, bool homo,
                                        int& dist_out, int& cols_out) {
    constexpr int S = nw_block_cols(K);
    const int pad = 64 * K - tl;
    NwState<K> st;
#pragma unroll
    for (int b = 0; b < K; ++b) {
        const int lo = pad - 64 * b;
        const unsigned long long m = lo <= 0 ? ~0ull : lo >= 64 ? 0ull : (~0ull << lo);
        st.PvL[b] = (uint32_t)m; st.PvH[b] = (uint32_t)(m >> 32);
        st.MvL[b] = 0u; st.MvH[b] = 0u;
    }
    int score = tl, c = 0, prev = -1, i = 0;
    auto next_kept = [&](int& r) -> bool {
        while (i < ql) {
            r = q.code(i++);
            const bool skip = homo && r == prev;
            prev = r;
            if (!skip) return true;
        }
        return false;
    };
    while (true) {   // S columns per trip, statically unrolled, as in pass 1 of nw_pair
        int r0 = 0;
        if (!next_kept(r0)) break;
        score += nw_column<K, false>(st, eqt + r0 * K, nullptr, nullptr);
        ++c;
#pragma unroll
        for (int x = 1; x < S; ++x) {
            int r = 0;
            const bool got = next_kept(r);
            const int h = nw_column<K, false>(st, eqt + (got ? r : 0) * K, nullptr, nullptr);
            score += got ? h : 0;
            c += got ? 1 : 0;
        }
    }
    dist_out = score;
    cols_out = c;
}

}  // namespace sd
