// sd_nw_long.hip -- the identity kernel for templates of 513 .. 2048 bp (monomers of a kilobase and more): what
// stringdecomposer/main.py:29-60 (`edist` + `aai`) obtains from python-edlib for a (block, monomer) pair -- the unit-cost
// global alignment edlib's traceback picks (priority up > left > diagonal from the bottom-right corner,
// edlib.cpp:945-1150), of which the final TSV needs the distance and the number of '=' columns.
//
// sd_nw_kernel.hpp keeps a pair on ONE lane: K <= 8 words of 64 template rows per column, the walk's history in S*K*4
// registers.  At K = 16 .. 32 that leaves room for one column per block, i.e. a checkpoint of K x 16 bytes per column
// (1.2 MB per 2-kb pair: no faster than the host).  Here a pair lies ACROSS lanes instead: lane b owns
// block b (64 template rows) and the column loop is a systolic pipeline -- at step s lane b computes column s - b,
// taking the horizontal carry of block b - 1 from the lane below (computed one step earlier, one DPP move).  A pair
// of n columns takes n + K - 1 steps whatever K is.  Per step and lane: Myers' block update (J. ACM 46(3) 1999; block
// form of Hyyro 2003; the same 64-bit step as nw_column) on one word.
//
// Traceback without a full history: pass 1 runs the steps forward and leaves {Pv, Mv, carries} of every lane every
// S = 16 steps in HBM (20 bytes per lane and slot: ~20 KB per 1 kb x 1 kb pair); pass 2 takes the step blocks from the
// last to the first, recomputes a block's 16 steps with {Ph before the shift, Pv after the column} of every
// (lane, step) kept in LDS (8 KB per pair), and walks through them: the cell (row, column) belongs to lane
// owner = block of the row at step column + owner, and a walk only ever moves to smaller steps.
//
// LPP = lanes per pair (16 or 32): 4 or 2 pairs per wave, every pair's lanes doing the walk redundantly (its state is
// uniform within the pair); loops run to the wave's longest pair.  The query symbols (homopolymer-compressed on the
// fly where asked, main.py:87-92) are staged in LDS, one byte each.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "sd_nw.hpp"

namespace sd {

namespace {
constexpr int NWL_S = 16;   // steps per block of pass 2

__device__ __forceinline__ int nwl_code(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }

struct NwlState {
    unsigned long long Pv, Mv;
    uint32_t oP, oM;   // horizontal carries this lane produced in its last step
};

// one step of lane b: column update of its block with the carries (iP, iM) coming in from the block above it in the
// matrix (the lane below); PhU = Ph before its shift (history of the walk)
__device__ __forceinline__ void nwl_step(NwlState& s, unsigned long long Eq, uint32_t iP, uint32_t iM, unsigned long long& PhU) {
    const unsigned long long Xv = Eq | s.Mv;
    Eq |= (unsigned long long)iM;
    const unsigned long long sum = (Eq & s.Pv) + s.Pv;
    const unsigned long long Xh = (sum ^ s.Pv) | Eq;
    unsigned long long Ph = s.Mv | ~(Xh | s.Pv);
    unsigned long long Mh = s.Pv & Xh;
    PhU = Ph;
    s.oP = (uint32_t)(Ph >> 63);
    s.oM = (uint32_t)(Mh >> 63);
    Ph = (Ph << 1) | (unsigned long long)iP;
    Mh = (Mh << 1) | (unsigned long long)iM;
    s.Pv = Mh | ~(Xv | Ph);
    s.Mv = Ph & Xv;
}
}  // namespace

// plist: the pairs of this launch (all-vs-all: segment * T + template; pair_tmpl: the segment), results at dist /
// matches [pair id].  peq: [T][5][K] top-aligned masks (nw_build_masks).  ck: [gridDim.x * slots][cap][5][LPP] dwords.
// qcap: room for the longest segment of the launch (bytes of LDS per pair).
template <int LPP>
__global__ __launch_bounds__(128) void sd_nw_long(const uint8_t* __restrict__ seq, const int64_t* __restrict__ seg_start,
                                                  const int32_t* __restrict__ seg_len, const int64_t* __restrict__ plist,
                                                  int64_t n_pairs, int T, const int32_t* __restrict__ pair_tmpl,
                                                  const unsigned long long* __restrict__ peq,
                                                  const int32_t* __restrict__ tlen, int K, int homo, int qcap, int cap,
                                                  uint32_t* __restrict__ ck, int32_t* __restrict__ dist,
                                                  int32_t* __restrict__ matches) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPP, b = lane % LPP;
    const int lbase = lane - b;                         // first lane of the pair in its wave
    const int slots = (int)(blockDim.x >> 6) * PPW;     // pairs per workgroup
    const int slot = wave * PPW + sub;
    const size_t qroom = ((size_t)qcap + 15) & ~(size_t)15;
    const size_t per = qroom + (size_t)NWL_S * LPP * 16 + (size_t)5 * LPP * 8;
    unsigned char* mine = smem + (size_t)slot * per;
    uint8_t* qbuf = mine;
    uint4* hist = reinterpret_cast<uint4*>(mine + qroom);
    unsigned long long* eqs = reinterpret_cast<unsigned long long*>(mine + qroom + (size_t)NWL_S * LPP * 16);
    const unsigned long long submask = (LPP == 64 ? ~0ull : ((1ull << LPP) - 1ull));
    uint32_t* ckl = ck + ((size_t)blockIdx.x * (size_t)slots + (size_t)slot) * (size_t)cap * 5 * LPP + b;
    auto wave_max = [](int v) {
        for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off));
        return v;
    };
    auto lds_sync = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    for (int64_t p0 = (int64_t)blockIdx.x * slots; p0 < n_pairs; p0 += (int64_t)gridDim.x * slots) {
        const int64_t pi = p0 + slot;
        const bool have = pi < n_pairs;
        int64_t id = 0, sg = 0;
        int t = 0, ql = 0, tl = 1;
        if (have) {
            id = plist[pi];
            sg = pair_tmpl ? id : id / T;
            t = pair_tmpl ? pair_tmpl[sg] : (int)(id - sg * T);
            ql = seg_len[sg];
            tl = tlen[t];
        }
        const int64_t q0 = have ? seg_start[sg] : 0;
        // ---- the query into LDS, homopolymer runs collapsed where asked (main.py:87-92)
        int n = 0;
        {
            const int qmax_w = wave_max(ql);
            for (int i0 = 0; i0 < qmax_w; i0 += LPP) {
                const int i = i0 + b;
                int sym = 0;
                bool keep = false;
                if (i < ql) {
                    sym = nwl_code(seq[q0 + i]);
                    keep = !homo || i == 0 || sym != nwl_code(seq[q0 + i - 1]);
                }
                const unsigned long long m = (__ballot(keep) >> lbase) & submask;
                if (keep) qbuf[n + __popcll(m & ((1ull << b) - 1ull))] = (uint8_t)sym;
                n += __popcll(m);
            }
        }
        for (int sym = 0; sym < 5; ++sym) eqs[sym * LPP + b] = (have && b < K) ? peq[((size_t)t * 5 + (size_t)sym) * (size_t)K + (size_t)b] : 0ull;
        lds_sync();
        const int pad = 64 * K - tl;           // padding rows below the template (they behave like row 0)
        auto init = [&](NwlState& s) {
            const int lo = pad - 64 * b;       // bits below `lo` of this lane's word are padding
            s.Pv = lo <= 0 ? ~0ull : lo >= 64 ? 0ull : (~0ull << lo);
            s.Mv = 0ull;
            s.oP = 0u;
            s.oM = 0u;
        };
        const int ns = have ? n + K - 1 : 0;   // steps of this pair
        const int ns_w = wave_max(ns);
        // one step s of this lane; hist_x >= 0: leave the walk's history of the step in LDS
        auto step = [&](NwlState& s, int sidx, int hist_x, int& score) {
            // carries of the lane below, from its previous step: one DPP move (a global alignment enters row 1 with +1)
            const uint32_t cc = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(s.oP | (s.oM << 1)), 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
            uint32_t iP = cc & 1u, iM = cc >> 1;
            if (b == 0) { iP = 1u; iM = 0u; }
            const int c = sidx - b;
            if (have && b < K && c >= 0 && c < n) {
                const unsigned long long Eq = eqs[(int)qbuf[c] * LPP + b];
                unsigned long long PhU;
                nwl_step(s, Eq, iP, iM, PhU);
                if (b == K - 1) score += (int)s.oP - (int)s.oM;
                if (hist_x >= 0) hist[hist_x * LPP + b] = make_uint4((uint32_t)PhU, (uint32_t)(PhU >> 32), (uint32_t)s.Pv, (uint32_t)(s.Pv >> 32));
            }
        };
        // ---- pass 1: forward; the state entering step S, 2S, ... goes to the checkpoint area
        NwlState st;
        init(st);
        int score = 0;
        bool fits = true;
        for (int s = 0; s < ns_w; ++s) {
            if (s > 0 && (s & (NWL_S - 1)) == 0 && s < ns) {
                const int k = s / NWL_S - 1;
                if (k < cap) {
                    uint32_t* w = ckl + (size_t)k * 5 * LPP;
                    w[0] = (uint32_t)st.Pv; w[LPP] = (uint32_t)(st.Pv >> 32);
                    w[2 * LPP] = (uint32_t)st.Mv; w[3 * LPP] = (uint32_t)(st.Mv >> 32);
                    w[4 * LPP] = st.oP | (st.oM << 1);
                } else {
                    fits = false;
                }
            }
            step(st, s, -1, score);
        }
        const int dtot = tl + __shfl(score, lbase + K - 1);
        // ---- pass 2: step blocks from the last to the first; walk with edlib's priority up > left > diagonal
        int row = tl, col = n - 1, nL = 0;
        const int nb = (ns + NWL_S - 1) / NWL_S;
        const int nb_w = wave_max(nb);
        for (int blk = nb_w - 1; blk >= 0; --blk) {
            const bool inb = have && blk < nb;
            const int s0 = blk * NWL_S;
            if (blk == 0) {
                init(st);
            } else if (inb) {
                const uint32_t* w = ckl + (size_t)(blk - 1) * 5 * LPP;
                st.Pv = (unsigned long long)w[0] | ((unsigned long long)w[LPP] << 32);
                st.Mv = (unsigned long long)w[2 * LPP] | ((unsigned long long)w[3 * LPP] << 32);
                const uint32_t cc = w[4 * LPP];
                st.oP = cc & 1u;
                st.oM = (cc >> 1) & 1u;
            }
            int dummy = 0;
            for (int x = 0; x < NWL_S; ++x) step(st, s0 + x, inb ? x : -1, dummy);
            lds_sync();
            // the walk inside this block (every lane of the pair carries the same row / col / nL)
            for (;;) {
                const int bitpos = pad + row - 1;
                const int owner = bitpos >> 6;
                const int s = col + owner;
                const bool go = inb && row > 0 && col >= 0 && s >= s0;
                if (__ballot(go) == 0ull) break;
                if (go) {
                    const uint4 h = hist[(s - s0) * LPP + owner];
                    const int bit = bitpos & 63;
                    const uint32_t ph = bit < 32 ? h.x : h.y, pw = bit < 32 ? h.z : h.w;
                    const uint32_t up = (ph >> (bit & 31)) & 1u;
                    const uint32_t lf = (pw >> (bit & 31)) & 1u & ~up;
                    if (up) {
                        --col;                 // 'I': a query symbol consumed
                    } else {
                        --row;                 // left ('D') or diagonal: a template symbol consumed
                        nL += (int)lf;
                        if (!lf) --col;        // a left move stays in its column
                    }
                }
            }
            lds_sync();
        }
        nL += row;   // column 0 left with template symbols to go: all "left" moves
        if (have && b == 0) {
            if (fits) { dist[id] = dtot; matches[id] = n - dtot + nL; }
            else { dist[id] = -2; matches[id] = 0; }   // more steps than checkpoint slots (the host sizes them: cannot happen)
        }
    }
}

size_t nw_long_lds_bytes(int lpp, int qcap, int block_threads) {
    const size_t qroom = ((size_t)qcap + 15) & ~(size_t)15;
    return (size_t)(block_threads / 64) * (size_t)(64 / lpp) * (qroom + (size_t)NWL_S * lpp * 16 + (size_t)5 * lpp * 8);
}
int nw_long_slots(int qmax, int K) { return (qmax + K - 1 + NWL_S - 1) / NWL_S + 1; }

void launch_nw_long(int K, hipStream_t st, int grid, int block_threads, const uint8_t* seq, const int64_t* seg_start,
                    const int32_t* seg_len, const int64_t* plist, int64_t n_pairs, int T, const int32_t* pair_tmpl,
                    const unsigned long long* peq, const int32_t* tlen, int homo, int qcap, int cap, void* ck,
                    int32_t* dist, int32_t* matches) {
    const int lpp = K <= 16 ? 16 : 32;
    const size_t lds = nw_long_lds_bytes(lpp, qcap, block_threads);
    if (lpp == 16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_nw_long<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(sd_nw_long<16>, dim3(grid), dim3(block_threads), lds, st, seq, seg_start, seg_len, plist, n_pairs, T,
                           pair_tmpl, peq, tlen, K, homo, qcap, cap, static_cast<uint32_t*>(ck), dist, matches);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sd_nw_long<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(sd_nw_long<32>, dim3(grid), dim3(block_threads), lds, st, seq, seg_start, seg_len, plist, n_pairs, T,
                           pair_tmpl, peq, tlen, K, homo, qcap, cap, static_cast<uint32_t*>(ck), dist, matches);
    }
}

}  // namespace sd
