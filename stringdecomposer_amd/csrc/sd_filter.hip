// sd_filter.hip -- the optional --ed_thr prefilter on the device.
//
// Reference: MonomersAligner::FilterMonomersForRead / MonomerEditDistance
// (stringdecomposer/src/main.cpp:128-149): per chunk, the infix ("HW") edit distance of every
// template against the chunk (there via vendored edlib), templates ordered by (distance, index),
// the first one and every other one with distance <= ed_thr kept IN THAT ORDER -- the order
// matters because AlignPartClassicDP breaks score ties towards the first template of its list.
//
//  sd_hw_dist<W> one lane per (chunk, template): Myers' bit-vector algorithm (J. ACM 46(3), 1999)
//                in its block form (Hyyro 2003), W 64-bit words per template (patterns of up to 512
//                symbols), match masks in LDS, search variant (free leading / trailing text), minimum
//                of the bottom-row score over all columns.
//  sd_rank_keep  one thread per (chunk, template): kept? + rank inside the filtered order; writes the
//                per-chunk lane constants the ranked fast fills use (end offsets of dropped templates
//                become -inf, ties between template ends go to the smallest rank) or the rank table of
//                the generic family.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "sd_device.hpp"
#include "sd_fast.hpp"

namespace sd {

// One lane per (chunk, template) pair, consecutive lanes = consecutive templates of a chunk: the lanes of a
// wave then read the same few chunks' bases (one broadcast load per 16 columns) and every lane does useful
// work for any template count (a wave per chunk with the templates across the lanes would idle 40 of 64
// lanes at 24 templates).  W = 64-bit words per template, a template parameter so that the delta vectors
// live in exactly 4*W registers; the match masks of all templates sit in LDS ([template][symbol][word]).
template <int W>
__global__ __launch_bounds__(256) void sd_hw_dist(const ChunkDesc* __restrict__ chunks, int n_chunks,
                                                  int T, const uint32_t* __restrict__ bases2,
                                                  const uint32_t* __restrict__ nmask,
                                                  const unsigned long long* __restrict__ peq,
                                                  const int32_t* __restrict__ tlen,
                                                  int32_t* __restrict__ dist, int lds_templates) {
    constexpr int PS = W <= 8 ? 8 : W;             // words per (template, symbol) row of peq (build_peq)
    extern __shared__ unsigned long long speq[];   // [min(T, lds_templates)][5][W]
    for (int idx = threadIdx.x; idx < lds_templates * 5 * W; idx += blockDim.x) {
        const int j = idx / (5 * W), rem = idx % (5 * W);
        speq[idx] = peq[(size_t)j * 5 * PS + (size_t)(rem / W) * PS + (rem % W)];
    }
    __syncthreads();
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)n_chunks * T) return;
    const int c = (int)(g / T), j = (int)(g % T);
    const ChunkDesc cd = chunks[c];
    const int m = tlen[j];
    const int lastW = (m - 1) >> 6;
    const unsigned long long lastBit = 1ull << ((m - 1) & 63);
    const bool in_lds = j < lds_templates;
    const unsigned long long* pq_l = speq + (size_t)j * 5 * W;
    const unsigned long long* pq_g = peq + (size_t)j * 5 * PS;
    unsigned long long Pv[W], Mv[W];
#pragma unroll
    for (int b = 0; b < W; ++b) { Pv[b] = ~0ull; Mv[b] = 0ull; }
    int score = m, best = m;
    const uint32_t* w = bases2 + cd.woff;
    const uint32_t* nm = cd.noff >= 0 ? nmask + cd.noff : nullptr;
    uint32_t wb = 0, nb = 0;
    for (int i = 0; i < cd.n; ++i) {
        if ((i & 15) == 0) wb = w[i >> 4];
        int r = (wb >> (2 * (i & 15))) & 3;
        if (nm) {
            if ((i & 31) == 0) nb = nm[i >> 5];
            if ((nb >> (i & 31)) & 1) r = 4;
        }
        int hin = 0;  // search variant: the row above the pattern costs nothing
#pragma unroll
        for (int b = 0; b < W; ++b) {
            if (b <= lastW) {
                unsigned long long Eq = in_lds ? pq_l[r * W + b] : pq_g[r * PS + b];
                const unsigned long long pv = Pv[b], mv = Mv[b];
                const unsigned long long Xv = Eq | mv;
                if (hin < 0) Eq |= 1ull;
                const unsigned long long Xh = (((Eq & pv) + pv) ^ pv) | Eq;
                unsigned long long Ph = mv | ~(Xh | pv);
                unsigned long long Mh = pv & Xh;
                const unsigned long long top = (b == lastW) ? lastBit : (1ull << 63);
                int hout = 0;
                if (Ph & top) hout = 1;
                if (Mh & top) hout = -1;
                Ph <<= 1;
                Mh <<= 1;
                if (hin < 0) Mh |= 1ull;
                if (hin > 0) Ph |= 1ull;
                Pv[b] = Mh | ~(Xv | Ph);
                Mv[b] = Ph & Xv;
                hin = hout;
            }
        }
        score += hin;  // vertical delta of the pattern's last row
        best = min(best, score);
    }
    dist[(size_t)c * T + j] = best;
}

// The common case made cheap: every template occupies the same number of 64-bit words, the end of every template
// falls into the same 32-bit half of its last word (true for any set of monomers of similar length, e.g. 161..192
// bp), and all match masks fit LDS.  Then nothing in the word loop depends on the lane but the data: the bit
// vectors live as 32-bit halves, the three-input boolean steps are single v_bitop3_b32 instructions (truth table
// over a = 0xF0, b = 0xCC, c = 0xAA), the carries between words are plain 0/1 values, and the masks come straight
// from LDS -- about 80 VALU instructions per column at three words instead of 136.  Same result as sd_hw_dist.
template <int W, bool HI>
__global__ __launch_bounds__(256) void sd_hw_dist_u(const ChunkDesc* __restrict__ chunks, int n_chunks, int T,
                                                    const uint32_t* __restrict__ bases2,
                                                    const uint32_t* __restrict__ nmask,
                                                    const unsigned long long* __restrict__ peq,
                                                    const int32_t* __restrict__ tlen, int32_t* __restrict__ dist) {
    // the 256 pairs of a workgroup are consecutive templates (of one or two chunks): only their masks are staged,
    // min(T, 256) x 5 x W words, whatever the size of the set
    extern __shared__ unsigned long long speq[];
    const long long g0 = (long long)blockIdx.x * blockDim.x;
    const int nl = T < 256 ? T : 256;
    const int j0 = T < 256 ? 0 : (int)(g0 % T);
    for (int idx = threadIdx.x; idx < nl * 5 * W; idx += blockDim.x) {
        const int jl = idx / (5 * W), rem = idx % (5 * W);
        int j = j0 + jl;
        if (j >= T) j -= T;
        speq[idx] = peq[(size_t)j * 40 + (size_t)(rem / W) * 8 + (rem % W)];
    }
    __syncthreads();
    const long long g = g0 + threadIdx.x;
    if (g >= (long long)n_chunks * T) return;
    const int c = (int)(g / T), j = (int)(g % T);
    const ChunkDesc cd = chunks[c];
    const int m = tlen[j];
    const uint32_t sh = (uint32_t)((m - 1) & 31);           // end of the template inside its half
    int jl = j - j0;
    if (jl < 0) jl += T;
    const unsigned long long* pq = speq + (size_t)jl * 5 * W;
    constexpr uint32_t TT_XH = (0xF0 ^ 0xCC) | 0xAA;        // (a ^ b) | c
    constexpr uint32_t TT_ORN = 0xF0 | (~(0xCC | 0xAA) & 0xFF);   // a | ~(b | c)
    uint32_t PvL[W], PvH[W], MvL[W], MvH[W];
#pragma unroll
    for (int b = 0; b < W; ++b) { PvL[b] = PvH[b] = ~0u; MvL[b] = MvH[b] = 0u; }
    int score = m, best = m;
    const uint32_t* w = bases2 + cd.woff;
    const uint32_t* nm = cd.noff >= 0 ? nmask + cd.noff : nullptr;
    uint32_t wb = 0, nb = 0;
    for (int i = 0; i < cd.n; ++i) {
        if ((i & 15) == 0) wb = w[i >> 4];
        int r = (wb >> (2 * (i & 15))) & 3;
        if (nm) {
            if ((i & 31) == 0) nb = nm[i >> 5];
            if ((nb >> (i & 31)) & 1) r = 4;
        }
        const unsigned long long* eqp = pq + r * W;
        uint32_t cP = 0, cM = 0;   // horizontal delta entering the word: +1 / -1 (search variant: 0 above the pattern)
#pragma unroll
        for (int b = 0; b < W; ++b) {
            const unsigned long long Eq64 = eqp[b];
            uint32_t EqL = (uint32_t)Eq64;
            const uint32_t EqH = (uint32_t)(Eq64 >> 32);
            const uint32_t pvL = PvL[b], pvH = PvH[b], mvL = MvL[b], mvH = MvH[b];
            const uint32_t XvL = EqL | mvL, XvH = EqH | mvH;
            EqL |= cM;
            const unsigned long long sum = (((unsigned long long)(EqH & pvH) << 32) | (EqL & pvL)) +
                                           (((unsigned long long)pvH << 32) | pvL);
            const uint32_t XhL = __builtin_amdgcn_bitop3_b32((uint32_t)sum, pvL, EqL, TT_XH);
            const uint32_t XhH = __builtin_amdgcn_bitop3_b32((uint32_t)(sum >> 32), pvH, EqH, TT_XH);
            uint32_t PhL = __builtin_amdgcn_bitop3_b32(mvL, XhL, pvL, TT_ORN);
            uint32_t PhH = __builtin_amdgcn_bitop3_b32(mvH, XhH, pvH, TT_ORN);
            uint32_t MhL = pvL & XhL, MhH = pvH & XhH;
            uint32_t oP, oM;   // horizontal delta leaving the word (of the template's last row in the last word)
            if (b == W - 1) {
                oP = ((HI ? PhH : PhL) >> sh) & 1u;
                oM = ((HI ? MhH : MhL) >> sh) & 1u;
            } else {
                oP = PhH >> 31;
                oM = MhH >> 31;
            }
            PhH = __builtin_amdgcn_alignbit(PhH, PhL, 31);
            MhH = __builtin_amdgcn_alignbit(MhH, MhL, 31);
            PhL = (PhL << 1) | cP;
            MhL = (MhL << 1) | cM;
            PvL[b] = __builtin_amdgcn_bitop3_b32(MhL, XvL, PhL, TT_ORN);
            PvH[b] = __builtin_amdgcn_bitop3_b32(MhH, XvH, PhH, TT_ORN);
            MvL[b] = PhL & XvL;
            MvH[b] = PhH & XvH;
            cP = oP;
            cM = oM;
        }
        score += (int)cP - (int)cM;  // vertical delta of the pattern's last row
        best = min(best, score);
    }
    dist[(size_t)c * T + j] = best;
}

// Kept set and order of a chunk (main.cpp:141-147): first = smallest (distance, index); kept = first or
// distance <= ed_thr; rank = position in the (distance, index) order among the kept.  Written as the
// per-chunk lane constants of the ranked fast fills (end offsets / ranks, [chunk][wave][64 lanes] dwords, packed
// {lo plane, hi plane} int16) or, for the generic family, as a rank table [chunk][T] (0xffff = dropped).
__global__ void sd_rank_keep(int n_chunks, int T, int ed_thr, const int32_t* __restrict__ dist,
                             const int32_t* __restrict__ end_vlane, const int32_t* __restrict__ end_off,
                             uint16_t* __restrict__ cendoff, uint16_t* __restrict__ crank,
                             uint16_t* __restrict__ grank, int W, uint16_t* __restrict__ kpos,
                             uint16_t* __restrict__ klist, int32_t* __restrict__ nkept,
                             const int32_t* __restrict__ vlane0) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)n_chunks * T) return;
    const int c = (int)(g / T), j = (int)(g % T);
    const int32_t* d = dist + (size_t)c * T;
    const int dj = d[j];
    int first = 0;
    for (int i = 1; i < T; ++i)
        if (d[i] < d[first]) first = i;
    const bool kept = j == first || dj <= ed_thr;
    int rank = 0, total = 0;
    for (int i = 0; i < T; ++i) {
        const bool ki = i == first || d[i] <= ed_thr;
        total += ki ? 1 : 0;
        if (ki && (d[i] < dj || (d[i] == dj && i < j))) ++rank;
    }
    if (kpos) {
        // compacted form for large template sets (sd_fast_wn_ck.hip): the kept templates of the chunk in their
        // filtered order (klist: [chunk][T], filled with 0xffff before), every template's place in it, the count
        kpos[(size_t)c * T + j] = (uint16_t)(kept ? rank : 0xffff);
        if (kept) klist[(size_t)c * T + rank] = (uint16_t)j;
        if (j == 0) nkept[c] = total;
    }
    if (grank) {
        grank[(size_t)c * T + j] = (uint16_t)(kept ? rank : 0xffff);
        return;
    }
    // every virtual lane of the template ((wave << 7) | (plane << 6) | lane; the lanes of a template are consecutive
    // in one plane): the fp16 fills reduce the totals of all lanes (the maximum over a template's lanes is its end)
    const int v1 = end_vlane[j], v0 = vlane0 ? vlane0[j] : v1;
    for (int v = v0; v <= v1; ++v) {
        const size_t at = (((size_t)c * W + (size_t)(v >> 7)) * 64 + (v & 63)) * 2 + ((v >> 6) & 1);
        cendoff[at] = (uint16_t)(kept ? end_off[j] : -32768);
        crank[at] = (uint16_t)(kept ? rank : 0x7fff);
    }
}

__global__ void sd_fill_u32(uint32_t* p, size_t n, uint32_t v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// match masks of the prefilter: [template][5 symbols][PS words], PS = 8 for templates of up to 512 bp (the layout the
// uniform kernel and the fast family know), 16 / 32 for sets with a longer template (up to 2048 bp: generic family)
void build_peq(const std::vector<std::string>& tseq, std::vector<unsigned long long>& peq) {
    size_t lmax = 1;
    for (const std::string& t : tseq) lmax = std::max(lmax, t.size());
    const size_t PS = lmax <= 512 ? 8 : lmax <= 1024 ? 16 : 32;
    peq.assign(tseq.size() * 5 * PS, 0ull);
    for (size_t j = 0; j < tseq.size(); ++j)
        for (size_t k = 0; k < tseq[j].size() && k < 64 * PS; ++k) {
            int code;
            switch (tseq[j][k]) {
                case 'A': code = 0; break;
                case 'C': code = 1; break;
                case 'G': code = 2; break;
                case 'T': code = 3; break;
                default: code = 4; break;
            }
            peq[(j * 5 + (size_t)code) * PS + (k >> 6)] |= 1ull << (k & 63);
        }
}

// Splits the chunk order (longest first) of a batch into W classes by the number of waves a chunk's kept templates
// need, ceil(kept / 128), each in the original order; counts[w-1] = size of class w.  One wave.
__global__ __launch_bounds__(64) void sd_split_order(const int* __restrict__ order, int n, const int32_t* __restrict__ nkept,
                                                     int* __restrict__ orders, int* __restrict__ counts, int W) {
    const int lane = threadIdx.x;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int base = 0; base < n; base += 64) {
        const int x = base + lane;
        const int c = x < n ? order[x] : -1;
        const int cls = (c >= 0 && nkept[c] >= 0) ? min(W, max(1, (nkept[c] + 127) / 128)) - 1 : -1;   // (-1: a skipped chunk)
        for (int w = 0; w < W; ++w) {
            const unsigned long long m = __ballot(cls == w);
            if (cls == w) orders[(size_t)w * n + cnt[w] + __popcll(m & ((1ull << lane) - 1ull))] = c;
            cnt[w] += __popcll(m);
        }
    }
    for (int w = 0; w < W; ++w)
        if (lane == w) counts[w] = cnt[w];
}

// --ed_thr on the tiled multi-wave layout (sd_fast_wt.hip): the kept templates of a chunk, in their filtered order, over
// consecutive virtual lanes -- ceil(L / P) lanes each, never across a plane (64 lanes), as fast_plan_build places the
// whole set.  One thread per chunk (a serial walk: where a template starts depends on the padding before it).
// Out: first lane of every kept template (into the place table kpos, which the tiled traceback reads as such), the
// template | part << 16 of every lane (lane_t, 0xffffffff = idle; cleared by the launcher), and the lanes used (into
// nkept, which the class split and the traceback read as "what the chunk needs": more than 128 (W - 1) = all W waves,
// the ranked kernel on the plan's layout).
__global__ void sd_tiled_place(int n_chunks, int T, int P, int W, const uint16_t* __restrict__ klist,
                               int32_t* __restrict__ nkept, const int32_t* __restrict__ tlen,
                               uint16_t* __restrict__ kpos, uint32_t* __restrict__ lane_t, int* __restrict__ overflow_flag) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int nk = nkept[c];
    const uint16_t* kl = klist + (size_t)c * (size_t)T;
    uint32_t* lt = lane_t + (size_t)c * (size_t)(W * 128);
    int cur = 0;
    for (int r = 0; r < nk; ++r) {
        const int j = kl[r];
        const int V = (tlen[j] + P - 1) / P;
        if (V > 64 - (cur & 63)) cur = (cur + 63) & ~63;
        if (cur + V > W * 128) {   // (padding in another order than the plan's: all W waves, on the plan's layout)
            cur = W * 128 + 1;
            // FastPlan::filter_only -- there is no layout of the whole set: the chunk is skipped (count -1) and the flag
            // makes the host repeat the batch on the generic family
            if (overflow_flag) { cur = -1; atomicOr(overflow_flag, 4); }
            break;
        }
        kpos[(size_t)c * (size_t)T + (size_t)j] = (uint16_t)cur;
        for (int u = 0; u < V; ++u) lt[cur + u] = (uint32_t)j | ((uint32_t)u << 16);
        cur += V;
    }
    nkept[c] = cur;
}

void launch_tiled_place(hipStream_t st, int n_chunks, int T, int P, int W, const uint16_t* klist, int32_t* nkept,
                        const int32_t* tlen, uint16_t* kpos, uint32_t* lane_t, int* overflow_flag) {
    const size_t words = (size_t)n_chunks * (size_t)W * 128;
    hipLaunchKernelGGL(sd_fill_u32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, lane_t, words, 0xffffffffu);
    hipLaunchKernelGGL(sd_tiled_place, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 0, st, n_chunks, T, P, W, klist, nkept,
                       tlen, kpos, lane_t, overflow_flag);
}

void launch_split_order(hipStream_t st, const int* order, int n, const int32_t* nkept, int* orders, int* counts, int W) {
    hipLaunchKernelGGL(sd_split_order, dim3(1), dim3(64), 0, st, order, n, nkept, orders, counts, W);
}

void launch_edthr_filter(hipStream_t st, const ChunkDesc* chunks, int n_chunks, int T, int Lmax, int ed_thr,
                         const uint32_t* bases2, const uint32_t* nmask, const unsigned long long* peq,
                         const int32_t* tlen, const int32_t* end_vlane, const int32_t* end_off,
                         int32_t* dist, uint32_t* cendoff, uint32_t* crank, uint16_t* grank, int waves,
                         uint16_t* kpos, uint16_t* klist, int32_t* nkept, int uniform_half, const int32_t* vlane0) {
    const long long total = (long long)n_chunks * T;
    const int grid = (int)((total + 255) / 256);
    if (!grank) {
        const size_t words = (size_t)n_chunks * 64 * (size_t)waves;
        hipLaunchKernelGGL(sd_fill_u32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, cendoff, words, 0x80008000u);
        hipLaunchKernelGGL(sd_fill_u32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, crank, words, 0x7fff7fffu);
    }
    if (klist) {
        const size_t words = ((size_t)n_chunks * T + 1) / 2;   // [chunk][T] uint16
        hipLaunchKernelGGL(sd_fill_u32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<uint32_t*>(klist), words, 0xffffffffu);
    }
    const int W = std::max(1, (Lmax + 63) / 64);
    // match masks in LDS: as many templates as fit in 64 KB (all of them up to ~540 at 3 words)
#define SD_HW(WW)                                                                                              \
    {                                                                                                          \
        const int lt = std::min(T, (int)((64 * 1024) / (5 * WW * sizeof(unsigned long long))));                \
        hipLaunchKernelGGL(sd_hw_dist<WW>, dim3(grid), dim3(256), (size_t)lt * 5 * WW * sizeof(unsigned long long), st, \
                           chunks, n_chunks, T, bases2, nmask, peq, tlen, dist, lt);                           \
    }
    // uniform variant (sd_hw_dist_u): same word count and same half of the last word for every template, masks in LDS
    bool done = false;
    if (uniform_half >= 0 && W >= 1 && W <= 4) {   // (SD_FLAG_FILTER_GENERAL: the engine passes uniform_half = -1)
        const size_t lds = (size_t)std::min(T, 256) * 5 * W * sizeof(unsigned long long);   // <= 40 KB
#define SD_HWU(WW)                                                                                             \
        {                                                                                                      \
            if (uniform_half) hipLaunchKernelGGL((sd_hw_dist_u<WW, true>), dim3(grid), dim3(256), lds, st, chunks, n_chunks, T, bases2, nmask, peq, tlen, dist); \
            else hipLaunchKernelGGL((sd_hw_dist_u<WW, false>), dim3(grid), dim3(256), lds, st, chunks, n_chunks, T, bases2, nmask, peq, tlen, dist); \
            done = true;                                                                                       \
        }
        if (W == 1) SD_HWU(1) else if (W == 2) SD_HWU(2) else if (W == 3) SD_HWU(3) else SD_HWU(4)
#undef SD_HWU
    }
    if (!done) {
    if (W <= 1) SD_HW(1) else if (W == 2) SD_HW(2) else if (W == 3) SD_HW(3) else if (W == 4) SD_HW(4)
    else if (W <= 8) SD_HW(8) else if (W <= 16) SD_HW(16) else SD_HW(32)   // 16 / 32 words: templates of up to 1024 / 2048 bp
    }
#undef SD_HW
    hipLaunchKernelGGL(sd_rank_keep, dim3(grid), dim3(256), 0, st, n_chunks, T, ed_thr, dist, end_vlane, end_off,
                       reinterpret_cast<uint16_t*>(cendoff), reinterpret_cast<uint16_t*>(crank), grank, waves, kpos, klist,
                       nkept, vlane0);
}

}  // namespace sd
