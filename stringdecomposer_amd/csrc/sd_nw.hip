// sd_nw.hip -- batched unit-cost global alignment identity on the device (post-processing of the
// drop-in CLI: what stringdecomposer/main.py:29-60 `edist` + `aai` obtain from python-edlib for every
// (block, monomer) pair; with --second-best that is 2*T alignments per output row, main.py:118-146).
//
// One lane per (read segment, template) pair.  Myers' bit-vector algorithm (J. ACM 46(3), 1999; block
// form of Hyyro 2003) with the bit rows along the TEMPLATE (its match masks are shared by every pair of
// that template) and one column per segment symbol, K 64-bit words per column.  edlib's traceback
// (obtainAlignmentTraceback, edlib.cpp:945-1150) walks from the bottom-right corner with priority
//     up   (consume a query symbol,  'I'):  D[i-1][j] + 1 == D[i][j]
//     left (consume a target symbol, 'D'):  D[i][j-1] + 1 == D[i][j]
//     diagonal ('=' / 'X')
// Here the query (segment) runs along the columns, so "up" is the positive HORIZONTAL delta of the
// cell and "left" the positive VERTICAL delta: the forward pass stores, per column, the two delta
// vectors {Ph (before its shift), Pv (after the column)} -- 16 B per word -- in a per-lane history in
// HBM, and the walk reads one such pair per step.  The walk only counts: with nL = number of "left"
// moves, an optimal path has  matches = qlen - dist + nL  ('=' columns) and dist + matches columns.
//
// History: [workgroup][column][K][thread] x 16 B: the 64 lanes of a wave write one contiguous KB per (column,
// word), and lanes that walk near the same column read neighbouring pieces of the same sectors.  ~8 KB per
// 171 x 171 pair, written once and read about once: the kernel is HBM-bound, not ALU-bound.  (A lane-major
// history -- every lane its own 8-KB region -- wrote 16-B pieces of 64 different sectors per instruction:
// 2.4x the bytes at the memory, 73 M instead of 141 M pairs/s end to end.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sd_hip.h"
#include "sd_host.hpp"
#include "sd_nw.hpp"

namespace sd {

namespace {
__device__ __forceinline__ int base_code_dev(uint32_t ch) {
    // A,C,G,T,N -> 0..4 (the caller guarantees the alphabet)
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
}
}  // namespace

template <int K>
__global__ __launch_bounds__(256) void sd_nw_pairs(const uint8_t* __restrict__ seq,
                                                   const int64_t* __restrict__ seg_start,
                                                   const int32_t* __restrict__ seg_len, int64_t n_seg, int T,
                                                   const int32_t* __restrict__ pair_tmpl,
                                                   const unsigned long long* __restrict__ peq,
                                                   const int32_t* __restrict__ tlen, int homo, int qmax,
                                                   uint4* __restrict__ hist, int32_t* __restrict__ dist,
                                                   int32_t* __restrict__ matches) {
    typedef unsigned long long u64;
    const int64_t n_pairs = pair_tmpl ? n_seg : n_seg * (int64_t)T;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // history of this lane: [column][word][thread of the workgroup] -- the 64 lanes of a wave write one
    // contiguous KB per (column, word), whole sectors (a lane-major layout wrote 16-B pieces of 64 different
    // sectors per instruction: 2.4x the bytes at the memory)
    uint4* H = hist + (size_t)blockIdx.x * 256 * (size_t)qmax * K + threadIdx.x;
    for (int64_t p = gid; p < n_pairs; p += stride) {
        const int64_t s = pair_tmpl ? p : p / T;
        const int t = pair_tmpl ? pair_tmpl[p] : (int)(p - s * T);
        const int ql = seg_len[s];
        const int tl = tlen[t];
        if (ql <= 0 || tl <= 0) {  // main.py:30-33: an empty side has no alignment
            dist[p] = -1;
            matches[p] = 0;
            continue;
        }
        const int64_t q0 = seg_start[s];
        const u64* pq = peq + (size_t)t * 5 * K;
        const int lastW = (tl - 1) >> 6;
        const u64 lastBit = 1ull << ((tl - 1) & 63);
        u64 Pv[K], Mv[K];
#pragma unroll
        for (int w = 0; w < K; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
        int score = tl;  // D[0][tl]
        int c = 0;       // columns done (= query symbols kept)
        uint32_t prev = 0x100, word = 0;
        for (int i = 0; i < ql; ++i) {
            const int64_t pos = q0 + i;
            if (i == 0 || (pos & 3) == 0) word = *reinterpret_cast<const uint32_t*>(seq + (pos & ~(int64_t)3));
            const uint32_t ch = (word >> (8 * (int)(pos & 3))) & 0xffu;
            if (homo && ch == prev) continue;  // homopolymer compression of the query, main.py:87-92
            prev = ch;
            const u64* eqp = pq + base_code_dev(ch) * K;
            int hin = 1;  // global alignment: D[c][0] - D[c-1][0] = 1
            uint4* hc = H + (size_t)c * K * 256;
#pragma unroll
            for (int w = 0; w < K; ++w) {
                if (w <= lastW) {
                    u64 Eq = eqp[w];
                    const u64 pv = Pv[w], mv = Mv[w];
                    const u64 Xv = Eq | mv;
                    if (hin < 0) Eq |= 1ull;
                    const u64 Xh = (((Eq & pv) + pv) ^ pv) | Eq;
                    u64 Ph = mv | ~(Xh | pv);
                    u64 Mh = pv & Xh;
                    const u64 top = w == lastW ? lastBit : (1ull << 63);
                    const int hout = (Ph & top) ? 1 : ((Mh & top) ? -1 : 0);
                    const u64 PhU = Ph;  // delta of row r lives in bit r-1 before the shift
                    Ph <<= 1;
                    Mh <<= 1;
                    if (hin < 0) Mh |= 1ull;
                    if (hin > 0) Ph |= 1ull;
                    const u64 npv = Mh | ~(Xv | Ph);
                    Pv[w] = npv;
                    Mv[w] = Ph & Xv;
                    hin = hout;
                    hc[w * 256] = make_uint4((uint32_t)PhU, (uint32_t)(PhU >> 32), (uint32_t)npv, (uint32_t)(npv >> 32));
                }
            }
            score += hin;  // vertical... the last word's carry is the horizontal delta of row tl
            ++c;
        }
        // walk (edlib priority: up > left > diagonal), counting the "left" moves
        int ci = c, r = tl, nL = 0;
        while (ci > 0 && r > 0) {
            const uint4 h = H[((size_t)(ci - 1) * K + ((r - 1) >> 6)) * 256];
            const int b = (r - 1) & 63;
            const u64 ph = ((u64)h.y << 32) | h.x, pv = ((u64)h.w << 32) | h.z;
            if ((ph >> b) & 1ull) { --ci; }
            else if ((pv >> b) & 1ull) { --r; ++nL; }
            else { --ci; --r; }
        }
        nL += r;  // column 0 reached with target symbols left: they are all "left" moves
        dist[p] = score;
        matches[p] = c - score + nL;
    }
}

void launch_nw_pairs(int K, hipStream_t st, int grid, const uint8_t* seq, const int64_t* seg_start,
                     const int32_t* seg_len, int64_t n_seg, int T, const int32_t* pair_tmpl,
                     const unsigned long long* peq, const int32_t* tlen, int homo, int qmax, void* hist,
                     int32_t* dist, int32_t* matches) {
#define SD_NW(KK)                                                                                        \
    hipLaunchKernelGGL(sd_nw_pairs<KK>, dim3(grid), dim3(256), 0, st, seq, seg_start, seg_len, n_seg, T, \
                       pair_tmpl, peq, tlen, homo, qmax, reinterpret_cast<uint4*>(hist), dist, matches)
    switch (K) {
        case 1: SD_NW(1); break;
        case 2: SD_NW(2); break;
        case 3: SD_NW(3); break;
        case 4: SD_NW(4); break;
        case 6: SD_NW(6); break;
        default: SD_NW(8); break;
    }
#undef SD_NW
}

}  // namespace sd

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
namespace {
struct NwBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool need(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        cap = want;
        return true;
    }
};
// device buffers of the identity kernel, kept between calls (one context per process; calls serialise)
struct NwCtx {
    std::mutex m;
    int dev = -1;
    NwBuf seq, starts, lens, pair, peq, tlen, hist, dist, matches;
    char* stage = nullptr;      // pinned staging of the text
    size_t stage_cap = 0;
    void release() {
        if (stage) (void)hipHostFree(stage);
        stage = nullptr;
        stage_cap = 0;
        for (NwBuf* b : {&seq, &starts, &lens, &pair, &peq, &tlen, &hist, &dist, &matches}) {
            if (b->p) (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
    }
};
NwCtx g_nw;

inline int nw_code(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'N': return 4;
        default: return -1;
    }
}
}  // namespace

extern "C" void sd_nw_release_cache(void) {
    std::lock_guard<std::mutex> g(g_nw.m);
    if (g_nw.dev >= 0) { (void)hipSetDevice(g_nw.dev); g_nw.release(); }
}

namespace sd {
// Core of the device identity: the text is the concatenation of `spans` (pointer, length); segment s is
// text[seg_start[s] .. seg_start[s] + seg_len[s]).  Templates as given (the caller compresses them for the
// homopolymer form).  Outputs per pair (all-vs-all: s * T + t; pair_tmpl: s).  SD_ERR_UNSUPPORTED for input
// the kernel does not take; nothing is written then.
// stage times of the device identity calls of this process (seconds): host preparation + staging copy, uploads,
// kernel (launch to completion), downloads -- printed with SD_TIMING by the command-line entry points
static std::atomic<long long> g_nw_ns[4];
void nw_stage_seconds(double out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = (double)g_nw_ns[i].load() / 1e9;
}
namespace {
struct NwLap {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void to(int i) {
        const auto n = std::chrono::steady_clock::now();
        g_nw_ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count();
        t = n;
    }
};
}  // namespace

int nw_identity_device(const std::vector<std::pair<const char*, int64_t>>& spans, const int64_t* seg_start,
                       const int32_t* seg_len, int64_t n_seg, const std::vector<std::string>& tmpl,
                       const int32_t* pair_tmpl, bool homo, int device, int threads, int32_t* dist,
                       int32_t* matches) {
    const int T = (int)tmpl.size();
    const int64_t n_pairs = pair_tmpl ? n_seg : n_seg * (int64_t)T;
    if (n_pairs == 0) return SD_OK;
    NwLap lap;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SD_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return SD_ERR_PARAM;
    // templates (homopolymer-compressed when asked, main.py:87-92,137-141) -> match masks
    std::vector<std::string> ts((size_t)T);
    int tmax = 1;
    for (int t = 0; t < T; ++t) {
        const std::string& in = tmpl[(size_t)t];
        std::string& o = ts[(size_t)t];
        for (size_t i = 0; i < in.size(); ++i) {
            if (nw_code(in[i]) < 0) return SD_ERR_UNSUPPORTED;   // outside ACGTN: the host path handles it
            if (!homo || i == 0 || in[i] != in[i - 1]) o.push_back(in[i]);
        }
        tmax = std::max(tmax, (int)o.size());
    }
    if (tmax > 512) return SD_ERR_UNSUPPORTED;
    int K = (tmax + 63) / 64;
    if (K == 5) K = 6;
    if (K == 7) K = 8;
    std::vector<unsigned long long> peq((size_t)T * 5 * K, 0ull);
    std::vector<int32_t> tl((size_t)T);
    for (int t = 0; t < T; ++t) {
        tl[(size_t)t] = (int32_t)ts[(size_t)t].size();
        for (size_t k = 0; k < ts[(size_t)t].size(); ++k)
            peq[((size_t)t * 5 + (size_t)nw_code(ts[(size_t)t][k])) * K + (k >> 6)] |= 1ull << (k & 63);
    }
    int64_t text = 0;
    std::vector<int64_t> span_off(spans.size() + 1, 0);
    for (size_t i = 0; i < spans.size(); ++i) { span_off[i] = text; text += spans[i].second; }
    span_off[spans.size()] = text;
    int qmax = 1;
    for (int64_t s = 0; s < n_seg; ++s) {
        if (seg_len[s] < 0 || seg_start[s] < 0 || seg_start[s] + seg_len[s] > text) return SD_ERR_PARAM;
        if (seg_len[s] > 65000) return SD_ERR_UNSUPPORTED;
        qmax = std::max(qmax, (int)seg_len[s]);
    }
    if (pair_tmpl)
        for (int64_t s = 0; s < n_seg; ++s)
            if (pair_tmpl[s] < 0 || pair_tmpl[s] >= T) return SD_ERR_PARAM;

    std::lock_guard<std::mutex> g(g_nw.m);
    if (hipSetDevice(device) != hipSuccess) return SD_ERR_HIP;
    if (g_nw.dev != device) { if (g_nw.dev >= 0) g_nw.release(); g_nw.dev = device; }
    // text -> pinned staging (alphabet checked on the way) -> device
    if ((size_t)text + 8 > g_nw.stage_cap) {
        if (g_nw.stage) (void)hipHostFree(g_nw.stage);
        g_nw.stage = nullptr;
        g_nw.stage_cap = 0;
        const size_t want = (size_t)text + (size_t)text / 4 + 4096;
        if (hipHostMalloc(reinterpret_cast<void**>(&g_nw.stage), want, hipHostMallocDefault) != hipSuccess) return SD_ERR_HIP;
        g_nw.stage_cap = want;
    }
    {
        // pieces of <= 1 MB so that one long sequence is copied by all threads as well
        struct Piece { size_t span; int64_t off, len; };
        std::vector<Piece> pieces;
        for (size_t i = 0; i < spans.size(); ++i)
            for (int64_t o = 0; o < spans[i].second; o += (1 << 20))
                pieces.push_back(Piece{i, o, std::min<int64_t>(1 << 20, spans[i].second - o)});
        std::vector<uint8_t> bad(pieces.size(), 0);
        sd::parallel_for((int64_t)pieces.size(), threads, 1, [&](int64_t x) {
            const Piece& pc = pieces[(size_t)x];
            const char* src = spans[pc.span].first + pc.off;
            char* dst = g_nw.stage + span_off[pc.span] + pc.off;
            uint8_t b = 0;
            for (int64_t i = 0; i < pc.len; ++i) {
                const char ch = src[i];
                b |= (uint8_t)!(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'N');
                dst[i] = ch;
            }
            bad[(size_t)x] = b;
        });
        for (uint8_t b : bad)
            if (b) return SD_ERR_UNSUPPORTED;
    }
    // resident lanes: up to 16 waves per CU, within a history budget of 8 GB (and a third of the free HBM)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SD_ERR_HIP;
    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t per_lane = (size_t)qmax * K * 16;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return SD_ERR_HIP;
    size_t budget = std::min<size_t>((size_t)8 << 30, (free_b + g_nw.hist.cap) / 3);
    // 16 waves per CU: the kernel is HBM-bound and measured the same from 8 to 32 waves per CU (137-146 M pairs/s),
    // and the history of the resident lanes is what this call allocates (SD_NW_BLOCKS_PER_CU: developer knob)
    const char* lpc = getenv("SD_NW_BLOCKS_PER_CU");
    int64_t lanes = std::min<int64_t>((int64_t)n_cu * (lpc ? std::max(1, atoi(lpc)) : 4) * 256, (int64_t)((n_pairs + 255) / 256 * 256));
    lanes = std::min<int64_t>(lanes, (int64_t)(budget / per_lane) / 256 * 256);
    if (lanes < 256) lanes = 256;
    const int grid = (int)(lanes / 256);
    bool ok = g_nw.seq.need((size_t)text + 8) && g_nw.starts.need(sizeof(int64_t) * (size_t)n_seg) &&
              g_nw.lens.need(sizeof(int32_t) * (size_t)n_seg) && g_nw.peq.need(sizeof(unsigned long long) * peq.size()) &&
              g_nw.tlen.need(sizeof(int32_t) * (size_t)std::max(T, 1)) && g_nw.hist.need((size_t)lanes * per_lane) &&
              g_nw.dist.need(sizeof(int32_t) * (size_t)n_pairs) && g_nw.matches.need(sizeof(int32_t) * (size_t)n_pairs) &&
              (!pair_tmpl || g_nw.pair.need(sizeof(int32_t) * (size_t)n_seg));
    if (!ok) return SD_ERR_HIP;
    lap.to(0);
    auto up = [](void* d, const void* h, size_t n) { return n == 0 || hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; };
    ok = up(g_nw.seq.p, g_nw.stage, (size_t)text) && up(g_nw.starts.p, seg_start, sizeof(int64_t) * (size_t)n_seg) &&
         up(g_nw.lens.p, seg_len, sizeof(int32_t) * (size_t)n_seg) &&
         up(g_nw.peq.p, peq.data(), sizeof(unsigned long long) * peq.size()) &&
         up(g_nw.tlen.p, tl.data(), sizeof(int32_t) * (size_t)T) &&
         (!pair_tmpl || up(g_nw.pair.p, pair_tmpl, sizeof(int32_t) * (size_t)n_seg));
    if (!ok) return SD_ERR_HIP;
    lap.to(1);
    sd::launch_nw_pairs(K, nullptr, grid, static_cast<const uint8_t*>(g_nw.seq.p), static_cast<const int64_t*>(g_nw.starts.p),
                        static_cast<const int32_t*>(g_nw.lens.p), n_seg, T,
                        pair_tmpl ? static_cast<const int32_t*>(g_nw.pair.p) : nullptr,
                        static_cast<const unsigned long long*>(g_nw.peq.p), static_cast<const int32_t*>(g_nw.tlen.p),
                        homo ? 1 : 0, qmax, g_nw.hist.p, static_cast<int32_t*>(g_nw.dist.p),
                        static_cast<int32_t*>(g_nw.matches.p));
    if (hipGetLastError() != hipSuccess) return SD_ERR_HIP;
    lap.to(2);
    if (hipMemcpy(dist, g_nw.dist.p, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(matches, g_nw.matches.p, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost) != hipSuccess)
        return SD_ERR_HIP;
    lap.to(3);
    return SD_OK;
}
}  // namespace sd

extern "C" int sd_identity_segments_dev(const char* seq, int64_t seqlen, const int64_t* starts, const int64_t* ends,
                                        int64_t n_seg, const char* const* tmpl, const int32_t* tlen, int32_t T,
                                        const int32_t* pair_tmpl, int32_t homo, int32_t device, int32_t threads,
                                        int32_t* dist, int32_t* matches, int32_t* columns) {
    if (n_seg < 0 || T < 0 || !seq || (n_seg && (!starts || !ends)) || (T && (!tmpl || !tlen)) || !matches || !columns)
        return SD_ERR_PARAM;
    const int64_t n_pairs = pair_tmpl ? n_seg : n_seg * (int64_t)T;
    if (n_pairs == 0) return SD_OK;
    std::vector<std::string> ts((size_t)T);
    for (int t = 0; t < T; ++t) {
        if (tlen[t] < 0) return SD_ERR_PARAM;
        ts[(size_t)t].assign(tmpl[t], (size_t)tlen[t]);
    }
    // only the text the segments span travels to the device
    std::vector<int64_t> st((size_t)n_seg);
    std::vector<int32_t> ln((size_t)n_seg);
    int64_t lo = seqlen, hi = 0;
    for (int64_t s = 0; s < n_seg; ++s) {
        if (starts[s] < 0 || ends[s] >= seqlen) return SD_ERR_PARAM;
        const int64_t l = std::max<int64_t>(0, ends[s] - starts[s] + 1);
        if (l > 65000) return SD_ERR_UNSUPPORTED;
        ln[(size_t)s] = (int32_t)l;
        if (l > 0) { lo = std::min(lo, starts[s]); hi = std::max(hi, ends[s] + 1); }
    }
    if (hi < lo) { lo = 0; hi = 0; }
    for (int64_t s = 0; s < n_seg; ++s) st[(size_t)s] = ln[(size_t)s] > 0 ? starts[s] - lo : 0;
    std::vector<int32_t> dtmp;
    int32_t* dh = dist;
    if (!dh) { dtmp.resize((size_t)n_pairs); dh = dtmp.data(); }
    std::vector<std::pair<const char*, int64_t>> spans(1, std::make_pair(seq + lo, hi - lo));
    const int rc = sd::nw_identity_device(spans, st.data(), ln.data(), n_seg, ts, pair_tmpl, homo != 0, device, threads, dh, matches);
    if (rc) return rc;
    sd::parallel_for((n_pairs + 65535) / 65536, threads, 1, [&](int64_t blk) {
        const int64_t e = std::min<int64_t>(n_pairs, (blk + 1) * 65536);
        for (int64_t p = blk * 65536; p < e; ++p) columns[p] = dh[p] < 0 ? 0 : dh[p] + matches[p];
    });
    return SD_OK;
}
