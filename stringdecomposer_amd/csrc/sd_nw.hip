// sd_nw.hip -- batched unit-cost global alignment identity on the device (post-processing of the
// drop-in CLI: what stringdecomposer/main.py:29-60 `edist` + `aai` obtain from python-edlib for every
// (block, monomer) pair; with --second-best that is 2*T alignments per output row, main.py:118-146).
//
// The kernel is in sd_nw_kernel.hpp: one lane per (read segment, template) pair, Myers' bit vectors along the
// template, forward pass with a checkpoint of the column state every S columns, then a block-wise recomputation
// whose history stays in registers while edlib's traceback priorities (up > left > diagonal,
// edlib.cpp:945-1150) are walked.  This file holds the launch for segments of an ASCII text and the host driver.
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
#include "sd_nw_kernel.hpp"

namespace sd {

// pairs: all-vs-all (pair_tmpl == nullptr: pair p = segment p / T, template p % T) or one template per segment
// (pair_tmpl[s]); seg_idx (optional) = the segments this launch works on (results are indexed by the true
// segment).  eq_lds: the match masks of all templates fit the dynamic LDS of the launch.
template <int K>
__global__ __launch_bounds__(256, 3) void sd_nw_pairs(const uint8_t* __restrict__ seq,
                                                      const int64_t* __restrict__ seg_start,
                                                      const int32_t* __restrict__ seg_len,
                                                      const int32_t* __restrict__ seg_idx, int64_t n_seg, int T,
                                                      const int32_t* __restrict__ pair_tmpl,
                                                      const unsigned long long* __restrict__ peq,
                                                      const int32_t* __restrict__ tlen, int homo, int cap,
                                                      uint4* __restrict__ ck, int* __restrict__ ckpos, int eq_lds,
                                                      int32_t* __restrict__ dist, int32_t* __restrict__ matches) {
    extern __shared__ unsigned long long speq[];   // [T][5][K] when eq_lds
    if (eq_lds) {
        for (int idx = threadIdx.x; idx < T * 5 * K; idx += blockDim.x) speq[idx] = peq[idx];
        __syncthreads();
    }
    const int64_t n_pairs = pair_tmpl ? n_seg : n_seg * (int64_t)T;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // checkpoints of this lane: [workgroup][slot][word][component][thread] dwords -- a wave writes 256 contiguous
    // bytes per (slot, word, component)
    uint32_t* ckl = reinterpret_cast<uint32_t*>(ck) + (size_t)blockIdx.x * (size_t)cap * K * 4 * 256 + threadIdx.x;
    int* ckp = ckpos + (size_t)blockIdx.x * (size_t)cap * 256 + threadIdx.x;
    for (int64_t p = gid; p < n_pairs; p += stride) {
        const int64_t sl = pair_tmpl ? p : p / T;
        const int64_t s = seg_idx ? seg_idx[sl] : sl;
        const int t = pair_tmpl ? pair_tmpl[s] : (int)(p - sl * T);
        const int64_t o = pair_tmpl ? s : s * T + t;
        const int ql = seg_len[s];
        const int tl = tlen[t];
        if (ql <= 0 || tl <= 0) {  // main.py:30-33: an empty side has no alignment
            dist[o] = -1;
            matches[o] = 0;
            continue;
        }
        NwQueryAscii q{seq, seg_start[s]};
        int d = -2, m = 0;
        // two instantiations: masks in LDS are read with ds_read (see sd_ident.hip)
        if (eq_lds) (void)nw_pair<K>(q, ql, reinterpret_cast<const uint2*>(speq + (size_t)t * 5 * K), tl, homo != 0, ckl, ckp, (size_t)256, cap, d, m);
        else (void)nw_pair<K>(q, ql, reinterpret_cast<const uint2*>(peq + (size_t)t * 5 * K), tl, homo != 0, ckl, ckp, (size_t)256, cap, d, m);
        dist[o] = d;       // -2: more columns than the launch has checkpoint slots for (the host driver sizes them)
        matches[o] = m;
    }
}

void launch_nw_pairs(int K, hipStream_t st, int grid, const uint8_t* seq, const int64_t* seg_start,
                     const int32_t* seg_len, const int32_t* seg_idx, int64_t n_seg, int T, const int32_t* pair_tmpl,
                     const unsigned long long* peq, const int32_t* tlen, int homo, int cap, void* ck, int* ckpos,
                     int32_t* dist, int32_t* matches) {
    const size_t eq_bytes = (size_t)T * 5 * (size_t)K * 8;
    const int eq_lds = eq_bytes <= 60 * 1024 ? 1 : 0;
    const size_t lds = eq_lds ? eq_bytes : 0;
#define SD_NW(KK)                                                                                              \
    hipLaunchKernelGGL(sd_nw_pairs<KK>, dim3(grid), dim3(256), lds, st, seq, seg_start, seg_len, seg_idx, n_seg, T, \
                       pair_tmpl, peq, tlen, homo, cap, reinterpret_cast<uint4*>(ck), ckpos, eq_lds, dist, matches)
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

// match masks of a template set, top-aligned in K words per (template, symbol): symbol k of a template of length
// L sits at bit 64 K - L + k (sd_nw_kernel.hpp)
void nw_build_masks(const std::vector<std::string>& ts, int K, std::vector<unsigned long long>& peq,
                    std::vector<int32_t>& tl) {
    const size_t T = ts.size();
    peq.assign(T * 5 * (size_t)K, 0ull);
    tl.assign(T, 0);
    for (size_t t = 0; t < T; ++t) {
        const int L = (int)ts[t].size();
        tl[t] = L;
        const int pad = 64 * K - L;
        for (int k = 0; k < L; ++k) {
            const char ch = ts[t][(size_t)k];
            const int code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
            const int bit = pad + k;
            peq[(t * 5 + (size_t)code) * K + (size_t)(bit >> 6)] |= 1ull << (bit & 63);
        }
    }
}

}  // namespace sd

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
namespace {
// A buffer that has to grow is NOT freed on the spot: hipFree / hipHostFree wait for every stream of the device, and this
// code runs on the post-processing thread of sd_run_files while the batch pipeline has kernels in flight on three streams --
// on this runtime a hipHostFree issued there never returned (round 6: the text-based identities of a job that followed a
// smaller identity call in the same process hung the job; found by running two tests of the suite on their own).  The old
// block goes on a list that NwCtx::release() empties (sd_release_cache, a change of device); blocks grow by doubling, so
// the list holds less than the current block.
struct NwRetired {
    std::vector<void*> dev, host;
};
NwRetired& nw_retired() { static NwRetired* r = new NwRetired; return *r; }
struct NwBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool need(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) nw_retired().dev.push_back(p);
        const size_t want = std::max(bytes + bytes / 4 + 4096, 2 * cap);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        cap = want;
        return true;
    }
};
// device buffers of the identity kernel, kept between calls (one context per process; calls serialise)
struct NwCtx {
    std::mutex m;
    int dev = -1;
    NwBuf seq, starts, lens, pair, peq, tlen, ck, ckpos, idx, dist, matches;
    char* stage = nullptr;      // pinned staging of the text
    size_t stage_cap = 0;
    void release() {
        if (stage) (void)hipHostFree(stage);
        stage = nullptr;
        stage_cap = 0;
        for (NwBuf* b : {&seq, &starts, &lens, &pair, &peq, &tlen, &ck, &ckpos, &idx, &dist, &matches}) {
            if (b->p) (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
        for (void* q : nw_retired().dev) (void)hipFree(q);
        for (void* q : nw_retired().host) (void)hipHostFree(q);
        nw_retired().dev.clear();
        nw_retired().host.clear();
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
    if (tmax > 2048) return SD_ERR_UNSUPPORTED;
    const bool long_t = tmax > 512;   // monomers of a kilobase: a pair across lanes (sd_nw_long.hip)
    if (long_t && getenv("SD_NW_LONG_OFF")) return SD_ERR_UNSUPPORTED;   // developer A/B: the host threads, as until round 4
    int K = (tmax + 63) / 64;
    if (!long_t) {
        if (K == 5) K = 6;
        if (K == 7) K = 8;
    }
    std::vector<unsigned long long> peq;
    std::vector<int32_t> tl;
    sd::nw_build_masks(ts, K, peq, tl);
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
    if (!long_t) {   // a pair edlib would align by Hirschberg's split (edlib.cpp:1186): the host identities follow it, this kernel does not
        size_t traw = 1;
        for (const std::string& t : tmpl) traw = std::max(traw, t.size());
        if (sd::edlib_splits(qmax, (int64_t)traw)) return SD_ERR_UNSUPPORTED;
    }
    // Long templates: edlib takes Hirschberg's split for ~1.7 kb x 1.7 kb and beyond (20 * ceil(q / 64) * t + 8 t >= 1 MB),
    // which is the common case at 2 kb and rare at 1 kb -- so the pairs are dealt one by one: those edlib walks by its
    // block traceback go to the device, the others (and segments beyond the kernel's LDS room) to the host
    // implementation, which restates the split (sd_post.hip).  (Lengths before compression: a conservative test.)
    constexpr int NWL_QCAP = 8192;
    std::vector<int64_t> dev_pairs, host_pairs;
    int qcap_dev = 1;
    if (long_t) {
        for (int64_t s = 0; s < n_seg; ++s)
            for (int t = pair_tmpl ? pair_tmpl[s] : 0, te = pair_tmpl ? t + 1 : T; t < te; ++t) {
                const int64_t id = pair_tmpl ? s : s * T + t;
                if (seg_len[s] <= 0 || tmpl[(size_t)t].empty()) { dist[id] = -1; matches[id] = 0; continue; }   // main.py:30-33
                if (seg_len[s] > NWL_QCAP || sd::edlib_splits(seg_len[s], (int64_t)tmpl[(size_t)t].size())) host_pairs.push_back(id);
                else { dev_pairs.push_back(id); qcap_dev = std::max(qcap_dev, (int)seg_len[s]); }
            }
    }

    std::lock_guard<std::mutex> g(g_nw.m);
    if (hipSetDevice(device) != hipSuccess) return SD_ERR_HIP;
    if (g_nw.dev != device) { if (g_nw.dev >= 0) g_nw.release(); g_nw.dev = device; }
    // text -> pinned staging (alphabet checked on the way) -> device
    if ((size_t)text + 8 > g_nw.stage_cap) {
        if (g_nw.stage) nw_retired().host.push_back(g_nw.stage);   // (not freed here: see NwRetired)
        const size_t want = std::max((size_t)text + (size_t)text / 4 + 4096, 2 * g_nw.stage_cap);
        g_nw.stage = nullptr;
        g_nw.stage_cap = 0;
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
    if (long_t) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SD_ERR_HIP;
        const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const int lpp = K <= 16 ? 16 : 32, bt = 128, slots = (bt / 64) * (64 / lpp);
        const size_t lds = sd::nw_long_lds_bytes(lpp, qcap_dev, bt);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, ((size_t)160 * 1024) / std::max<size_t>(lds, 1)));
        const int64_t nd = (int64_t)dev_pairs.size();
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nd + slots - 1) / slots, (int64_t)n_cu * per_cu));
        const int cap = sd::nw_long_slots(qcap_dev, K);
        const size_t ck_bytes = (size_t)grid * slots * (size_t)cap * 5 * lpp * 4;
        bool ok = lds <= (size_t)160 * 1024 && g_nw.seq.need((size_t)text + 8) && g_nw.starts.need(sizeof(int64_t) * (size_t)n_seg) &&
                  g_nw.lens.need(sizeof(int32_t) * (size_t)n_seg) && g_nw.peq.need(sizeof(unsigned long long) * peq.size()) &&
                  g_nw.tlen.need(sizeof(int32_t) * (size_t)std::max(T, 1)) && g_nw.ck.need(ck_bytes) &&
                  g_nw.idx.need(sizeof(int64_t) * (size_t)std::max<int64_t>(nd, 1)) && g_nw.dist.need(sizeof(int32_t) * (size_t)n_pairs) &&
                  g_nw.matches.need(sizeof(int32_t) * (size_t)n_pairs) && (!pair_tmpl || g_nw.pair.need(sizeof(int32_t) * (size_t)n_seg));
        if (!ok) return SD_ERR_UNSUPPORTED;
        lap.to(0);
        if (nd > 0) {
            auto up = [](void* d, const void* h, size_t n) { return n == 0 || hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; };
            ok = up(g_nw.seq.p, g_nw.stage, (size_t)text) && up(g_nw.starts.p, seg_start, sizeof(int64_t) * (size_t)n_seg) &&
                 up(g_nw.lens.p, seg_len, sizeof(int32_t) * (size_t)n_seg) && up(g_nw.peq.p, peq.data(), sizeof(unsigned long long) * peq.size()) &&
                 up(g_nw.tlen.p, tl.data(), sizeof(int32_t) * (size_t)T) && up(g_nw.idx.p, dev_pairs.data(), sizeof(int64_t) * (size_t)nd) &&
                 (!pair_tmpl || up(g_nw.pair.p, pair_tmpl, sizeof(int32_t) * (size_t)n_seg));
            if (!ok) return SD_ERR_HIP;
            lap.to(1);
            sd::launch_nw_long(K, nullptr, grid, bt, static_cast<const uint8_t*>(g_nw.seq.p), static_cast<const int64_t*>(g_nw.starts.p),
                               static_cast<const int32_t*>(g_nw.lens.p), static_cast<const int64_t*>(g_nw.idx.p), nd, T,
                               pair_tmpl ? static_cast<const int32_t*>(g_nw.pair.p) : nullptr,
                               static_cast<const unsigned long long*>(g_nw.peq.p), static_cast<const int32_t*>(g_nw.tlen.p),
                               homo ? 1 : 0, qcap_dev, cap, g_nw.ck.p, static_cast<int32_t*>(g_nw.dist.p), static_cast<int32_t*>(g_nw.matches.p));
            if (hipGetLastError() != hipSuccess) return SD_ERR_HIP;
        }
        // the pairs edlib splits: host threads, under the kernel
        int hrc = SD_OK;
        if (!host_pairs.empty()) {
            const size_t nh = host_pairs.size();
            std::vector<std::string> qs(homo ? nh : 0);
            std::vector<const char*> qp(nh), tp(nh);
            std::vector<int32_t> qn(nh), tn(nh), hd(nh), hm(nh), hc(nh);
            for (size_t x = 0; x < nh; ++x) {
                const int64_t id = host_pairs[x], sg = pair_tmpl ? id : id / T;
                const int t = pair_tmpl ? pair_tmpl[sg] : (int)(id - sg * T);
                const char* q = g_nw.stage + seg_start[sg];
                if (homo) {
                    std::string& o = qs[x];
                    for (int32_t i = 0; i < seg_len[sg]; ++i) if (i == 0 || q[i] != q[i - 1]) o.push_back(q[i]);
                    qp[x] = o.data(); qn[x] = (int32_t)o.size();
                } else {
                    qp[x] = q; qn[x] = seg_len[sg];
                }
                tp[x] = ts[(size_t)t].data(); tn[x] = (int32_t)ts[(size_t)t].size();
            }
            hrc = sd_nw_identity_batch(qp.data(), qn.data(), tp.data(), tn.data(), (int64_t)nh, threads, hd.data(), hm.data(), hc.data());
            if (hrc == SD_OK)
                for (size_t x = 0; x < nh; ++x) { dist[host_pairs[x]] = hd[x]; matches[host_pairs[x]] = hm[x]; }
        }
        if (nd > 0) {
            std::vector<int32_t> dd((size_t)n_pairs), dm((size_t)n_pairs);
            if (hipMemcpy(dd.data(), g_nw.dist.p, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(dm.data(), g_nw.matches.p, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost) != hipSuccess)
                return SD_ERR_HIP;
            for (int64_t id : dev_pairs) {
                if (dd[(size_t)id] == -2) return SD_ERR_INTERNAL;
                dist[id] = dd[(size_t)id];
                matches[id] = dm[(size_t)id];
            }
        }
        lap.to(2);
        return hrc == SD_OK ? SD_OK : (hrc == SD_ERR_UNSUPPORTED ? SD_ERR_UNSUPPORTED : hrc);
    }
    // Two launches: the segments of up to NW_SHORT symbols -- practically all of them, a block is about a monomer long --
    // with NW_SHORT / S checkpoint slots per lane and 12 waves per CU, and the few long ones (a non-satellite flank can
    // be a whole chunk) with slots for the longest, on as many lanes as 1 GB of checkpoints allows.  (Round 2 sized the
    // history of EVERY lane by the longest segment of the call: one 5.5-kb block cost 264 KB per lane.)
    constexpr int NW_SHORT = 1024;
    const int S = sd::nw_block_cols(K);
    std::vector<int32_t> idx_short, idx_long;
    for (int64_t s = 0; s < n_seg; ++s) (seg_len[s] > NW_SHORT ? idx_long : idx_short).push_back((int32_t)s);
    if (n_seg > 0x7fffffff) return SD_ERR_UNSUPPORTED;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SD_ERR_HIP;
    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t slot_bytes = (size_t)K * 16 + 4;
    struct Launch { const std::vector<int32_t>* idx; int cap; int grid; size_t idx_off; };
    std::vector<Launch> launches;
    size_t ck_lanes_bytes = 0, ckpos_bytes = 0;
    {
        const char* lpc = getenv("SD_NW_BLOCKS_PER_CU");   // developer knob
        const int bpc = lpc ? std::max(1, atoi(lpc)) : 3;
        auto add = [&](const std::vector<int32_t>& ix, int maxlen, size_t budget, size_t off) {
            if (ix.empty()) return;
            const int cap = std::max(1, (maxlen + S - 1) / S);
            const int64_t pairs = pair_tmpl ? (int64_t)ix.size() : (int64_t)ix.size() * T;
            int64_t lanes = std::min<int64_t>((int64_t)n_cu * bpc * 256, (pairs + 255) / 256 * 256);
            lanes = std::min<int64_t>(lanes, (int64_t)(budget / ((size_t)cap * slot_bytes)) / 256 * 256);
            if (lanes < 256) lanes = 256;
            launches.push_back(Launch{&ix, cap, (int)(lanes / 256), off});
            ck_lanes_bytes = std::max(ck_lanes_bytes, (size_t)lanes * cap * (size_t)K * 16);
            ckpos_bytes = std::max(ckpos_bytes, (size_t)lanes * cap * 4);
        };
        add(idx_short, std::min(qmax, NW_SHORT), (size_t)2 << 30, 0);
        add(idx_long, qmax, (size_t)1 << 30, idx_short.size());
    }
    bool ok = g_nw.seq.need((size_t)text + 8) && g_nw.starts.need(sizeof(int64_t) * (size_t)n_seg) &&
              g_nw.lens.need(sizeof(int32_t) * (size_t)n_seg) && g_nw.peq.need(sizeof(unsigned long long) * peq.size()) &&
              g_nw.tlen.need(sizeof(int32_t) * (size_t)std::max(T, 1)) && g_nw.ck.need(ck_lanes_bytes) &&
              g_nw.ckpos.need(ckpos_bytes) && g_nw.idx.need(sizeof(int32_t) * (size_t)n_seg) &&
              g_nw.dist.need(sizeof(int32_t) * (size_t)n_pairs) && g_nw.matches.need(sizeof(int32_t) * (size_t)n_pairs) &&
              (!pair_tmpl || g_nw.pair.need(sizeof(int32_t) * (size_t)n_seg));
    if (!ok) return SD_ERR_UNSUPPORTED;   // no memory for the device form: the caller's host implementation runs
    lap.to(0);
    auto up = [](void* d, const void* h, size_t n) { return n == 0 || hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; };
    ok = up(g_nw.seq.p, g_nw.stage, (size_t)text) && up(g_nw.starts.p, seg_start, sizeof(int64_t) * (size_t)n_seg) &&
         up(g_nw.lens.p, seg_len, sizeof(int32_t) * (size_t)n_seg) &&
         up(g_nw.peq.p, peq.data(), sizeof(unsigned long long) * peq.size()) &&
         up(g_nw.tlen.p, tl.data(), sizeof(int32_t) * (size_t)T) &&
         up(g_nw.idx.p, idx_short.data(), sizeof(int32_t) * idx_short.size()) &&
         up(static_cast<int32_t*>(g_nw.idx.p) + idx_short.size(), idx_long.data(), sizeof(int32_t) * idx_long.size()) &&
         (!pair_tmpl || up(g_nw.pair.p, pair_tmpl, sizeof(int32_t) * (size_t)n_seg));
    if (!ok) return SD_ERR_HIP;
    lap.to(1);
    for (const Launch& L : launches) {
        sd::launch_nw_pairs(K, nullptr, L.grid, static_cast<const uint8_t*>(g_nw.seq.p), static_cast<const int64_t*>(g_nw.starts.p),
                            static_cast<const int32_t*>(g_nw.lens.p), static_cast<const int32_t*>(g_nw.idx.p) + L.idx_off,
                            (int64_t)L.idx->size(), T, pair_tmpl ? static_cast<const int32_t*>(g_nw.pair.p) : nullptr,
                            static_cast<const unsigned long long*>(g_nw.peq.p), static_cast<const int32_t*>(g_nw.tlen.p),
                            homo ? 1 : 0, L.cap, g_nw.ck.p, static_cast<int*>(g_nw.ckpos.p),
                            static_cast<int32_t*>(g_nw.dist.p), static_cast<int32_t*>(g_nw.matches.p));
        if (hipGetLastError() != hipSuccess) return SD_ERR_HIP;
    }
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
