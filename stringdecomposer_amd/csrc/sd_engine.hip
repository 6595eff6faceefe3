// sd_engine.hip -- host engine + C-ABI of libsd_hip.so (include/sd_hip.h).
//
// Host side of the reference's AlignReadsSet (stringdecomposer/src/main.cpp:67-122): chunk table,
// device batches, ordered gather, per-read assembly, raw TSV.  The DP itself runs only in the HIP
// kernels (sd_generic.hip, sd_fast.hip); there is no CPU implementation of it in this library.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <chrono>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sd_hip.h"
#include "sd_convert.hpp"
#include "sd_ident.hpp"
#include "sd_nw.hpp"
#include "sd_device.hpp"
#include "sd_fast.hpp"
#include "sd_host.hpp"
#include "sd_kernels.hpp"
#include "sd_records.hpp"
#include "sd_seam.hpp"

namespace {

void set_err(char* buf, size_t len, const std::string& m) {
    if (buf && len) {
        std::snprintf(buf, len, "%s", m.c_str());
    }
}

struct HipFail {
    std::string msg;
};

#define SD_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess)                                                               \
            throw HipFail{std::string(#call) + ": " + hipGetErrorString(_e)};               \
    } while (0)

// Process-wide cache of large device buffers.  hipMalloc / hipFree of the multi-GB workspaces
// (checkpoints: ~280 B per chunk row) cost anything from 10 ms to more than a second per call, so
// engines hand their big buffers back to this pool instead of the driver and the next engine (the
// next sd_decompose / chunk-range call of the process) takes them from here.  sd_release_cache()
// returns everything to the driver; SD_DEVICE_POOL=0 disables the cache.
struct DevPool {
    struct Block { int dev; void* p; size_t bytes; };
    std::mutex m;
    std::vector<Block> blocks;
    static constexpr size_t kMin = (size_t)4 << 20;  // smaller buffers are cheap: plain hipMalloc / hipFree
    static bool enabled() {
        static const bool on = [] { const char* e = getenv("SD_DEVICE_POOL"); return !(e && e[0] == '0'); }();
        return on;
    }
    void* take(int dev, size_t bytes, size_t& got) {
        std::lock_guard<std::mutex> g(m);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].dev == dev && blocks[i].bytes >= bytes && blocks[i].bytes <= 2 * bytes + ((size_t)64 << 20) &&
                (best == blocks.size() || blocks[i].bytes < blocks[best].bytes))
                best = i;
        if (best == blocks.size()) return nullptr;
        void* p = blocks[best].p;
        got = blocks[best].bytes;
        blocks.erase(blocks.begin() + (long)best);
        return p;
    }
    void give(int dev, void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(m);
        blocks.push_back(Block{dev, p, bytes});
    }
    void release_all() {
        std::lock_guard<std::mutex> g(m);
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (const Block& b : blocks) {
            (void)hipSetDevice(b.dev);
            (void)hipFree(b.p);
        }
        blocks.clear();
        (void)hipSetDevice(cur);
    }
};
DevPool g_pool;
std::atomic<long long> g_alloc_ns{0};   // time spent in hipMalloc / hipHostMalloc (SD_TIMING report)
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { g_alloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// hipFree and hipHostFree wait for EVERY queue of the process (a barrier packet and a completion handler per hardware queue).
// Issued while batch pipelines had work in flight -- a pinned buffer that had to grow in sd_engine_fetch, a small device
// buffer in a load -- that wait never returned on this runtime once the process held enough queues (AMD_LOG_LEVEL=4: "hsa_amd_
// signal_async_handler() failed to set the handler!" on the last queue, then nothing; round 6, found by running two tests of
// the GPU suite on their own; rounds 1-5 had the same calls).  So nothing is handed back to the runtime on a hot path any
// more: a block that is replaced goes to its pool where one exists, else on this list, which sd_release_cache() empties --
// the caller's statement that the device is idle.  Buffers grow by doubling, so the list stays below what is in use.
struct DeferredFrees {
    std::mutex m;
    std::vector<void*> dev, host;
    void dev_later(void* p) { if (p) { std::lock_guard<std::mutex> g(m); dev.push_back(p); } }
    void host_later(void* p) { if (p) { std::lock_guard<std::mutex> g(m); host.push_back(p); } }
    void drain() {
        std::lock_guard<std::mutex> g(m);
        for (void* p : dev) (void)hipFree(p);
        for (void* p : host) (void)hipHostFree(p);
        dev.clear();
        host.clear();
    }
};
static DeferredFrees& g_deferred_ref() { static DeferredFrees* d = new DeferredFrees; return *d; }
#define g_deferred g_deferred_ref()

// Page-locked blocks that change hands: the identity words of a batch (up to 2 x 84 MB with --second-best) go with
// the batch's rows to the thread that turns them into text, while the engine already fetches the next batch; a freed
// block waits here for the next taker instead of going through hipHostFree / hipHostMalloc (milliseconds per 10 MB).
struct PinPool {
    struct Blk { void* p; size_t bytes; };
    std::mutex m;
    std::vector<Blk> free_;
    void* take(size_t bytes, size_t& got) {
        {
            std::lock_guard<std::mutex> g(m);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].bytes >= bytes && free_[i].bytes <= 2 * bytes + ((size_t)1 << 20) &&   // (no 80-MB block for a 4-byte flag)
                    (best == free_.size() || free_[i].bytes < free_[best].bytes)) best = i;
            if (best < free_.size()) {
                Blk b = free_[best];
                free_.erase(free_.begin() + (long)best);
                got = b.bytes;
                return b.p;
            }
        }
        const size_t want = bytes + bytes / 8 + 4096;
        void* q = nullptr;
        AllocTimer at;
        SD_HIP(hipHostMalloc(&q, want, hipHostMallocDefault));
        got = want;
        return q;
    }
    void give(void* p, size_t bytes) {
        if (!p) return;
        std::lock_guard<std::mutex> g(m);
        free_.push_back(Blk{p, bytes});
        while (free_.size() > 12) {   // keep a dozen; the oldest waits for sd_release_cache (no hipHostFree here: see DeferredFrees)
            g_deferred.host_later(free_.front().p);
            free_.erase(free_.begin());
        }
    }
    void release_all() {
        std::lock_guard<std::mutex> g(m);
        for (Blk& b : free_) (void)hipHostFree(b.p);
        free_.clear();
    }
};
PinPool g_pinpool;

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int dev = 0;
    void free_() {
        if (p) {
            if (DevPool::enabled() && cap * sizeof(T) >= DevPool::kMin) g_pool.give(dev, p, cap * sizeof(T));
            else g_deferred.dev_later(p);   // (small: a few KB to 4 MB; no hipFree on a hot path, see DeferredFrees)
        }
        p = nullptr;
        n = 0;
        cap = 0;
    }
    size_t cap = 0;  // allocated elements (grow-only: batches of similar size reuse the buffer)
    void alloc(size_t count) {
        if (count == 0) count = 1;
        const size_t asked = count;
        if (count > cap) {
            if (cap * sizeof(T) < DevPool::kMin && count < 2 * cap) count = 2 * cap;   // small buffers double (their old blocks wait on a list)
            free_();
            SD_HIP(hipGetDevice(&dev));
            const size_t bytes = count * sizeof(T);
            size_t got = 0;
            void* q = (DevPool::enabled() && bytes >= DevPool::kMin) ? g_pool.take(dev, bytes, got) : nullptr;
            if (q) {
                p = static_cast<T*>(q);
                cap = got / sizeof(T);
            } else {
                AllocTimer at;
                hipError_t er = hipMalloc(reinterpret_cast<void**>(&p), bytes);
                if (er != hipSuccess) {  // give the cached blocks back to the driver and retry once
                    (void)hipGetLastError();
                    g_pool.release_all();
                    SD_HIP(hipMalloc(reinterpret_cast<void**>(&p), bytes));
                }
                cap = count;
            }
        }
        n = asked;
    }
    void upload(const std::vector<T>& h) {
        alloc(h.size());
        if (!h.empty()) SD_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    size_t bytes() const { return cap * sizeof(T); }
    ~DevBuf() { free_(); }
};

// Page-locked host buffer (grow-only): staging for asynchronous H2D / D2H copies.
template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    void alloc(size_t count) {
        if (count == 0) count = 1;
        if (count <= cap) return;
        const size_t want = std::max(count + count / 8, 2 * cap);  // a little slack: batches of similar size reuse it
        free_();
        size_t got = 0;
        p = static_cast<T*>(g_pinpool.take(want * sizeof(T), got));   // (a block another buffer gave up, or a new one)
        cap = got / sizeof(T);
        bytes_ = got;
    }
    size_t bytes_ = 0;
    void free_() {
        if (p) g_pinpool.give(p, bytes_);   // never hipHostFree on a hot path (DeferredFrees)
        p = nullptr;
        cap = 0;
        bytes_ = 0;
    }
    ~PinBuf() { free_(); }
};

}  // namespace

struct sd_engine {
    sd_params p{};
    int device = 0;
    sd::ScoreArgs sc{};      // scores used on the device: the caller's divided by score_scale
    int score_scale = 1;
    // templates (monomers + reverse complements, main.cpp:364-371)
    std::vector<std::string> tseq;
    std::vector<int32_t> tlen, toff;
    int T = 0;
    int64_t sumL = 0;
    int Lmax = 0;
    int family = 0;  // 1 generic, 2 fast

    // generic family
    int Q = 0, threads = 0, rowBytes = 0, n_tiles = 1;
    DevBuf<int32_t> d_estate;        // tiled generic fill: previous row of every resident chunk
    DevBuf<uint16_t> d_grank;        // --ed_thr, generic family: rank table [chunk][T]
    DevBuf<uint8_t> d_tmeta;
    DevBuf<int32_t> d_tend_kd, d_tend_j, d_toff, d_tlen;
    DevBuf<uint8_t> d_ptr;
    size_t ptr_budget = 0;
    std::vector<std::pair<int, int>> subs;  // [begin, end) chunk ranges of the pointer workspace

    // fast family
    sd::FastPlan fplan;
    DevBuf<uint32_t> d_ftable;       // LDS image of the (mm - del) table
    DevBuf<uint32_t> d_flane;        // per-lane constants
    DevBuf<uint32_t> d_fslot;        // (wave, slot, virtual lane) of template cell (j,k) in the lane layout
    DevBuf<uint8_t> d_ftcodes;       // base code of template cell (j,k)
    DevBuf<uint32_t> d_ftr2;         // tables of the packed two-block traceback (FastPlan::tr2_tab)
    DevBuf<long long> d_scanws;      // sd_scan_compact: per-range counts and launch stamps (persist between launches)
    long long scan_epoch = 0, scan_tickets = 0;
    DevBuf<uint32_t> d_fckpt;        // checkpoints
    DevBuf<int32_t> d_fckbase;       // per-checkpoint rebase values
    // --ed_thr prefilter (fast family only)
    DevBuf<unsigned long long> d_peq;
    DevBuf<int32_t> d_endvl, d_endoff, d_dist;
    DevBuf<uint32_t> d_cendoff, d_crank;
    DevBuf<int32_t> d_vlane0;        // --ed_thr, fast family: first virtual lane of each template
    // --ed_thr with more than 128 templates (compacted fill, sd_fast_wn_ck.hip): per chunk the kept templates in
    // filtered order [T], every template's place [T], the kept count; the W chunk classes (by waves needed) and their sizes
    DevBuf<uint16_t> d_klist, d_kpos;
    DevBuf<uint32_t> d_lanet;        // --ed_thr on the tiled layout: [chunk][W * 128] template | part << 16 of every lane (sd_tiled_place)
    DevBuf<int32_t> d_nkept;
    DevBuf<int> d_orders, d_cls;
    bool compact_edthr = false;
    int filter_uniform = -1;         // prefilter: -1 general kernel; 0 / 1 every template ends in the low / high half of the same word
    DevBuf<int> d_guard;             // fp16 range guard of the fills: raised by a wave whose cells left the exact range
    PinBuf<int> h_guard;
    DevBuf<int> d_queue;             // work-queue heads of the persistent kernels: a fresh zeroed (fill, trace) pair per run
    int q_run = 0;                   // pairs handed out since the array was last zeroed
    static constexpr int QN = 2048;
    static constexpr int QS = 16;    // queue heads per run: fill, traceback, then one per further fill class (--ed_thr)
    int n_cu = 256;

    // batch
    std::vector<sd::ChunkDesc> chunks;
    std::vector<int32_t> chunk_read;
    std::vector<int64_t> chunk_off;
    std::vector<int32_t> read_nchunks;
    int32_t n_reads = 0;
    int64_t rows = 0;
    // batch input: one pinned staging buffer and one device buffer, sections [chunk descriptors]
    // [chunk order, longest first][2-bit bases][N mask], one asynchronous H2D copy per load
    PinBuf<uint8_t> h_in;
    DevBuf<uint8_t> d_in;
    sd::ChunkDesc* dp_chunks = nullptr;
    int* dp_order = nullptr;
    uint32_t* dp_bases2 = nullptr;
    uint32_t* dp_nmask = nullptr;
    hipEvent_t ev_in = nullptr;       // the H2D copy of the staging buffer has completed
    bool in_pending = false;
    PinBuf<int64_t> h_roff;           // record offsets of the last run (copied right behind the compaction)
    PinBuf<sd_rec> h_recs;            // compact records of the last fetch
    DevBuf<int32_t> d_B, d_argB, d_cnt;
    DevBuf<sd::DevRec> d_recs, d_dense;
    DevBuf<int64_t> d_roff;
    int64_t dense_cap = 0;
    // in-stream identities of the final TSV (sd_ident.hip), set up by engine_set_identity: 0 off, 1 the record's
    // own template (main.py:112-116), 2 every template, plain and homopolymer-compressed (--second-best)
    int ident_mode = 0;
    int iT = 0, iK = 0, iKh = 0;                 // interleaved templates (m0, m0', m1, ...), words per template
    DevBuf<unsigned long long> d_ipeq, d_ihpeq;  // match masks, plain / compressed templates
    DevBuf<int32_t> d_itlen, d_ihtlen, d_iown;   // lengths; DP template index -> interleaved index (mode 1)
    DevBuf<int32_t> d_recchunk, d_ilong;
    DevBuf<int> d_ilongcnt, d_ickpos;
    DevBuf<uint4> d_ick;
    DevBuf<uint32_t> d_ident, d_identh;
    DevBuf<uint32_t> d_icand;      // pruned homopolymer pass: pairs to align in full (sd_ident.hpp: IdentArgs::cand_list)
    DevBuf<int> d_icandcnt;
    uint32_t* h_ident = nullptr;                 // pinned blocks from g_pinpool, owned until a sink takes them
    uint32_t* h_identh = nullptr;
    size_t h_ident_bytes = 0, h_identh_bytes = 0;
    int64_t ident_cap = 0;                       // records the identity outputs have room for
    bool ident_valid = false;                    // the last fetch brought identities for every record
    sd::IdentArgs ia_plain{}, ia_homo{};
    hipEvent_t ev_id0 = nullptr, ev_id1 = nullptr;
    // Identities in slices: a --second-best batch's identity launches take as long as its DP, and the text of its rows as
    // long again.  With slice_end set (chunk indices, ascending, the last = number of chunks) the identity kernels run once
    // per range of chunks -- the ranges' record bounds are read on the device from the record offsets -- with an event
    // behind each, so that the host fetches, assembles and formats slice s while the device computes slice s + 1: ONE fill
    // and traceback launch for the whole batch (cutting the job into four batches made four under-filled launches: C4's
    // fill 47.6 instead of 22.6 ms) and the hand-over still in pieces.
    std::vector<int> slice_end;
    std::vector<hipEvent_t> ev_slice;
    hipEvent_t ev_dp = nullptr;          // DP + compaction done, record offsets and guard flag on the host
    bool sliced_run = false;             // the last run launched its identities in slices

    // run state
    hipStream_t last_stream = nullptr;
    hipStream_t run_st = nullptr, run_ts = nullptr;   // streams of the last run (a guard trip repeats it on them)
    hipStream_t copy_stream = nullptr;   // pipeline: H2D of the batch / D2H of its records (not owned)
    bool lds_gate = false;               // pipeline mode 2: the fill asks for LDS that admits two workgroups per CU only
    bool ran = false;
    bool replanned = false;              // a guard trip made this engine give up the layout it was created with
    std::vector<hipEvent_t> ev_fill, ev_trace;  // pairs
    hipEvent_t ev_run0 = nullptr, ev_run1 = nullptr, ev_cmp0 = nullptr, ev_cmp1 = nullptr;
    int fill_launches = 0;

    ~sd_engine() {
        for (hipEvent_t e : ev_fill) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_trace) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_run0, ev_run1, ev_cmp0, ev_cmp1, ev_in, ev_id0, ev_id1, ev_dp})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_slice) (void)hipEventDestroy(e);
        g_pinpool.give(h_ident, h_ident_bytes);
        g_pinpool.give(h_identh, h_identh_bytes);
    }

    size_t workspace_bytes() const {
        return d_tmeta.bytes() + d_tend_kd.bytes() + d_tend_j.bytes() + d_ptr.bytes() + d_estate.bytes() + d_grank.bytes() +
               d_ftable.bytes() + d_flane.bytes() + d_fslot.bytes() + d_ftcodes.bytes() + d_fckpt.bytes() +
               d_fckbase.bytes() + d_in.bytes() +
               d_B.bytes() + d_argB.bytes() + d_cnt.bytes() + d_recs.bytes() + d_dense.bytes() +
               d_roff.bytes() + d_recchunk.bytes() + d_ilong.bytes() + d_ick.bytes() + d_ickpos.bytes() +
               d_ident.bytes() + d_identh.bytes() + d_icand.bytes();
    }
};

namespace {

int device_count_checked() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void ensure_events(std::vector<hipEvent_t>& v, size_t pairs) {
    while (v.size() < 2 * pairs) {
        hipEvent_t e;
        SD_HIP(hipEventCreate(&e));
        v.push_back(e);
    }
}

// Developer overrides of sd_params.reserved[] from the environment, read HERE and nowhere else (the switches are part
// of the parameters; the variables exist so that a test or an A/B run can flip one without touching the caller):
// SD_PIPE_MODE=0|1|2, SD_FILL_CELLS=i16, SD_FILL_FULLFLOOR=1, SD_EDTHR_COMPACT=0, SD_FILTER_GENERAL=1, SD_IDENT_STREAM=0, SD_TRACE=1,
// SD_F16_GUARD=<limit>.
void apply_env_overrides(sd_params& p) {
    if (p.reserved[0] == 0)
        if (const char* ev = getenv("SD_PIPE_MODE")) p.reserved[0] = std::max(0, std::min(2, atoi(ev))) + 1;
    auto on = [](const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; };
    if (on("SD_FILL_CELLS", 'i')) p.reserved[1] |= SD_FLAG_NO_F16;
    if (on("SD_FILL_CELLS", 'f')) p.reserved[1] |= SD_FLAG_NO_U16;   // f16: the narrow layout's cells of rounds 1-5 (A/B)
    if (getenv("SD_FILL_FULLFLOOR")) p.reserved[1] |= SD_FLAG_FULL_FLOOR;
    if (on("SD_EDTHR_COMPACT", '0')) p.reserved[1] |= SD_FLAG_NO_EDTHR_COMPACT;
    if (getenv("SD_FILTER_GENERAL")) p.reserved[1] |= SD_FLAG_FILTER_GENERAL;
    if (on("SD_IDENT_STREAM", '0')) p.reserved[1] |= SD_FLAG_NO_STREAM_IDENT;
    if (on("SD_TRACE", '1')) p.reserved[1] |= SD_FLAG_TRACE_V1;
    if (on("SD_IDENT_PRUNE", '0')) p.reserved[1] |= SD_FLAG_NO_IDENT_PRUNE;
    if (p.reserved[2] == 0)
        if (const char* ev = getenv("SD_F16_GUARD")) p.reserved[2] = std::max(0, atoi(ev));
}

int validate_params(const sd_params* p, std::string& err) {
    if (!p) { err = "null params"; return SD_ERR_PARAM; }
    if (p->part_size <= 0) { err = "part_size must be > 0"; return SD_ERR_PARAM; }
    if (p->overlap < 0) { err = "overlap must be >= 0"; return SD_ERR_PARAM; }
    return SD_OK;
}

// Everything must stay above the reference's INF = -1e6 sentinel (main.cpp:156), which makes its
// `> INF` guards vacuous, and below 2^24 so that its float arithmetic on scores is exact.
int check_score_range(const sd_params& p, int Lmax, std::string& err) {
    auto a = [](int v) { return (int64_t)(v < 0 ? -v : v); };
    const int64_t n = (int64_t)p.part_size + p.overlap;
    const int64_t ms = std::max(std::max(a(p.ins), a(p.match)), a(p.mismatch));
    const int64_t bound = n * ms + (int64_t)Lmax * a(p.del) + a(p.mismatch);
    if (bound >= 1000000) {
        err = "scores x chunk length reach the reference's INF sentinel (-1e6): unsupported";
        return SD_ERR_UNSUPPORTED;
    }
    return SD_OK;
}

void build_generic_tables(sd_engine* e) {
    const int Q = e->Q;
    const int64_t need = (e->sumL + Q - 1) / Q;
    int threads = (int)((need + 63) / 64 * 64);
    if (threads < 64) threads = 64;
    e->n_tiles = 1;
    if (threads > 1024) {  // more than 32 768 cells: tiles of 1024 threads x 32 cells, processed in order
        e->n_tiles = (int)((need + 1023) / 1024);
        threads = 1024;
    }
    e->threads = threads;
    const int64_t cells = (int64_t)e->n_tiles * threads * Q;
    e->rowBytes = (int)(cells / 4);
    std::vector<uint8_t> meta((size_t)cells, (uint8_t)(7 | sd::CELL_START));
    std::vector<int32_t> kd((size_t)cells, 0), tj((size_t)cells, 0);
    for (int j = 0; j < e->T; ++j) {
        for (int k = 0; k < e->tlen[j]; ++k) {
            const size_t x = (size_t)e->toff[j] + k;
            uint8_t m = (uint8_t)sd::base_code(e->tseq[j][k]);
            if (k == 0) m |= sd::CELL_START;
            if (k == e->tlen[j] - 1) {
                m |= sd::CELL_END;
                kd[x] = (e->tlen[j] - 1) * e->sc.del;
                tj[x] = j;
            }
            meta[x] = m;
        }
    }
    e->d_tmeta.upload(meta);
    e->d_tend_kd.upload(kd);
    e->d_tend_j.upload(tj);
}

}  // namespace

// Kernel family and layout plan of an engine (sd_params.kernel: 0 auto, 1 generic, 2 fast) and their tables on the
// device.  allow_f16 = false: no fp16 cell format (the fills' range guard tripped, or sd_params.reserved[1] bit 0).
static int engine_pick_family(sd_engine* e, bool allow_f16, std::string& err) {
    const sd_params* p = &e->p;
    int family = p->kernel;
    std::string why;
    const bool no_f16 = !allow_f16 || (p->reserved[1] & SD_FLAG_NO_F16);
    // the packed two-block traceback unless switched off, and not after a range guard tripped (its own check raises the same flag)
    const bool tr2 = allow_f16 && !(p->reserved[1] & SD_FLAG_TRACE_V1);
    const bool fast_ok = sd::fast_plan_build(e->tseq, e->sc, p->part_size + p->overlap, e->fplan, why, !no_f16, tr2,
                                             p->ed_thr > -1 && !(p->reserved[1] & SD_FLAG_NO_EDTHR_COMPACT),
                                             !(p->reserved[1] & SD_FLAG_NO_U16));
    e->fplan.full_floor = (p->reserved[1] & SD_FLAG_FULL_FLOOR) != 0;
    e->sc.rebase_mask = fast_ok ? e->fplan.rebase - 1 : 127;
    if (family == 0) family = fast_ok ? 2 : 1;
    if (family == 2 && !fast_ok) { err = "fast kernel family not applicable: " + why; return SD_ERR_UNSUPPORTED; }
    if (family != 1 && family != 2) { err = "bad kernel family"; return SD_ERR_PARAM; }
    if (p->ed_thr > -1 && e->Lmax > 2048) { err = "--ed_thr supports templates of up to 2048 bp"; return SD_ERR_UNSUPPORTED; }
    if (e->T > 65534) { err = "more than 65534 templates"; return SD_ERR_UNSUPPORTED; }
    e->family = family;
    e->d_toff.upload(e->toff);
    e->d_tlen.upload(e->tlen);
    if (family == 1) {
        e->Q = sd::generic_pick_q(e->sumL);
        build_generic_tables(e);
    } else {
        e->d_ftable.upload(e->fplan.table);
        e->d_flane.upload(e->fplan.lane_consts);
        e->d_fslot.upload(e->fplan.slot_of);
        e->d_ftcodes.upload(e->fplan.tcodes);
        if (e->fplan.tr2_ok) e->d_ftr2.upload(e->fplan.tr2_tab);
        if (p->ed_thr > -1) {
            e->d_endvl.upload(e->fplan.end_vlane);
            e->d_endoff.upload(e->fplan.end_off);
            e->d_vlane0.upload(e->fplan.vlane0);
        }
        // run-time guard of the fp16 cell formats (sd_fast_dev.hpp: F16Guard); reserved[2]: a smaller limit (tests)
        // (biased-u16 cells: the window the plan left around the bias, FastPlan::u16_lim)
        e->sc.guard_lim = p->reserved[2] > 0 ? p->reserved[2] : e->fplan.u16 ? e->fplan.u16_lim : 2040;
        // the same hook lowers the range check of the packed traceback's 16-bit words (its own run-time guard: a
        // checkpoint cell or start term beyond it raises the same flag, and the batch is repeated with sd_fast_trace)
        if (p->reserved[2] > 0 && e->fplan.tr2_ok) e->fplan.tr2_xlim = std::min(e->fplan.tr2_xlim, (int)p->reserved[2]);
        e->d_guard.alloc(1);
        SD_HIP(hipMemset(e->d_guard.p, 0, sizeof(int)));
        e->sc.guard_flag = e->d_guard.p;
    }
    return SD_OK;
}

extern "C" {

void sd_params_default(sd_params* p) {
    std::memset(p, 0, sizeof *p);
    p->ins = -1; p->del = -1; p->mismatch = -1; p->match = 1;
    p->part_size = 5000; p->overlap = 500; p->ed_thr = -1; p->threads = 1; p->device = 0;
    p->kernel = 0;
}

const char* sd_version(void) { return "stringdecomposer_amd 0.1.0 (gfx950)"; }

int sd_device_count(void) { return device_count_checked(); }

void sd_free(void* p) { std::free(p); }

// Host only: the layout the fast kernel family would use for a monomer set and scoring -- what sd_engine_create
// decides before it touches the device.  For tests without a GPU and for users who want to know which kernels a
// set will run on.
int sd_plan_info(const sd_params* p, const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono,
                 int64_t info[8], char* errbuf, size_t errlen) {
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (n_mono <= 0 || !mono_seqs || !mono_lens || !info) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    auto gcd = [](int a, int b) { a = a < 0 ? -a : a; b = b < 0 ? -b : b; while (b) { const int t = a % b; a = b; b = t; } return a; };
    int g = gcd(gcd(p->ins, p->del), gcd(p->mismatch, p->match));
    if (g < 1) g = 1;
    const sd::ScoreArgs sc{p->ins / g, p->del / g, p->mismatch / g, p->match / g};
    std::vector<std::string> tseq((size_t)2 * n_mono);
    for (int j = 0; j < n_mono; ++j) {
        if (mono_lens[j] <= 0) { set_err(errbuf, errlen, "ERROR: empty monomer sequence"); return SD_ERR_EMPTY; }
        tseq[(size_t)j].assign(mono_seqs[j], (size_t)mono_lens[j]);
        rc = sd::check_alphabet("<monomer>", mono_seqs[j], mono_lens[j], err);
        if (rc) { set_err(errbuf, errlen, err); return rc; }
        if (!sd::reverse_complement(tseq[(size_t)j], tseq[(size_t)n_mono + j])) { set_err(errbuf, errlen, "map::at"); return SD_ERR_SYMBOL; }
    }
    sd::FastPlan plan;
    std::string why;
    sd_params pe = *p;
    apply_env_overrides(pe);
    const bool ok = sd::fast_plan_build(tseq, sc, p->part_size + p->overlap, plan, why, !(pe.reserved[1] & SD_FLAG_NO_F16), true,
                                        pe.ed_thr > -1 && !(pe.reserved[1] & SD_FLAG_NO_EDTHR_COMPACT),
                                        !(pe.reserved[1] & SD_FLAG_NO_U16));
    for (int i = 0; i < 8; ++i) info[i] = 0;
    info[0] = ok ? 2 : 1;                       // kernel family "auto" would take: 2 fast, 1 generic
    if (!ok) { set_err(errbuf, errlen, why); return SD_OK; }
    info[1] = plan.P;
    info[2] = plan.tiled ? (plan.f16 ? 6 : 8) : plan.waves > 1 ? (plan.f16 ? 5 : 7) : plan.wide ? (plan.f16 ? 4 : 3) : plan.u16 ? 9 : plan.f16 ? 2 : 1;   // as sd_engine_info [4] >> 8
    info[3] = plan.floor_slots;
    info[4] = plan.waves | ((int64_t)plan.range_bound << 8) | ((int64_t)plan.rebase << 40);
    // narrow layout: cells in the shortest first lane of a template and in the fullest lane (from slot_of)
    int64_t min_first = 1 << 30, max_lane = 0, x = 0;
    for (size_t j = 0; j < tseq.size(); ++j) {
        const int64_t L = (int64_t)tseq[j].size();
        int64_t run = 0, lanes_seen = 0;
        for (int64_t k = 0; k < L; ++k, ++x) {
            const uint32_t so = plan.slot_of[(size_t)x];
            const bool new_lane = k == 0 || (so & 127u) != (plan.slot_of[(size_t)x - 1] & 127u) || (so >> 16) != (plan.slot_of[(size_t)x - 1] >> 16);
            if (new_lane && k > 0) {
                if (lanes_seen == 0) min_first = std::min(min_first, run);
                max_lane = std::max(max_lane, run);
                ++lanes_seen;
                run = 0;
            }
            ++run;
        }
        if (lanes_seen == 0) min_first = std::min(min_first, run);
        max_lane = std::max(max_lane, run);
    }
    info[5] = min_first;
    info[6] = max_lane;
    info[7] = (int64_t)g | ((int64_t)(plan.tr2_ok ? plan.tr2_qm : 0) << 16) | ((int64_t)(plan.tr2_ok ? plan.tr2_bound : 0) << 24) |
              ((int64_t)((plan.Hx >> 8) & 1) << 56);
    return SD_OK;
}

int sd_engine_create(sd_engine** out, const sd_params* p, const char* const* mono_seqs,
                     const int32_t* mono_lens, int32_t n_mono, char* errbuf, size_t errlen) {
    if (!out) return SD_ERR_PARAM;
    *out = nullptr;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (n_mono <= 0) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    std::unique_ptr<sd_engine> e(new sd_engine);
    e->p = *p;
    apply_env_overrides(e->p);
    // A common factor of the four scores scales every DP value, every difference and every tie alike:
    // the device works with the reduced scores (more scorings fit the packed fp16 / int16 cells) and
    // the record scores are multiplied back when they are fetched.
    {
        auto gcd = [](int a, int b) { a = a < 0 ? -a : a; b = b < 0 ? -b : b; while (b) { const int t = a % b; a = b; b = t; } return a; };
        int g = gcd(gcd(p->ins, p->del), gcd(p->mismatch, p->match));
        if (g < 1) g = 1;
        e->score_scale = g;
        e->sc = sd::ScoreArgs{p->ins / g, p->del / g, p->mismatch / g, p->match / g};
    }
    e->T = 2 * n_mono;
    e->tseq.resize((size_t)e->T);
    for (int j = 0; j < n_mono; ++j) {
        if (mono_lens[j] <= 0) { set_err(errbuf, errlen, "ERROR: empty monomer sequence"); return SD_ERR_EMPTY; }
        e->tseq[j].assign(mono_seqs[j], (size_t)mono_lens[j]);
        rc = sd::check_alphabet("<monomer>", mono_seqs[j], mono_lens[j], err);
        if (rc) { set_err(errbuf, errlen, err); return rc; }
        if (!sd::reverse_complement(e->tseq[j], e->tseq[(size_t)n_mono + j])) {
            set_err(errbuf, errlen, "map::at");
            return SD_ERR_SYMBOL;
        }
    }
    e->tlen.resize((size_t)e->T);
    e->toff.resize((size_t)e->T + 1);
    e->toff[0] = 0;
    for (int j = 0; j < e->T; ++j) {
        e->tlen[j] = (int32_t)e->tseq[j].size();
        e->toff[j + 1] = e->toff[j] + e->tlen[j];
        e->Lmax = std::max(e->Lmax, (int)e->tlen[j]);
    }
    e->sumL = e->toff[e->T];
    {
        const int w0 = (e->tlen[0] - 1) >> 6, h0 = ((e->tlen[0] - 1) >> 5) & 1;
        bool same = true;
        for (int j = 1; j < e->T; ++j) same = same && ((e->tlen[j] - 1) >> 6) == w0 && (((e->tlen[j] - 1) >> 5) & 1) == h0;
        e->filter_uniform = same && w0 == ((e->Lmax + 63) / 64) - 1 ? h0 : -1;
        if (e->p.reserved[1] & SD_FLAG_FILTER_GENERAL) e->filter_uniform = -1;
    }
    rc = check_score_range(*p, e->Lmax, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }

    if (device_count_checked() <= 0) {
        set_err(errbuf, errlen, "no HIP device available (libsd_hip has no CPU fallback)");
        return SD_ERR_NO_DEVICE;
    }
    try {
        SD_HIP(hipSetDevice(p->device));
        {
            // Host threads that wait for the device sleep instead of spinning.  A rank of the pipelined path waits
            // ~14 of every 16 ms; spinning, it burns a whole CPU for that (measured: 34.5 -> 21.5 ms of CPU time
            // per 16.3-ms step and rank, same step time), which an 8-GPU node whose ranks share the host CPUs
            // (or a container CPU quota) cannot spare.  SD_HOST_WAIT=spin keeps the runtime's default.
            const char* hw = getenv("SD_HOST_WAIT");
            if (!(hw && hw[0] == 's') && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess)
                (void)hipGetLastError();
        }
        e->device = p->device;
        {
            // (hipGetDeviceProperties fills a 1.5-KB struct from the driver: milliseconds; one attribute is enough)
            static std::atomic<int> cu_of[64];
            int ncu = (p->device >= 0 && p->device < 64) ? cu_of[p->device].load() : 0;
            if (ncu <= 0) {
                SD_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device));
                if (p->device >= 0 && p->device < 64) cu_of[p->device].store(ncu);
            }
            e->n_cu = ncu > 0 ? ncu : 256;
            e->d_queue.alloc(sd_engine::QS * (size_t)sd_engine::QN);
            SD_HIP(hipMemset(e->d_queue.p, 0, sizeof(int) * sd_engine::QS * (size_t)sd_engine::QN));
        }
        rc = engine_pick_family(e.get(), true, err);
        if (rc) { set_err(errbuf, errlen, err); return rc; }
        if (p->ed_thr > -1) {
            std::vector<unsigned long long> peq;
            sd::build_peq(e->tseq, peq);
            e->d_peq.upload(peq);
        }
        SD_HIP(hipEventCreate(&e->ev_run0));
        // the events the host waits on put the waiting thread to sleep (interrupt) instead of spinning: a rank
        // waits ~14 of 16 ms per step, and on a node where the host CPUs are shared by 8 ranks (or capped by a
        // cgroup quota) a spinning waiter per rank takes the time the packers need
        SD_HIP(hipEventCreateWithFlags(&e->ev_run1, hipEventBlockingSync));
        SD_HIP(hipEventCreate(&e->ev_cmp0));
        SD_HIP(hipEventCreate(&e->ev_cmp1));
        SD_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming | hipEventBlockingSync));
    } catch (const HipFail& f) {
        set_err(errbuf, errlen, f.msg);
        return SD_ERR_HIP;
    }
    *out = e.release();
    return SD_OK;
}

void sd_engine_destroy(sd_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    delete e;
}

// In-stream identities (sd_ident.hip).  il_seq = the monomers and their reverse complements interleaved (m0, m0', m1,
// ..., main.py:79-84), own[t] = interleaved index of the DP's template t; second_best = every template, plain and
// homopolymer-compressed, instead of the record's own.  Returns false (mode stays 0: the post-processing computes the
// identities from the read text as before) for template sets the kernel does not take.
static bool engine_set_identity(sd_engine* e, const std::vector<std::string>& il_seq, const std::vector<int32_t>& own,
                                bool second_best) {
    e->ident_mode = 0;
    if (il_seq.empty() || (int)own.size() != e->T) return false;
    auto hpc = [](const std::string& x) {
        std::string o;
        for (size_t i = 0; i < x.size(); ++i)
            if (i == 0 || x[i] != x[i - 1]) o.push_back(x[i]);
        return o;
    };
    std::vector<std::string> hs;
    size_t tmax = 1, hmax = 1;
    for (const std::string& t : il_seq) {
        for (char c : t)
            if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N')) return false;
        if (t.empty()) return false;
        hs.push_back(hpc(t));
        tmax = std::max(tmax, t.size());
        hmax = std::max(hmax, hs.back().size());
    }
    if (tmax > 512) return false;
    auto words = [](size_t L) { int k = (int)((L + 63) / 64); return k == 5 ? 6 : k == 7 ? 8 : k; };
    if ((int64_t)e->p.part_size + e->p.overlap > 60000) return false;   // dist / matches travel as 16-bit fields
    try {
        SD_HIP(hipSetDevice(e->device));
        std::vector<unsigned long long> peq;
        std::vector<int32_t> tl;
        e->iT = (int)il_seq.size();
        e->iK = words(tmax);
        sd::nw_build_masks(il_seq, e->iK, peq, tl);
        e->d_ipeq.upload(peq);
        e->d_itlen.upload(tl);
        e->d_iown.upload(own);
        if (second_best) {
            e->iKh = words(hmax);
            sd::nw_build_masks(hs, e->iKh, peq, tl);
            e->d_ihpeq.upload(peq);
            e->d_ihtlen.upload(tl);
        }
        if (!e->ev_id0) SD_HIP(hipEventCreate(&e->ev_id0));
        if (!e->ev_id1) SD_HIP(hipEventCreate(&e->ev_id1));
    } catch (const HipFail&) {
        return false;
    }
    e->ident_mode = second_best ? 2 : 1;
    return true;
}

// Packs the given chunks (pointer + length each) into the pinned staging buffer, starts their copy to
// the device on `st` (asynchronous) and sizes the per-batch device buffers.  Chunk c refers to
// cptr[c][0 .. clen[c]).  The kernels of sd_engine_run must be enqueued on the same stream (or after
// a synchronisation with it).
// Device buffers of the loaded batch that depend on the kernel family and its layout plan (throws HipFail).
// Separate from the packing / upload of load_chunks_impl because a batch is re-run under another plan when the
// fp16 range guard of the fills trips (engine_rerun_without_f16).
static void engine_alloc_batch(sd_engine* e, int64_t nck) {
    const size_t C = e->chunks.size();
    e->d_B.alloc((size_t)e->rows + C);
    e->d_argB.alloc((size_t)e->rows + C);
    e->d_cnt.alloc(C);
    e->d_roff.alloc(C + 1);
    e->h_roff.alloc(C + 1);
    e->d_recs.alloc((size_t)e->rows);
    e->dense_cap = std::max<int64_t>(4096, e->rows / 16);
    e->d_dense.alloc((size_t)e->dense_cap);
    e->dense_cap = (int64_t)e->d_dense.cap;
    if (e->ident_mode) {
        // identity outputs for up to one record per 48 rows (a block is about a monomer long: ~170 rows); a batch
        // with more records falls back to the text-based identities of the post-processing
        int32_t maxlen = 1;
        for (const sd::ChunkDesc& cd : e->chunks) maxlen = std::max(maxlen, cd.n);
        const int per = e->ident_mode == 2 ? e->iT : 1;
        e->ident_cap = std::min<int64_t>(e->dense_cap, std::max<int64_t>(4096, e->rows / 48));
        e->d_recchunk.alloc((size_t)e->dense_cap);
        e->d_ilong.alloc((size_t)e->ident_cap);
        e->d_ilongcnt.alloc(1);
        e->d_ident.alloc((size_t)e->ident_cap * per);
        if (e->ident_mode == 2) e->d_identh.alloc((size_t)e->ident_cap * per);
        auto fill_args = [&](sd::IdentArgs& a, bool homo) {
            a = sd::IdentArgs{};
            a.chunks = e->dp_chunks; a.bases2 = e->dp_bases2; a.nmask = e->dp_nmask;
            a.dense = e->d_dense.p; a.rec_chunk = e->d_recchunk.p; a.total = e->d_roff.p + C;
            a.rec_cap = e->ident_cap;
            a.T = per; a.own = e->ident_mode == 1 ? e->d_iown.p : nullptr;
            a.peq = homo ? e->d_ihpeq.p : e->d_ipeq.p;
            a.tlen = homo ? e->d_ihtlen.p : e->d_itlen.p;
            a.Tmask = e->iT; a.K = homo ? e->iKh : e->iK; a.homo = homo ? 1 : 0;
            const int S = sd::nw_block_cols(a.K);
            a.short_max = std::min<int>(512, maxlen);
            a.cap_short = (a.short_max + S - 1) / S;
            a.grid_short = e->n_cu * 3;
            a.cap_long = (maxlen + S - 1) / S;
            // the long launch: as many lanes as 192 MB of checkpoints allow, at most one workgroup per CU
            const size_t per_block = (size_t)a.cap_long * 256 * ((size_t)a.K * 16 + 4);
            a.grid_long = (int)std::max<size_t>(1, std::min<size_t>((size_t)e->n_cu, ((size_t)192 << 20) / per_block));
            a.long_cnt = e->d_ilongcnt.p; a.long_list = e->d_ilong.p;
            a.out = homo ? e->d_identh.p : e->d_ident.p;
            // the homopolymer pass in its pruned form (distances, bounds, full alignments of the possible two best only)
            if (homo && per >= 3 && (int64_t)e->ident_cap * per < ((int64_t)1 << 32) && !(e->p.reserved[1] & SD_FLAG_NO_IDENT_PRUNE)) {
                e->d_icand.alloc((size_t)e->ident_cap * per);
                e->d_icandcnt.alloc(1);
                a.cand_list = e->d_icand.p;
                a.cand_cnt = e->d_icandcnt.p;
                a.grid_cand = e->n_cu * 3;
            }
        };
        fill_args(e->ia_plain, false);
        size_t lanes = sd::ident_ck_lanes(e->ia_plain) * (size_t)e->ia_plain.K;
        size_t pos = sd::ident_ck_lanes(e->ia_plain);
        if (e->ident_mode == 2) {
            fill_args(e->ia_homo, true);
            lanes = std::max(lanes, sd::ident_ck_lanes(e->ia_homo) * (size_t)e->ia_homo.K);
            pos = std::max(pos, sd::ident_ck_lanes(e->ia_homo));
        }
        e->d_ick.alloc(lanes);
        e->d_ickpos.alloc(pos);
        e->ia_plain.ck = e->d_ick.p; e->ia_plain.ckpos = e->d_ickpos.p;
        e->ia_homo.ck = e->d_ick.p; e->ia_homo.ckpos = e->d_ickpos.p;
    }
    e->subs.clear();
    if (e->family == 1) {
        // pointer workspace: sub-batches of consecutive chunks within the budget
        size_t free_b = 0, total_b = 0;
        SD_HIP(hipMemGetInfo(&free_b, &total_b));
        size_t budget = std::min<size_t>(free_b / 2 + e->d_ptr.bytes(), (size_t)48 << 30);
        size_t max_sub = 0, cur = 0;
        int begin = 0;
        for (size_t c = 0; c < C; ++c) {
            const size_t need = (size_t)e->chunks[c].n * (size_t)e->rowBytes;
            if (cur + need > budget && cur > 0) {
                e->subs.emplace_back(begin, (int)c);
                max_sub = std::max(max_sub, cur);
                begin = (int)c;
                cur = 0;
            }
            cur += need;
        }
        if (C > 0) { e->subs.emplace_back(begin, (int)C); max_sub = std::max(max_sub, cur); }
        e->d_ptr.alloc(max_sub);
        if (e->n_tiles > 1) {
            size_t most = 0;
            for (const auto& sb : e->subs) most = std::max(most, (size_t)(sb.second - sb.first));
            e->d_estate.alloc(most * (size_t)e->n_tiles * 1024 * 32);
        }
        if (e->p.ed_thr > -1) {
            e->d_dist.alloc(C * (size_t)e->T);
            e->d_grank.alloc(C * (size_t)e->T);
        }
        ensure_events(e->ev_fill, e->subs.size());
        ensure_events(e->ev_trace, e->subs.size());
    } else {
        if (e->p.ed_thr > -1) {
            e->d_dist.alloc(C * (size_t)e->T);
            e->d_cendoff.alloc(C * 64 * (size_t)e->fplan.waves);
            e->d_crank.alloc(C * 64 * (size_t)e->fplan.waves);
            // SD_FLAG_NO_EDTHR_COMPACT (SD_EDTHR_COMPACT=0 arrives as that flag through apply_env_overrides):
            // every chunk on the W-wave ranked kernel (A/B, tests)
            e->compact_edthr = e->fplan.wide && e->fplan.waves > 1 && e->fplan.f16 && !(e->p.reserved[1] & SD_FLAG_NO_EDTHR_COMPACT);
            if (e->compact_edthr && e->fplan.tiled) e->d_lanet.alloc(C * (size_t)e->fplan.waves * 128);
            if (e->compact_edthr) {
                e->d_klist.alloc(C * (size_t)e->T + 2);
                e->d_kpos.alloc(C * (size_t)e->T);
                e->d_nkept.alloc(C);
                e->d_orders.alloc((size_t)e->fplan.waves * C);
                e->d_cls.alloc(8);
            }
        }
        e->d_fckpt.alloc((size_t)nck * (size_t)e->fplan.P * 64 * (size_t)e->fplan.waves);
        e->d_fckbase.alloc((size_t)nck + 1);
        ensure_events(e->ev_fill, 1);
        ensure_events(e->ev_trace, 1);
    }
}

static int load_chunks_impl(sd_engine* e, const std::vector<const char*>& cptr,
                            const std::vector<int32_t>& clen, hipStream_t st, char* errbuf, size_t errlen) {
    e->ran = false;
    e->chunks.clear();
    const size_t C = cptr.size();
    e->chunks.resize(C);
    uint64_t row0 = 0;
    size_t words_total = 0;
    for (size_t c = 0; c < C; ++c) {
        sd::ChunkDesc& cd = e->chunks[c];
        cd = sd::ChunkDesc{};
        cd.woff = (uint32_t)words_total;
        cd.n = clen[c];
        cd.noff = -1;
        cd.row0 = row0;
        row0 += (uint64_t)clen[c];
        words_total += ((size_t)clen[c] + 15) / 16;
    }
    if (words_total >= (1ull << 31)) {
        set_err(errbuf, errlen, "batch too large: split the reads into smaller groups");
        return SD_ERR_UNSUPPORTED;
    }
    e->rows = (int64_t)row0;
    try {
        SD_HIP(hipSetDevice(e->device));
        if (e->in_pending) {  // the previous batch's copy still reads the staging buffer
            SD_HIP(hipEventSynchronize(e->ev_in));
            e->in_pending = false;
        }
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_chunks = 0;
        const size_t o_order = al(o_chunks + C * sizeof(sd::ChunkDesc));
        const size_t o_bases = al(o_order + C * sizeof(int));
        const size_t o_nmask = al(o_bases + words_total * sizeof(uint32_t));
        // worst-case N mask (every chunk has N): one bit per base
        size_t nwords_max = 0;
        for (size_t c = 0; c < C; ++c) nwords_max += ((size_t)clen[c] + 31) / 32;
        e->h_in.alloc(o_nmask + nwords_max * sizeof(uint32_t) + 256);
        uint32_t* bases2 = reinterpret_cast<uint32_t*>(e->h_in.p + o_bases);
        uint32_t* nmask = reinterpret_cast<uint32_t*>(e->h_in.p + o_nmask);
        size_t nwords = 0;
        {
            std::vector<uint8_t> hasn(C, 0);
            sd::parallel_for((int64_t)C, e->p.threads, 16, [&](int64_t c) {
                const sd::ChunkDesc& cd = e->chunks[(size_t)c];
                hasn[(size_t)c] = sd::pack_chunk(cptr[(size_t)c], cd.n, bases2 + cd.woff) ? 1 : 0;
            });
            for (size_t c = 0; c < C; ++c)
                if (hasn[c]) {
                    e->chunks[c].noff = (int32_t)nwords;
                    nwords += ((size_t)e->chunks[c].n + 31) / 32;
                }
            if (nwords) {
                std::memset(nmask, 0, nwords * sizeof(uint32_t));
                sd::parallel_for((int64_t)C, e->p.threads, 16, [&](int64_t c) {
                    const sd::ChunkDesc& cd = e->chunks[(size_t)c];
                    if (cd.noff < 0) return;
                    const char* s = cptr[(size_t)c];
                    for (int32_t i = 0; i < cd.n; ++i)
                        if (s[i] == 'N') nmask[(size_t)cd.noff + (size_t)(i >> 5)] |= 1u << (i & 31);
                });
            }
        }
        const int64_t nck = e->family == 2 ? sd::fast_ckpt_rows_total(e->fplan, e->chunks) : 0;
        if (C) std::memcpy(e->h_in.p + o_chunks, e->chunks.data(), C * sizeof(sd::ChunkDesc));
        {
            // longest chunks first; almost every chunk has the full length, so a counting pass over the
            // two or three distinct lengths would do as well -- the sort is O(C log C) on ints
            int* order = reinterpret_cast<int*>(e->h_in.p + o_order);
            for (size_t c = 0; c < C; ++c) order[c] = (int)c;
            std::stable_sort(order, order + C, [&](int a, int b) { return e->chunks[(size_t)a].n > e->chunks[(size_t)b].n; });
        }
        const size_t in_bytes = o_nmask + nwords * sizeof(uint32_t);
        e->d_in.alloc(o_nmask + nwords_max * sizeof(uint32_t) + 256);
        e->dp_chunks = reinterpret_cast<sd::ChunkDesc*>(e->d_in.p + o_chunks);
        e->dp_order = reinterpret_cast<int*>(e->d_in.p + o_order);
        e->dp_bases2 = reinterpret_cast<uint32_t*>(e->d_in.p + o_bases);
        e->dp_nmask = reinterpret_cast<uint32_t*>(e->d_in.p + o_nmask);
        SD_HIP(hipMemcpyAsync(e->d_in.p, e->h_in.p, in_bytes, hipMemcpyHostToDevice, st));
        SD_HIP(hipEventRecord(e->ev_in, st));
        e->in_pending = true;
        engine_alloc_batch(e, nck);
    } catch (const HipFail& f) {
        set_err(errbuf, errlen, f.msg);
        return SD_ERR_HIP;
    }
    return SD_OK;
}

int sd_engine_load_reads(sd_engine* e, const char* const* read_seqs, const int64_t* read_lens,
                         int32_t n_reads, int64_t* n_chunks, char* errbuf, size_t errlen) {
    if (!e) return SD_ERR_PARAM;
    e->chunk_read.clear();
    e->chunk_off.clear();
    e->read_nchunks.assign((size_t)std::max(n_reads, 0), 0);
    e->n_reads = n_reads;
    // chunk table (main.cpp:70-81)
    std::vector<const char*> cptr;
    std::vector<int32_t> clen;
    for (int32_t r = 0; r < n_reads; ++r) {
        int cnt = sd::chunk_plan(read_lens[r], e->p.part_size, e->p.overlap, [&](int64_t off, int32_t l) {
            cptr.push_back(read_seqs[r] + off);
            clen.push_back(l);
            e->chunk_read.push_back(r);
            e->chunk_off.push_back(off);
        });
        if (cnt == 0) {
            set_err(errbuf, errlen, "ERROR: Sequence #" + std::to_string(r) + " is empty");
            return SD_ERR_EMPTY;
        }
        e->read_nchunks[(size_t)r] = cnt;
    }
    if (n_chunks) *n_chunks = (int64_t)cptr.size();
    // default (null) stream: a later sd_engine_run on any blocking stream is ordered behind the copy;
    // the explicit wait also covers non-blocking streams
    const int rc = load_chunks_impl(e, cptr, clen, nullptr, errbuf, errlen);
    if (rc == SD_OK && hipEventSynchronize(e->ev_in) != hipSuccess) {
        set_err(errbuf, errlen, "H2D copy of the packed reads failed");
        return SD_ERR_HIP;
    }
    e->in_pending = false;
    return rc;
}

// One pass over the loaded batch.  `st` carries the fill (and, for the generic family, everything);
// with a distinct `ts` the traceback + compaction of the fast family go there behind an event, so that
// a pipeline can put the next batch's fill on `st` right behind this one: the traceback of batch b
// then shares the machine with the fill of batch b+1 and runs in the slots its drain leaves free.
// `in_stream` (may be null) is the stream the batch's H2D copy was issued on.
static int engine_run2(sd_engine* e, hipStream_t st, hipStream_t ts, char* errbuf, size_t errlen) {
    const int C = (int)e->chunks.size();
    if (e->family == 1) ts = st;
    try {
        SD_HIP(hipSetDevice(e->device));
        if (e->in_pending) SD_HIP(hipStreamWaitEvent(st, e->ev_in, 0));
        SD_HIP(hipEventRecord(e->ev_run0, st));
        e->fill_launches = 0;
        if (C > 0) {
            if (e->family == 1) {
                const bool ranked = e->p.ed_thr > -1;
                if (ranked)  // main.cpp:91-93: per-chunk template prefilter -> rank table
                    sd::launch_edthr_filter(st, e->dp_chunks, C, e->T, e->Lmax, e->p.ed_thr, e->dp_bases2, e->dp_nmask,
                                            e->d_peq.p, e->d_tlen.p, nullptr, nullptr, e->d_dist.p, nullptr, nullptr,
                                            e->d_grank.p, 1, nullptr, nullptr, nullptr, e->filter_uniform);
                for (size_t s = 0; s < e->subs.size(); ++s) {
                    const int b = e->subs[s].first, n_sub = e->subs[s].second - b;
                    const uint64_t row0_base = e->chunks[(size_t)b].row0;
                    SD_HIP(hipEventRecord(e->ev_fill[2 * s], st));
                    sd::launch_generic_fill(e->Q, e->threads, n_sub, st, e->dp_chunks, b,
                                            e->dp_bases2, e->dp_nmask, e->d_tmeta.p,
                                            e->d_tend_kd.p, e->d_tend_j.p, e->sc, e->rowBytes,
                                            e->d_ptr.p, row0_base, e->d_B.p, e->d_argB.p,
                                            ranked ? e->d_grank.p : nullptr, e->T, e->n_tiles, e->d_estate.p);
                    SD_HIP(hipEventRecord(e->ev_fill[2 * s + 1], st));
                    SD_HIP(hipEventRecord(e->ev_trace[2 * s], st));
                    sd::launch_generic_trace(n_sub, st, e->dp_chunks, b, e->d_ptr.p, row0_base,
                                             e->rowBytes, e->d_B.p, e->d_argB.p, e->d_toff.p,
                                             e->d_tlen.p, e->d_recs.p, e->d_cnt.p);
                    SD_HIP(hipEventRecord(e->ev_trace[2 * s + 1], st));
                    ++e->fill_launches;
                }
            } else {
                const bool ranked = e->p.ed_thr > -1;
                if (e->q_run == sd_engine::QN) {  // every queue head used once: zero them again (no kernel of this
                    SD_HIP(hipMemsetAsync(e->d_queue.p, 0, sizeof(int) * sd_engine::QS * (size_t)sd_engine::QN, st));  // engine is running)
                    e->q_run = 0;
                }
                int* qfill = e->d_queue.p + sd_engine::QS * e->q_run;   // heads of this run: fill, traceback, fill classes 2..8
                int* qtrace = qfill + 1;
                ++e->q_run;
                const bool compact = ranked && e->compact_edthr;
                if (ranked)  // main.cpp:91-93: per-chunk template prefilter
                    sd::launch_edthr_filter(st, e->dp_chunks, C, e->T, e->Lmax, e->p.ed_thr, e->dp_bases2,
                                            e->dp_nmask, e->d_peq.p, e->d_tlen.p, e->d_endvl.p,
                                            e->d_endoff.p, e->d_dist.p, e->d_cendoff.p, e->d_crank.p, nullptr,
                                            e->fplan.waves, compact ? e->d_kpos.p : nullptr,
                                            compact ? e->d_klist.p : nullptr, compact ? e->d_nkept.p : nullptr,
                                            e->filter_uniform, e->d_vlane0.p);
                SD_HIP(hipEventRecord(e->ev_fill[0], st));
                if (compact) {
                    // more than 128 templates: a chunk is filled by as many waves as its kept templates need, holding
                    // exactly those (the point of the reference's prefilter, main.cpp:128-149: less DP work); chunks
                    // that need all W waves by the W-wave ranked kernel.  The class sizes stay on the device.
                    const int W = e->fplan.waves;
                    int* ord = e->d_orders.p;   // [W][C]: class w-1 = the chunks whose kept templates need w waves
                    // tiled layout: the kept templates' lanes per chunk (d_kpos becomes "first lane", d_nkept "lanes used")
                    if (e->fplan.tiled)
                        sd::launch_tiled_place(st, C, e->T, e->fplan.P, W, e->d_klist.p, e->d_nkept.p, e->d_tlen.p,
                                               e->d_kpos.p, e->d_lanet.p, e->fplan.filter_only ? e->d_guard.p : nullptr);
                    sd::launch_split_order(st, e->dp_order, C, e->d_nkept.p, ord, e->d_cls.p, W);
                    for (int w = 1; w < W; ++w)
                        if (e->fplan.tiled)
                            sd::launch_fast_fill_wt_compact(e->fplan, st, e->dp_chunks, e->dp_bases2, e->dp_nmask, e->d_flane.p,
                                                            e->sc, e->d_B.p, e->d_fckpt.p, e->d_fckbase.p,
                                                            w == 1 ? qfill : qfill + 1 + w, ord + (size_t)(w - 1) * C,
                                                            e->d_cls.p + (w - 1), e->n_cu, e->d_lanet.p, e->d_ftcodes.p,
                                                            e->d_toff.p, e->d_tlen.p, w);
                        else
                        sd::launch_fast_fill_wn_compact(e->fplan, st, e->dp_chunks, e->dp_bases2, e->dp_nmask, e->d_flane.p,
                                                        e->sc, e->d_B.p, e->d_fckpt.p, e->d_fckbase.p,
                                                        w == 1 ? qfill : qfill + 1 + w, ord + (size_t)(w - 1) * C,
                                                        e->d_cls.p + (w - 1), e->n_cu, e->d_klist.p, e->d_ftcodes.p,
                                                        e->d_toff.p, e->d_tlen.p, w);
                    if (e->fplan.filter_only)   // no layout of the whole set: the chunks that need all W waves are compacted too
                        sd::launch_fast_fill_wt_compact(e->fplan, st, e->dp_chunks, e->dp_bases2, e->dp_nmask, e->d_flane.p,
                                                        e->sc, e->d_B.p, e->d_fckpt.p, e->d_fckbase.p, qfill + 2,
                                                        ord + (size_t)(W - 1) * C, e->d_cls.p + (W - 1), e->n_cu,
                                                        e->d_lanet.p, e->d_ftcodes.p, e->d_toff.p, e->d_tlen.p, W);
                    else if (e->fplan.tiled)
                        sd::launch_fast_fill_wt(e->fplan, st, e->dp_chunks, C, e->dp_bases2, e->dp_nmask, e->d_ftable.p,
                                                e->d_flane.p, e->sc, e->d_B.p, e->d_fckpt.p, e->d_fckbase.p, qfill + 2,
                                                ord + (size_t)(W - 1) * C, e->n_cu, e->d_cendoff.p, e->d_crank.p,
                                                e->d_cls.p + (W - 1));
                    else
                    sd::launch_fast_fill_wn(e->fplan, st, e->dp_chunks, C, e->dp_bases2, e->dp_nmask, e->d_ftable.p,
                                            e->d_flane.p, e->sc, e->d_B.p, e->d_fckpt.p, e->d_fckbase.p, qfill + 2,
                                            ord + (size_t)(W - 1) * C, e->n_cu, e->d_cendoff.p, e->d_crank.p,
                                            e->d_cls.p + (W - 1));
                } else
                sd::launch_fast_fill(e->fplan, st, e->dp_chunks, C, e->dp_bases2, e->dp_nmask,
                                     e->d_ftable.p, e->d_flane.p, e->sc, e->d_B.p, e->d_argB.p,
                                     e->d_fckpt.p, e->d_fckbase.p, qfill, e->dp_order, e->n_cu,
                                     ranked ? e->d_cendoff.p : nullptr, ranked ? e->d_crank.p : nullptr,
                                     e->lds_gate ? 54 * 1024 : 0);
                SD_HIP(hipEventRecord(e->ev_fill[1], st));
                if (ts != st) SD_HIP(hipStreamWaitEvent(ts, e->ev_fill[1], 0));
                SD_HIP(hipEventRecord(e->ev_trace[0], ts));
                sd::launch_fast_trace(e->fplan, ts, e->dp_chunks, C, e->dp_bases2, e->dp_nmask,
                                      e->d_fslot.p, e->d_ftcodes.p, e->d_flane.p, e->d_toff.p,
                                      e->d_tlen.p, e->sc, e->d_B.p, e->d_argB.p, e->d_fckpt.p,
                                      e->d_fckbase.p, e->d_recs.p, e->d_cnt.p, qtrace, e->dp_order,
                                      e->n_cu, compact ? e->d_klist.p : nullptr, compact ? e->d_kpos.p : nullptr,
                                      compact ? e->d_nkept.p : nullptr, e->fplan.tr2_ok ? e->d_ftr2.p : nullptr,
                                      compact && e->fplan.tiled ? e->d_lanet.p : nullptr);
                SD_HIP(hipEventRecord(e->ev_trace[1], ts));
                e->fill_launches = 1;
            }
            SD_HIP(hipEventRecord(e->ev_cmp0, ts));
            if (!e->d_scanws.p) {
                e->d_scanws.alloc(520);
                SD_HIP(hipMemsetAsync(e->d_scanws.p, 0, 520 * sizeof(long long), ts));
                e->scan_tickets = 0;
            }
            sd::launch_compact(ts, e->dp_chunks, C, e->d_cnt.p, e->d_roff.p, e->d_recs.p,
                               e->d_dense.p, e->dense_cap, true, e->ident_mode ? e->d_recchunk.p : nullptr,
                               e->d_scanws.p, ++e->scan_epoch, &e->scan_tickets);
            SD_HIP(hipEventRecord(e->ev_cmp1, ts));
            auto copy_offsets = [&]() {
                // the record offsets travel right behind the compaction: the fetch then knows the record
                // count as soon as the stream is idle, without a second round trip
                SD_HIP(hipMemcpyAsync(e->h_roff.p, e->d_roff.p, sizeof(int64_t) * ((size_t)C + 1), hipMemcpyDeviceToHost, ts));
                if (e->family == 2) {   // and the fp16 range guard of the fills (reset for the next run behind the copy)
                    e->h_guard.alloc(1);
                    SD_HIP(hipMemcpyAsync(e->h_guard.p, e->d_guard.p, sizeof(int), hipMemcpyDeviceToHost, ts));
                    SD_HIP(hipMemsetAsync(e->d_guard.p, 0, sizeof(int), ts));
                }
            };
            e->sliced_run = e->ident_mode != 0 && !e->slice_end.empty() && e->slice_end.back() == C;
            if (e->sliced_run) {
                copy_offsets();
                if (!e->ev_dp) SD_HIP(hipEventCreateWithFlags(&e->ev_dp, hipEventBlockingSync));
                SD_HIP(hipEventRecord(e->ev_dp, ts));
            }
            if (e->ident_mode) {   // identities of the final TSV on the batch's compact records (sd_ident.hip)
                SD_HIP(hipEventRecord(e->ev_id0, ts));
                e->ia_plain.dense = e->ia_homo.dense = e->d_dense.p;            // (a fetch may have grown them)
                e->ia_plain.rec_chunk = e->ia_homo.rec_chunk = e->d_recchunk.p;
                e->ia_plain.dense_cap = e->ia_homo.dense_cap = e->dense_cap;
                if (e->sliced_run) {
                    while (e->ev_slice.size() < e->slice_end.size()) {
                        hipEvent_t ev;
                        SD_HIP(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
                        e->ev_slice.push_back(ev);
                    }
                    int c_lo = 0;
                    for (size_t sl = 0; sl < e->slice_end.size(); ++sl) {
                        const int c_hi = e->slice_end[sl];
                        e->ia_plain.rec_lo = e->ia_homo.rec_lo = e->d_roff.p + c_lo;
                        e->ia_plain.rec_hi = e->ia_homo.rec_hi = e->d_roff.p + c_hi;
                        sd::launch_ident(ts, e->ia_plain);
                        if (e->ident_mode == 2) sd::launch_ident_pruned(ts, e->ia_homo);
                        SD_HIP(hipEventRecord(e->ev_slice[sl], ts));
                        c_lo = c_hi;
                    }
                } else {
                    e->ia_plain.rec_lo = e->ia_homo.rec_lo = e->ia_plain.rec_hi = e->ia_homo.rec_hi = nullptr;
                    sd::launch_ident(ts, e->ia_plain);
                    if (e->ident_mode == 2) sd::launch_ident_pruned(ts, e->ia_homo);
                }
                SD_HIP(hipEventRecord(e->ev_id1, ts));
            }
            if (!e->sliced_run) copy_offsets();
        }
        SD_HIP(hipEventRecord(e->ev_run1, ts));
        SD_HIP(hipGetLastError());
    } catch (const HipFail& f) {
        // sd_scan_compact takes its range from (device ticket counter - the host's count of tickets handed out): a launch
        // that was rejected, or a kernel that died before every workgroup drew its ticket, leaves the two apart for the
        // life of the workspace -- later launches would then wait on counts nobody publishes.  After any error the
        // workspace is dropped: the next run allocates and zeroes a new one and counts from zero.
        e->d_scanws.free_();
        e->scan_tickets = 0;
        set_err(errbuf, errlen, f.msg);
        return SD_ERR_HIP;
    }
    e->last_stream = ts;
    e->run_st = st;
    e->run_ts = ts;
    e->ran = true;
    return SD_OK;
}

int sd_engine_run(sd_engine* e, void* hip_stream, char* errbuf, size_t errlen) {
    if (!e) return SD_ERR_PARAM;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    return engine_run2(e, st, st, errbuf, errlen);
}

// Waits for the last run and brings its compact records into the pinned buffers h_roff / h_recs
// (valid until the next load / run of this engine).
static std::atomic<long long> g_guard_trips{0};
extern "C" int64_t sd_guard_trips(void) { return (int64_t)g_guard_trips.load(); }

// First half of a fetch: wait until the DP of the last run is done and its record offsets are on the host (a sliced run:
// ev_dp, the identity slices may still be running; else the whole run), repeat the batch if a guard tripped, size the
// host buffer of the records.
static int fetch_begin(sd_engine* e, int64_t& total, char* errbuf, size_t errlen) {
    total = 0;
    if (!e->ran) { set_err(errbuf, errlen, "sd_engine_fetch before sd_engine_run"); return SD_ERR_PARAM; }
    const size_t C = e->chunks.size();
    try {
        SD_HIP(hipSetDevice(e->device));
        SD_HIP(hipEventSynchronize(e->sliced_run ? e->ev_dp : e->ev_run1));
        e->in_pending = false;
        if (C == 0) { e->h_roff.alloc(1); e->h_roff.p[0] = 0; return SD_OK; }
        if (e->family == 2 && e->h_guard.p && e->h_guard.p[0] != 0) {
            // A wave's fp16 cells left the range in which they are exact integers (F16Guard, sd_fast_dev.hpp): the
            // layout plan's bound did not hold for this input.  Nothing of the run is used; the batch (still packed on
            // the device) is repeated with integer cells, which this engine keeps from now on.
            ++g_guard_trips;
            e->replanned = true;
            if (e->sliced_run) SD_HIP(hipEventSynchronize(e->ev_run1));   // (the identity slices of the dropped run)
            std::string err2;
            int rc2 = engine_pick_family(e, false, err2);
            if (rc2) { set_err(errbuf, errlen, "fp16 cell range exceeded, and no integer-cell layout: " + err2); return rc2; }
            const int64_t nck = e->family == 2 ? sd::fast_ckpt_rows_total(e->fplan, e->chunks) : 0;
            SD_HIP(hipMemcpy(e->dp_chunks, e->chunks.data(), C * sizeof(sd::ChunkDesc), hipMemcpyHostToDevice));
            engine_alloc_batch(e, nck);
            rc2 = engine_run2(e, e->run_st, e->run_ts, errbuf, errlen);
            if (rc2) return rc2;
            SD_HIP(hipEventSynchronize(e->ev_run1));
            if (e->family == 2 && e->h_guard.p[0] != 0) {
                set_err(errbuf, errlen, "cell range exceeded in the integer-cell fill");
                return SD_ERR_INTERNAL;
            }
        }
        total = e->h_roff.p[C];
        e->ident_valid = e->ident_mode != 0 && total <= e->ident_cap && total <= e->dense_cap;
        if (total > e->dense_cap) {
            if (e->sliced_run) SD_HIP(hipEventSynchronize(e->ev_run1));
            e->d_dense.alloc((size_t)total);
            e->dense_cap = (int64_t)e->d_dense.cap;
            if (e->ident_mode) e->d_recchunk.alloc((size_t)e->dense_cap);   // a later run of the same load compacts into it
            sd::launch_compact(e->last_stream, e->dp_chunks, (int)C, e->d_cnt.p, e->d_roff.p,
                               e->d_recs.p, e->d_dense.p, e->dense_cap, false);
            SD_HIP(hipStreamSynchronize(e->last_stream));
        }
        e->h_recs.alloc((size_t)std::max<int64_t>(total, 1));
    } catch (const HipFail& f) {
        set_err(errbuf, errlen, f.msg);
        return SD_ERR_HIP;
    }
    return SD_OK;
}

// Second half: the records [r_lo, r_hi) into h_recs (at their own indices) and, if the run has them and the caller gives
// room (pinned, word 0 = record r_lo), their identity words.
static int fetch_range(sd_engine* e, int64_t r_lo, int64_t r_hi, uint32_t* id_dst, uint32_t* idh_dst, char* errbuf, size_t errlen) {
    static_assert(sizeof(sd_rec) == sizeof(sd::DevRec), "record layout");
    const int64_t n = r_hi - r_lo;
    if (n <= 0) return SD_OK;
    try {
        hipStream_t cs = e->copy_stream ? e->copy_stream : e->last_stream;
        SD_HIP(hipMemcpyAsync(e->h_recs.p + r_lo, e->d_dense.p + r_lo, sizeof(sd_rec) * (size_t)n, hipMemcpyDeviceToHost, cs));
        if (e->ident_valid && id_dst) {
            const size_t per = e->ident_mode == 2 ? (size_t)e->iT : 1;
            const size_t nb = sizeof(uint32_t) * (size_t)n * per;
            SD_HIP(hipMemcpyAsync(id_dst, e->d_ident.p + (size_t)r_lo * per, nb, hipMemcpyDeviceToHost, cs));
            if (e->ident_mode == 2 && idh_dst)
                SD_HIP(hipMemcpyAsync(idh_dst, e->d_identh.p + (size_t)r_lo * per, nb, hipMemcpyDeviceToHost, cs));
        }
        SD_HIP(hipStreamSynchronize(cs));
        if (e->score_scale != 1)
            for (int64_t x = r_lo; x < r_hi; ++x) e->h_recs.p[x].score *= e->score_scale;
    } catch (const HipFail& f) {
        set_err(errbuf, errlen, f.msg);
        return SD_ERR_HIP;
    }
    return SD_OK;
}

// the engine's own identity blocks, large enough for `total` records (throws HipFail)
static void engine_grow_ident(sd_engine* e, int64_t total) {
    if (!e->ident_valid || total <= 0) return;
    const size_t nb = sizeof(uint32_t) * (size_t)total * (e->ident_mode == 2 ? (size_t)e->iT : 1);
    auto grow = [&](uint32_t** p, size_t* have) {
        if (*have >= nb) return;
        g_pinpool.give(*p, *have);
        *p = nullptr;
        *have = 0;   // take() may throw: no stale size beside a null pointer
        *p = static_cast<uint32_t*>(g_pinpool.take(nb, *have));
    };
    grow(&e->h_ident, &e->h_ident_bytes);
    if (e->ident_mode == 2) grow(&e->h_identh, &e->h_identh_bytes);
}

// Waits for the last run and brings its compact records (and identities) into the pinned buffers h_roff / h_recs /
// h_ident (valid until the next load / run of this engine).
static int fetch_pinned(sd_engine* e, int64_t& total, char* errbuf, size_t errlen) {
    int rc = fetch_begin(e, total, errbuf, errlen);
    if (rc || e->chunks.empty()) return rc;
    if (e->sliced_run && hipEventSynchronize(e->ev_run1) != hipSuccess) { set_err(errbuf, errlen, "device run failed"); return SD_ERR_HIP; }
    try {
        engine_grow_ident(e, total);
    } catch (const HipFail& f) {
        set_err(errbuf, errlen, f.msg);
        return SD_ERR_HIP;
    }
    return fetch_range(e, 0, total, e->h_ident, e->h_identh, errbuf, errlen);
}

int sd_engine_fetch(sd_engine* e, sd_rec** recs, int64_t** rec_off, char* errbuf, size_t errlen) {
    if (!e || !recs || !rec_off) return SD_ERR_PARAM;
    *recs = nullptr;
    *rec_off = nullptr;
    int64_t total = 0;
    const int rc = fetch_pinned(e, total, errbuf, errlen);
    if (rc) return rc;
    const size_t C = e->chunks.size();
    int64_t* off = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (C + 1)));
    sd_rec* out = static_cast<sd_rec*>(std::malloc(sizeof(sd_rec) * (size_t)std::max<int64_t>(total, 1)));
    if (!off || !out) { std::free(off); std::free(out); set_err(errbuf, errlen, "out of host memory"); return SD_ERR_INTERNAL; }
    std::memcpy(off, e->h_roff.p, sizeof(int64_t) * (C + 1));
    if (total > 0) std::memcpy(out, e->h_recs.p, sizeof(sd_rec) * (size_t)total);
    *recs = out;
    *rec_off = off;
    return SD_OK;
}

int sd_engine_assemble(sd_engine* e, const sd_rec* recs, const int64_t* rec_off, sd_rec** rows,
                       int64_t** row_off, char* errbuf, size_t errlen) {
    (void)errbuf; (void)errlen;
    if (!e || !rows || !row_off) return SD_ERR_PARAM;
    std::vector<sd_rec> all;
    std::vector<int64_t> offs((size_t)e->n_reads + 1, 0);
    std::vector<sd_rec> batch;
    size_t c = 0;
    for (int32_t r = 0; r < e->n_reads; ++r) {
        batch.clear();
        for (int a = 0; a < e->read_nchunks[(size_t)r]; ++a, ++c) {
            const int32_t add = (int32_t)e->chunk_off[c];  // main.cpp:109-111
            for (int64_t x = rec_off[c]; x < rec_off[c + 1]; ++x) {
                sd_rec t = recs[x];
                t.start += add;
                t.end += add;
                batch.push_back(t);
            }
        }
        sd::seam_merge(batch);  // main.cpp:116
        all.insert(all.end(), batch.begin(), batch.end());
        offs[(size_t)r + 1] = (int64_t)all.size();
    }
    sd_rec* o = static_cast<sd_rec*>(std::malloc(sizeof(sd_rec) * std::max<size_t>(all.size(), 1)));
    if (!all.empty()) std::memcpy(o, all.data(), sizeof(sd_rec) * all.size());
    int64_t* ro = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * offs.size()));
    std::memcpy(ro, offs.data(), sizeof(int64_t) * offs.size());
    *rows = o;
    *row_off = ro;
    return SD_OK;
}

int sd_engine_timings(sd_engine* e, float ms[4]) {
    if (!e || !e->ran) return SD_ERR_PARAM;
    ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    if (hipSetDevice(e->device) != hipSuccess) return SD_ERR_HIP;
    if (hipEventSynchronize(e->ev_run1) != hipSuccess) return SD_ERR_HIP;
    const size_t pairs = e->family == 1 ? e->subs.size() : (e->chunks.empty() ? 0 : 1);
    for (size_t s = 0; s < pairs; ++s) {
        float a = 0.f, b = 0.f;
        (void)hipEventElapsedTime(&a, e->ev_fill[2 * s], e->ev_fill[2 * s + 1]);
        (void)hipEventElapsedTime(&b, e->ev_trace[2 * s], e->ev_trace[2 * s + 1]);
        ms[0] += a;
        ms[1] += b;
    }
    if (!e->chunks.empty()) (void)hipEventElapsedTime(&ms[2], e->ev_cmp0, e->ev_cmp1);
    (void)hipEventElapsedTime(&ms[3], e->ev_run0, e->ev_run1);
    return SD_OK;
}

int sd_engine_info(sd_engine* e, int64_t info[8]) {
    if (!e) return SD_ERR_PARAM;
    info[0] = e->T;
    info[1] = e->sumL;
    info[2] = (int64_t)e->chunks.size();
    info[3] = e->rows;
    info[4] = e->family | ((e->family == 1 ? 0 : e->fplan.tiled ? (e->fplan.f16 ? 6 : 8) : e->fplan.waves > 1 ? (e->fplan.f16 ? 5 : 7) : e->fplan.wide ? (e->fplan.f16 ? 4 : 3) : e->fplan.u16 ? 9 : e->fplan.f16 ? 2 : 1) << 8);
    info[5] = e->family == 1 ? e->Q : (e->fplan.P | ((int64_t)e->fplan.floor_slots << 16));
    info[6] = (int64_t)e->workspace_bytes();
    info[7] = (e->family == 1 ? (int64_t)e->subs.size() : 1) |
              ((int64_t)(e->family == 1 ? 0 : (e->fplan.tr2_ok && !e->compact_edthr) ? 2 : 1) << 16);
    return SD_OK;
}

// -------------------------------------------------------------------------------------------
// one-shot entry points
// -------------------------------------------------------------------------------------------
namespace {
struct ReadView {  // borrowed for the duration of the call
    const char* name;
    size_t name_len;
    const char* seq;
    int64_t len;
};
}  // namespace

namespace {
struct CRef { int32_t read; int64_t off; int32_t len; };

// Global chunk table (main.cpp:70-81) of a read set; nch[r] = chunks of read r.
void build_chunk_table(const std::vector<ReadView>& reads, const sd_params* p, std::vector<CRef>& table,
                       std::vector<int32_t>& nch) {
    nch.assign(reads.size(), 0);
    for (size_t r = 0; r < reads.size(); ++r)
        nch[r] = sd::chunk_plan(reads[r].len, p->part_size, p->overlap,
                                [&](int64_t off, int32_t l) { table.push_back(CRef{(int32_t)r, off, l}); });
}

struct TemplateSet {
    std::vector<const char*> mseq;
    std::vector<int32_t> mlen;
    std::vector<std::string> tnames;
    explicit TemplateSet(const std::vector<sd::Seq>& monos) {
        for (const sd::Seq& m : monos) {
            mseq.push_back(m.seq.data());
            mlen.push_back((int32_t)m.seq.size());
            tnames.push_back(m.name);
        }
        for (const sd::Seq& m : monos) tnames.push_back(m.name + "'");  // main.cpp:367
    }
};
}  // namespace

// Device pipeline: up to three batches of chunks in flight on three engines (fills alternate between two streams).
// push() packs a batch into the engine's pinned staging buffer, starts its H2D copy and enqueues its
// kernels (all asynchronous); pop() waits for the oldest batch, brings its records into pinned host
// memory and hands them to that batch's sink.  While the device works on batch b the host packs and
// enqueues b+1 and then assembles b; kernels of consecutive batches sit on different streams, so the
// tail of one launch overlaps the head of the next.
// recs of the chunks [first, first + n) of a batch (word 0 = the first record of chunk `first`), their offsets (n + 1,
// relative to recs); a batch arrives in one call (first = 0) or, with identity slices, in one call per slice
using RecSink = std::function<void(const sd_rec*, const int64_t*, size_t, size_t)>;
namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pipeline {
    static constexpr int NSMAX = 3;
    // Batches in flight.  Three since round 5: with two, the engine of batch b is busy until b's traceback -- which shares
    // the machine with the fill of b+1 at low priority and so ends with it -- has been fetched; only then can b+2 be packed
    // and enqueued, and every second fill ended with nothing but two tracebacks behind it (SD_TIMELINE=1 shows it: 4-5 ms
    // of a 28-ms pair).  With a third engine the next fill is already queued: C2 14.3 -> 13.7 ms per step on the same box
    // (a caller of the stream form gets that with two jobs outstanding before it collects).  SD_PIPE_SLOTS=2: A/B.
    int NS = 3;
    sd_params p{};
    std::vector<const char*> mseq;
    std::vector<int32_t> mlen;
    sd_engine* eng[NSMAX] = {nullptr, nullptr, nullptr};
    hipStream_t copy_st[NSMAX] = {nullptr, nullptr, nullptr};  // per slot: H2D of the batch, D2H of its records
    hipStream_t fill_st = nullptr;                 // fills of all batches, in order
    hipStream_t fill_st2 = nullptr;                // mode 2: fills of the odd batches (see make_streams)
    int mode = 1;
    hipStream_t trace_st = nullptr;                // traceback + compaction of all batches (lower priority)
    bool streams_tried = false;
    RecSink sinks[NSMAX];
    std::function<void(sd_engine*)> on_engine;     // called once for every engine the pipeline creates
    // identities that came with the batch a sink is being called for (in-stream, sd_ident.hip); id == nullptr: none
    // a sink may TAKE the blocks (take_ident: they are then its to give back to g_pinpool): the engine fetches its
    // next batch into other blocks
    // own_*: set when id / idh point INTO shared blocks (identity slices): the holder just drops the references
    struct IdentOut {
        uint32_t* id = nullptr; uint32_t* idh = nullptr; int per = 0; size_t id_bytes = 0, idh_bytes = 0;
        std::shared_ptr<void> own_id, own_idh;
    } cur_ident;
    sd_engine* cur_engine = nullptr;   // the engine whose own blocks cur_ident shows (null: blocks of a slice, owned by pop_fetch)
    IdentOut take_ident() {
        IdentOut o = cur_ident;
        if (cur_engine && o.id) {
            cur_engine->h_ident = nullptr; cur_engine->h_ident_bytes = 0;
            if (o.idh) { cur_engine->h_identh = nullptr; cur_engine->h_identh_bytes = 0; }
        }
        cur_ident = IdentOut{};
        return o;
    }
    uint64_t pushed = 0, popped = 0;
    char eb[1024] = {0};
    // accumulated over all batches: HIP-event kernel times (ms) and host stage times (s)
    double fill_ms = 0, trace_ms = 0, compact_ms = 0, run_ms = 0, ident_ms = 0;
    int64_t ident_pairs = 0;
    double pack_s = 0, wait_s = 0, sink_s = 0;
    int64_t launches = 0, batches = 0, rows = 0;

    const bool timeline = getenv("SD_TIMELINE") != nullptr;   // developer knob, see pop_fetch
    hipEvent_t tl_ref = nullptr;
    double tl_host0 = 0, tl_push0[NSMAX] = {0, 0, 0}, tl_push1[NSMAX] = {0, 0, 0};
    bool restart_idle = false;   // an idle pipeline starts over at slot 0 (see push)
    bool ident_ok = false;   // a cached pipeline's engines carry the identity tables of their job (run_files_impl)
    // a pipeline kept from an earlier job with the same parameters and monomers: new borrowed arrays, fresh counters
    void begin_job(const sd_params* pp, const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono) {
        p.threads = pp->threads;
        for (sd_engine* e : eng) if (e) e->p.threads = pp->threads;
        mseq.assign(mono_seqs, mono_seqs + n_mono);
        mlen.assign(mono_lens, mono_lens + n_mono);
        fill_ms = trace_ms = compact_ms = run_ms = ident_ms = 0;
        ident_pairs = 0;
        pack_s = wait_s = sink_s = 0;
        launches = batches = rows = 0;
        eb[0] = 0;
    }
    int create(const sd_params* pp, const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono) {
        p = *pp;
        apply_env_overrides(p);
        if (const char* ev = getenv("SD_PIPE_SLOTS")) NS = std::min(NSMAX, std::max(1, atoi(ev)));
        mseq.assign(mono_seqs, mono_seqs + n_mono);
        mlen.assign(mono_lens, mono_lens + n_mono);
        const int rc = sd_engine_create(&eng[0], &p, mseq.data(), mlen.data(), n_mono, eb, sizeof eb);
        if (rc == SD_OK && on_engine) on_engine(eng[0]);
        return rc;
    }
    // rows one batch may hold: <= 64 M (~1200 reads of 50 kb, 18 GB of checkpoints) and <= 80 % / NS of the free HBM.
    // The kernels are persistent -- 4096 resident waves pull chunks from a queue -- so a launch is efficient
    // only with a few chunks per wave: batches are kept large (C2's 10 000 chunks are ONE batch; cutting them
    // into 4 x 2 500 costs 1.4x, measured) and overlap comes from pipelining whole batches.  Larger batches
    // would not pay: the launch drain they amortise is hidden by the default stream mode, and multi-ten-GB
    // allocations make a process start slow right after another one released the memory.
    int64_t row_budget() const {
        int64_t budget = (int64_t)64 << 20;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // per row: B + argB + records (24 B) + fast-family checkpoints (P*256 B every FAST_R rows)
            const double per_row = 26.0 + (eng[0]->family == 2 ? eng[0]->fplan.P * 256.0 * eng[0]->fplan.waves / sd::FAST_R : 0.0);
            budget = std::min<int64_t>(budget, (int64_t)(0.8 / NS * (double)(free_b + held_bytes()) / per_row));
            budget = std::max<int64_t>(budget, (int64_t)p.part_size + p.overlap);
        }
        if (const char* ev = getenv("SD_BATCH_ROWS")) { const long long v = atoll(ev); if (v > 0) budget = v; }  // developer A/B
        if (p.max_batch_rows > 0) budget = p.max_batch_rows;  // explicit cap (tests, small GPUs)
        return budget;
    }
    size_t held_bytes() const {
        size_t h = 0;
        for (sd_engine* e : eng) if (e) h += e->workspace_bytes();
        return h;
    }
    // an engine repeated a batch under another layout (fp16 guard trip, filter-only overflow) and kept it
    bool degraded() const {
        for (sd_engine* e : eng) if (e && e->replanned) return true;
        return false;
    }
    int inflight() const { return (int)(pushed - popped); }
    // SD_PIPE_MODE: 0 = every kernel of every batch in order on one stream (clean per-kernel event spans);
    // 1 = fills in order on one stream, traceback + compaction on a second, lower-priority one (the traceback
    // of batch b shares the machine with the fill of batch b+1); 2 (default) = as 1, and consecutive fills sit
    // on two streams without a dependency while the fill asks for enough LDS that only two of its workgroups
    // fit a CU: the next batch's fill moves in workgroup by workgroup as the current one drains instead of
    // waiting for its last wave (C2: 18.6 -> 17.4 ms per step).
    void make_streams() {
        if (streams_tried) return;
        streams_tried = true;
        mode = p.reserved[0] > 0 ? p.reserved[0] - 1 : 2;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least urgent (numerically largest)
        const char* pe = getenv("SD_PIPE_PRIO");
        const bool prio = !(pe && pe[0] == '0');
        const char* ce = getenv("SD_PIPE_COPY");
        if (!(ce && ce[0] == '0'))
            for (int q = 0; q < NS; ++q)
                if (hipStreamCreateWithFlags(&copy_st[q], hipStreamNonBlocking) != hipSuccess) copy_st[q] = nullptr;
        const char* ne = getenv("SD_PIPE_NULL");
        if (ne && ne[0] == '1') return;   // kernels on the null stream
        if (hipStreamCreateWithPriority(&fill_st, hipStreamNonBlocking, prio ? hi : 0) != hipSuccess) fill_st = nullptr;
        const char* te = getenv("SD_PIPE_TRACE_PRIO");   // developer A/B: "hi" = traceback stream as urgent as the fills
        if (mode >= 1 && fill_st &&
            hipStreamCreateWithPriority(&trace_st, hipStreamNonBlocking, prio ? ((te && te[0] == 'h') ? hi : lo) : 0) != hipSuccess)
            trace_st = nullptr;
        if (mode == 2 && fill_st && hipStreamCreateWithPriority(&fill_st2, hipStreamNonBlocking, prio ? hi : 0) != hipSuccess)
            fill_st2 = nullptr;
    }
    // slice_end (may be empty): chunk indices at which the batch's identities are cut into slices (sd_engine::slice_end)
    int push(const std::vector<const char*>& cptr, const std::vector<int32_t>& clen, RecSink sink,
             const std::vector<int>& slice_end = std::vector<int>()) {
        int rc = SD_OK;
        // All slots busy: the oldest batch has to leave its engine first.  Only its device work and the copy of its
        // records are waited for here; its sink (per-read assembly, text) runs AFTER the new batch is packed and
        // enqueued -- the records sit in the engine's pinned buffers, which the new batch does not touch before its own
        // fetch -- so that the device gets its next fill as early as possible (with few host threads the assembly +
        // packing of 5 ms used to end after the running fill's last round had begun: 17.0 instead of 14.6 ms per C2
        // step at two host threads).
        bool deferred = false;
        if (inflight() == NS) { rc = pop_fetch(); deferred = rc == SD_OK; }
        if (rc) return rc;
        struct RunSink { Pipeline* p; bool on; ~RunSink() { if (on) p->pop_sink(); } } run_sink{this, deferred};
        // nothing in flight: start over at slot 0 -- a job of ONE batch then always meets the engine that already holds
        // buffers of its size (alternating slots made every second single-batch job allocate 17 GB anew: 0.5 s)
        // (jobs from files / chunk ranges only: a stream's caller overlaps its jobs, and its second engine should come to
        // life during the caller's warm-up, not when two jobs first overlap)
        if (restart_idle && inflight() == 0 && sink_slot < 0) pushed = popped = 0;
        const int k = (int)(pushed % NS);
        if (!eng[k]) {
            rc = sd_engine_create(&eng[k], &p, mseq.data(), mlen.data(), (int32_t)mseq.size(), eb, sizeof eb);
            if (rc) return rc;
            if (on_engine) on_engine(eng[k]);
        }
        make_streams();
        const double t0 = now_s();
        if (timeline && !tl_ref && fill_st) {
            if (hipEventCreate(&tl_ref) == hipSuccess) {
                (void)hipEventRecord(tl_ref, fill_st);
                (void)hipEventSynchronize(tl_ref);
                tl_host0 = now_s();
            } else tl_ref = nullptr;
        }
        tl_push0[k] = t0;
        eng[k]->copy_stream = copy_st[k];
        rc = load_chunks_impl(eng[k], cptr, clen, copy_st[k] ? copy_st[k] : fill_st, eb, sizeof eb);
        hipStream_t fs = (fill_st2 && (pushed & 1)) ? fill_st2 : fill_st;
        eng[k]->lds_gate = fill_st2 != nullptr;
        eng[k]->slice_end = slice_end;
        if (rc == SD_OK) rc = engine_run2(eng[k], fs, trace_st ? trace_st : fs, eb, sizeof eb);
        pack_s += now_s() - t0;
        tl_push1[k] = now_s();
        if (rc) return rc;
        sinks[k] = std::move(sink);
        ++pushed;
        return SD_OK;
    }
    int pop() {
        const int rc = pop_fetch();
        if (rc == SD_OK) pop_sink();
        return rc;
    }
    int sink_slot = -1;          // slot whose records are fetched and whose sink has not run yet
    RecSink sink_fn;             // ... its sink, its record offsets (a copy: the engine's pinned array is the target of the
    std::vector<int64_t> sink_roff;   // next run's copy) and its chunk count
    size_t sink_chunks = 0;
    // first half of pop(): wait for the oldest batch, copy its records to the host, book its times.  A batch whose
    // identities run in slices is handed to its sink here, slice by slice, as the slices complete on the device.
    int pop_fetch() {
        if (sink_slot >= 0) pop_sink();
        if (inflight() == 0) return SD_OK;
        const int k = (int)(popped % NS);
        sd_engine* e = eng[k];
        int64_t total = 0;
        double t0 = now_s();
        int rc = fetch_begin(e, total, eb, sizeof eb);
        const bool sliced = rc == SD_OK && e->sliced_run && e->ident_valid && !e->chunks.empty();
        if (rc == SD_OK && !sliced && !e->chunks.empty()) {
            if (e->sliced_run && hipEventSynchronize(e->ev_run1) != hipSuccess) { std::snprintf(eb, sizeof eb, "device run failed"); rc = SD_ERR_HIP; }
            if (rc == SD_OK) {
                try {
                    engine_grow_ident(e, total);
                } catch (const HipFail& f) {
                    std::snprintf(eb, sizeof eb, "%s", f.msg.c_str());
                    rc = SD_ERR_HIP;
                }
            }
            if (rc == SD_OK) rc = fetch_range(e, 0, total, e->h_ident, e->h_identh, eb, sizeof eb);
        }
        wait_s += now_s() - t0;
        ++popped;
        if (rc) { sinks[k] = nullptr; return rc; }
        launches += e->fill_launches;
        ++batches;
        rows += e->rows;
        if (sliced) {
            RecSink fn = std::move(sinks[k]);
            sinks[k] = nullptr;
            // The identity words of the whole batch land in ONE pair of pinned blocks, slice by slice; every slice's sink
            // gets a reference (IdentOut::own_*) and the blocks go back to the pool when the last one lets go (a block per
            // slice meant 16 hipHostMalloc / hipHostFree of 19 MB per job: 100 ms).
            const size_t per = e->ident_mode == 2 ? (size_t)e->iT : 1;
            const size_t nb = sizeof(uint32_t) * (size_t)std::max<int64_t>(total, 1) * per;
            std::shared_ptr<void> own_id, own_idh;
            try {
                size_t got = 0;
                void* q = g_pinpool.take(nb, got);
                own_id.reset(q, [got](void* x) { g_pinpool.give(x, got); });
                if (e->ident_mode == 2) {
                    q = g_pinpool.take(nb, got);
                    own_idh.reset(q, [got](void* x) { g_pinpool.give(x, got); });
                }
            } catch (const HipFail& f) {
                std::snprintf(eb, sizeof eb, "%s", f.msg.c_str());
                rc = SD_ERR_HIP;
            }
            int c_lo = 0;
            std::vector<int64_t> ro;
            for (size_t sl = 0; sl < e->slice_end.size() && rc == SD_OK; ++sl) {
                const int c_hi = e->slice_end[sl];
                const int64_t r_lo = e->h_roff.p[c_lo], r_hi = e->h_roff.p[c_hi];
                t0 = now_s();
                uint32_t* idp = static_cast<uint32_t*>(own_id.get()) + (size_t)r_lo * per;
                uint32_t* idhp = own_idh ? static_cast<uint32_t*>(own_idh.get()) + (size_t)r_lo * per : nullptr;
                if (hipEventSynchronize(e->ev_slice[sl]) != hipSuccess) { std::snprintf(eb, sizeof eb, "device run failed"); rc = SD_ERR_HIP; }
                if (rc == SD_OK) rc = fetch_range(e, r_lo, r_hi, idp, idhp, eb, sizeof eb);
                wait_s += now_s() - t0;
                if (rc == SD_OK) {
                    t0 = now_s();
                    ro.resize((size_t)(c_hi - c_lo) + 1);
                    for (int c = c_lo; c <= c_hi; ++c) ro[(size_t)(c - c_lo)] = e->h_roff.p[c] - r_lo;
                    cur_engine = nullptr;
                    cur_ident = IdentOut{};
                    if (r_hi > r_lo) {
                        cur_ident.id = idp; cur_ident.idh = idhp; cur_ident.per = (int)per;
                        cur_ident.own_id = own_id; cur_ident.own_idh = own_idh;
                    }
                    if (fn) fn(e->h_recs.p + r_lo, ro.data(), (size_t)c_lo, (size_t)(c_hi - c_lo));
                    cur_ident = IdentOut{};
                    sink_s += now_s() - t0;
                }
                c_lo = c_hi;
            }
            (void)hipEventSynchronize(e->ev_run1);
        }
        float ms[4];
        if (sd_engine_timings(e, ms) == SD_OK) { fill_ms += ms[0]; trace_ms += ms[1]; compact_ms += ms[2]; run_ms += ms[3]; }
        if (timeline && tl_ref && e->family == 2 && !e->chunks.empty()) {
            // developer knob SD_TIMELINE=1: where each kernel of the batch began and ended on the DEVICE clock (ms since the
            // pipeline's reference event) next to the host's clock for its enqueue and fetch -- shows whether the device waited
            float f0 = 0, f1 = 0, t0e = 0, t1e = 0, c1 = 0;
            (void)hipEventElapsedTime(&f0, tl_ref, e->ev_fill[0]);
            (void)hipEventElapsedTime(&f1, tl_ref, e->ev_fill[1]);
            (void)hipEventElapsedTime(&t0e, tl_ref, e->ev_trace[0]);
            (void)hipEventElapsedTime(&t1e, tl_ref, e->ev_trace[1]);
            (void)hipEventElapsedTime(&c1, tl_ref, e->ev_cmp1);
            std::fprintf(stderr, "[sd timeline] batch %llu slot %d: device fill %.2f-%.2f trace %.2f-%.2f compact end %.2f | host enqueue %.2f-%.2f fetch done %.2f\n",
                         (unsigned long long)popped, k, f0, f1, t0e, t1e, c1, (tl_push0[k] - tl_host0) * 1e3, (tl_push1[k] - tl_host0) * 1e3,
                         (now_s() - tl_host0) * 1e3);
        }
        if (e->ident_mode && !e->chunks.empty()) {
            float im = 0.f;
            if (hipEventElapsedTime(&im, e->ev_id0, e->ev_id1) == hipSuccess) ident_ms += im;
            if (e->ident_valid) ident_pairs += total * (e->ident_mode == 2 ? 2 * (int64_t)e->iT : 1);
        }
        if (sliced) return rc;
        sink_slot = k;
        sink_fn = std::move(sinks[k]);
        sinks[k] = nullptr;
        sink_chunks = e->chunks.size();
        sink_roff.assign(e->h_roff.p, e->h_roff.p + sink_chunks + 1);
        return SD_OK;
    }
    // second half: hand the fetched records to the batch's sink
    void pop_sink() {
        if (sink_slot < 0) return;
        const int k = sink_slot;
        sink_slot = -1;
        const double t0 = now_s();
        cur_ident = IdentOut{};
        cur_engine = eng[k];
        if (eng[k]->ident_valid)
        {
            cur_ident.id = eng[k]->h_ident;
            cur_ident.idh = eng[k]->ident_mode == 2 ? eng[k]->h_identh : nullptr;
            cur_ident.per = eng[k]->ident_mode == 2 ? eng[k]->iT : 1;
            cur_ident.id_bytes = eng[k]->h_ident_bytes;
            cur_ident.idh_bytes = eng[k]->ident_mode == 2 ? eng[k]->h_identh_bytes : 0;
        }
        if (sink_fn) sink_fn(eng[k]->h_recs.p, sink_roff.data(), 0, sink_chunks);
        sink_fn = nullptr;
        cur_ident = IdentOut{};
        cur_engine = nullptr;
        sink_s += now_s() - t0;
    }
    int drain() {
        int rc = SD_OK;
        while (inflight() > 0) {
            const int r2 = pop();
            if (r2 && !rc) rc = r2;
        }
        return rc;
    }
    ~Pipeline() {
        if (inflight() > 0) (void)hipDeviceSynchronize();  // nothing may still run on buffers we free
        for (sd_engine* e : eng)
            if (e) sd_engine_destroy(e);
        for (hipStream_t s2 : {copy_st[0], copy_st[1], copy_st[2], fill_st, fill_st2, trace_st})
            if (s2) (void)hipStreamDestroy(s2);
    }
};

// Cuts the chunks [c_lo, c_hi) of a table into device batches of consecutive chunks: at most `budget`
// rows each, and at least `min_batches` batches (when there are that many chunks) of about equal rows.
void plan_batches(const std::vector<CRef>& table, size_t c_lo, size_t c_hi, int64_t budget, int min_batches,
                  std::vector<std::pair<size_t, size_t>>& out) {
    out.clear();
    {
        // equal shares: the last batch of a job must not be a small remainder (a launch with less than one
        // chunk per resident wave takes as long as a full round)
        int64_t tot = 0;
        for (size_t c = c_lo; c < c_hi; ++c) tot += table[c].len;
        const int64_t nb = std::max<int64_t>(std::max(min_batches, 1), (tot + budget - 1) / std::max<int64_t>(budget, 1));
        const int64_t lmax = c_lo < c_hi ? table[c_lo].len : 1;   // slack of one chunk: shares need not split evenly
        budget = std::min<int64_t>(budget, std::max<int64_t>(1, (tot + nb - 1) / nb + (nb > 1 ? lmax : 0)));
    }
    for (size_t c0 = c_lo; c0 < c_hi;) {
        int64_t rows = 0;
        size_t c1 = c0;
        while (c1 < c_hi && (c1 == c0 || rows + table[c1].len <= budget)) rows += table[c1++].len;
        out.emplace_back(c0, c1);
        c0 = c1;
    }
}
// Rows per batch for a pipeline whose engines do not exist yet (the first job of a process, or of a parameter set).
// Such a job pays for every byte it allocates -- the driver scrubs memory before it hands it out, ~29 ms per GB on the
// GPU box: the three full-size engines of a 500-Mbp job (50 GB) cost 1.45 s for 0.14 s of device work.  Buffers scale
// with the rows of a batch, launches get less efficient below two rounds of the persistent kernels (C2 per 50 Mbp:
// 13.7 ms in batches of 10 000 chunks, 14.0 at 5 000, 15.0 at 2 500, 22.7 at 1 250), so between "one batch" and "many
// full batches" a job is cut into batches of 14.7 M rows (2 670 chunks of 5.5 kb: 4.4 GB per engine), a job of less than
// eight of those into eight (down to 7 M rows), and only a job beyond 3 G rows (15 Gbp on this GPU) takes 29 M-row
// batches, beyond 20 G rows full ones.  A pipeline from the cache has its buffers and takes `budget` as it is.
// (SD_FRESH_ROWS: developer A/B, 0 = off.)
int64_t fresh_row_budget(int64_t budget, int64_t job_rows) {
    int64_t cap = job_rows > 20000000000ll ? budget : job_rows > 3000000000ll ? ((int64_t)28 << 20) : ((int64_t)14 << 20);
    if (const char* ev = getenv("SD_FRESH_ROWS")) { const long long v = atoll(ev); if (v > 0) cap = v; else return budget; }
    if (job_rows <= cap) return std::min(budget, cap);                 // (a job of one batch: the callers' own rules)
    return std::min(budget, std::max<int64_t>(cap / 2, std::min(cap, job_rows / 8)));
}
}  // namespace

// Pipelines of finished jobs (sd_run_files, the chunk-range calls), kept for the next job with the same parameters and monomer set: creating
// the engines (layout plan, tables, identity masks, streams, events, pinned staging) is 25-40 ms per call, a quarter of
// a C4 --second-best job.  A process that decomposes many read sets against one monomer set (a service behind the
// C-ABI, bench.py's steps) pays it once; sd_release_cache() drops them.  At most two are kept, none that holds more
// than SD_PIPE_CACHE_GB (default 96) GB of device memory, and none whose engines left the layout they were created
// with (a tripped fp16 guard, an overflowing filter-only batch: the next job should start from the plan again).
// The entries are never destroyed at process exit (as g_pool / g_pinpool: the HIP runtime may be gone by then);
// sd_release_cache() is the only place that tears them down.
extern "C++" {
namespace {
struct PipeCacheEntry { std::string key; std::unique_ptr<Pipeline> pipe; };
struct PipeCache { std::mutex m; std::vector<PipeCacheEntry> v; };
PipeCache& pipe_cache() { static PipeCache* c = new PipeCache; return *c; }   // leaked on purpose
// the cache key, field by field (the raw bytes of an sd_params would carry its padding and the host-thread count)
std::string pipe_cache_key(const sd_params& pe, char kind, const std::vector<const char*>& mseq, const std::vector<int32_t>& mlen) {
    std::string k;
    for (int32_t v : {pe.ins, pe.del, pe.mismatch, pe.match, pe.part_size, pe.overlap, pe.ed_thr, pe.device, pe.kernel,
                      pe.max_batch_rows, pe.reserved[0], pe.reserved[1], pe.reserved[2], pe.reserved[3], pe.reserved[4]}) {
        k += std::to_string(v);
        k.push_back(',');
    }
    k.push_back(kind);
    for (size_t m = 0; m < mseq.size(); ++m) { k.append(mseq[m], (size_t)mlen[m]); k.push_back('\n'); }
    return k;
}
std::unique_ptr<Pipeline> pipe_cache_take(const std::string& key) {
    PipeCache& c = pipe_cache();
    std::lock_guard<std::mutex> g(c.m);
    for (size_t i = 0; i < c.v.size(); ++i)
        if (c.v[i].key == key) {
            std::unique_ptr<Pipeline> q = std::move(c.v[i].pipe);
            c.v.erase(c.v.begin() + (long)i);
            return q;
        }
    return nullptr;
}
void pipe_cache_give(const std::string& key, std::unique_ptr<Pipeline> q) {
    static const size_t cap = [] { const char* e = getenv("SD_PIPE_CACHE_GB"); return (size_t)(e ? std::max(0, atoi(e)) : 96) << 30; }();
    if (!q || q->degraded() || q->held_bytes() > cap) return;   // (destroyed here, outside the lock)
    std::vector<PipeCacheEntry> drop;   // destroyed outside the lock
    {
        PipeCache& c = pipe_cache();
        std::lock_guard<std::mutex> g(c.m);
        c.v.push_back(PipeCacheEntry{key, std::move(q)});
        while (c.v.size() > 2) { drop.push_back(std::move(c.v.front())); c.v.erase(c.v.begin()); }
    }
}
void pipe_cache_clear() {
    std::vector<PipeCacheEntry> drop;
    {
        PipeCache& c = pipe_cache();
        std::lock_guard<std::mutex> g(c.m);
        drop.swap(c.v);
    }
}
}  // namespace
}  // extern "C++"

// Runs the chunks [c_lo, c_hi) of `table` through the device in batches of consecutive chunks sized to
// the free HBM, so a single 200-Mb sequence and a million reads take the same path; the records of
// every batch go to `sink(c0, c1, recs, rec_off)` in table order (chunk-local coordinates, rec_off
// relative to the batch).
using BatchSink = std::function<void(size_t, size_t, const sd_rec*, const int64_t*)>;
// `while_busy` (may be empty) runs on the calling thread once the last batch is enqueued, i.e. while the device
// works: the chunk-range calls check their share's alphabet there instead of before the first upload (a bad symbol
// still fails the call -- the records of the run are dropped -- it just no longer delays the device by the 0.2 ms the
// check of 27 Mb takes; the packer maps any byte to a 2-bit code, so the kernels run on whatever the bytes are).
static int run_chunk_batches(const std::vector<ReadView>& reads, const std::vector<CRef>& table, size_t c_lo,
                             size_t c_hi, const TemplateSet& ts, const sd_params* p, std::string& err,
                             const BatchSink& sink, const std::function<int(std::string&)>& while_busy = nullptr) {
    const bool timing = getenv("SD_TIMING") != nullptr;  // developer knob: stage times on stderr
    const double t_begin = now_s();
    std::string pkey;
    {
        sd_params pe = *p;
        apply_env_overrides(pe);
        pkey = pipe_cache_key(pe, 'C', ts.mseq, ts.mlen);
    }
    std::unique_ptr<Pipeline> pipe_h = getenv("SD_PIPE_CACHE_OFF") ? nullptr : pipe_cache_take(pkey);
    int rc = SD_OK;
    const bool from_cache = pipe_h != nullptr;
    if (pipe_h) {
        pipe_h->begin_job(p, ts.mseq.data(), ts.mlen.data(), (int32_t)ts.mseq.size());
    } else {
        pipe_h.reset(new Pipeline);
        rc = pipe_h->create(p, ts.mseq.data(), ts.mlen.data(), (int32_t)ts.mseq.size());
    }
    Pipeline& pipe = *pipe_h;
    if (rc) { err = pipe.eb; return rc; }
    pipe.restart_idle = true;
    std::vector<std::pair<size_t, size_t>> batches;
    {
        int64_t budget = pipe.row_budget();
        if (!from_cache) {
            int64_t rows = 0;
            for (size_t c = c_lo; c < c_hi; ++c) rows += table[c].len;
            if (rows > budget) budget = fresh_row_budget(budget, rows);   // (a job of one batch stays one batch)
        }
        plan_batches(table, c_lo, c_hi, budget, 1, batches);
    }
    std::vector<const char*> cptr;
    std::vector<int32_t> clen;
    for (size_t b = 0; b < batches.size() && rc == SD_OK; ++b) {
        const size_t c0 = batches[b].first, c1 = batches[b].second;
        cptr.clear();
        clen.clear();
        for (size_t c = c0; c < c1; ++c) {
            cptr.push_back(reads[(size_t)table[c].read].seq + table[c].off);
            clen.push_back(table[c].len);
        }
        rc = pipe.push(cptr, clen, [&sink, c0](const sd_rec* r, const int64_t* ro, size_t first, size_t n) { sink(c0 + first, c0 + first + n, r, ro); });
    }
    std::string busy_err;
    const double t_busy0 = now_s();
    const int busy_rc = (rc == SD_OK && while_busy) ? while_busy(busy_err) : SD_OK;
    const double t_busy = now_s() - t_busy0;
    const int rc2 = pipe.drain();
    if (rc == SD_OK) rc = rc2;
    if (rc) err = pipe.eb;
    // an undefined symbol in the share is the reference's own diagnostic (main.cpp:335): it is what the caller sees, also
    // when the device run over those bytes failed as well
    if (busy_rc) { rc = busy_rc; err = busy_err; }
    if (timing)
        std::fprintf(stderr, "[sd timing] host work under the device: %.2f ms\n", t_busy * 1e3);
    if (timing)
        std::fprintf(stderr, "[sd timing] %zu batches: pack+enqueue %.1f ms, wait %.1f ms, sink %.1f ms, kernels fill %.1f "
                     "trace %.1f compact %.2f ms, total %.1f ms\n", batches.size(), pipe.pack_s * 1e3, pipe.wait_s * 1e3,
                     pipe.sink_s * 1e3, pipe.fill_ms, pipe.trace_ms, pipe.compact_ms, (now_s() - t_begin) * 1e3);
    if (rc == SD_OK && !getenv("SD_PIPE_CACHE_OFF")) pipe_cache_give(pkey, std::move(pipe_h));
    return rc;
}

// Per-read assembly (main.cpp:104-117) of per-chunk records arriving in chunk order: chunk offsets,
// seam merge, raw TSV text.  Reads complete in input order.
namespace {
struct ReadAssembler {
    const std::vector<ReadView>& reads;
    const std::vector<CRef>& table;
    const std::vector<int32_t>& nch;
    const std::vector<std::string>& tnames;
    int threads;
    std::string& tsv;
    std::vector<sd_rec> cur;      // records of the read being assembled
    size_t next_read = 0;         // first read not yet written
    int32_t chunks_seen = 0;
    sd::RecordsWriter* rec_out = nullptr;   // set: completed reads go to the binary record stream and no text is made
    std::vector<std::string>* part_sink = nullptr;   // set: the text stays in pieces (in order) instead of being appended to tsv
    double t_merge = 0, t_text = 0;
    // A read of many chunks (a chromosome: 40 000) is merged and formatted AS ITS CHUNKS ARRIVE, so that the text of a
    // device batch is made (and, by sd_decompose_files, written) while the next batch is on the device instead of all at
    // the end.  The seam merge (main.cpp:287-302) is a scan whose state is one index: at decision index i it looks at the
    // six records behind i, keeps b[i], and either goes on at i + 1 or -- b[i] overlaps b[j] by more than half of b[j] --
    // also keeps b[j + 1] unchecked and goes on at j + 2.  A decision needs the records up to i + 7, so with more chunks
    // to come the scan stops eight records before the end of what has arrived; `cur` then holds that undecided tail.
    static constexpr int32_t kStreamChunks = 256;   // reads of more chunks than this take the streaming form
    std::vector<sd_rec> s_rows;   // kept rows not yet formatted
    int s_prev_end = 0;           // end of the last kept row (SaveBatch's prev_end)
    void stream_advance(bool final, std::vector<std::string>& parts) {
        const double t_m0 = now_s();
        const size_t N = cur.size();
        size_t i = 0;
        while (i < N && (final || i + 8 <= N)) {
            const size_t lim = i + 7 < N ? i + 7 : N;
            for (size_t j = i + 1; j < lim; ++j)
                if ((cur[i].end - cur[j].start) * 2 > (cur[j].end - cur[j].start)) {
                    s_rows.push_back(cur[i]);
                    i = j + 1;
                    break;
                }
            if (i < N) s_rows.push_back(cur[i]);
            ++i;
        }
        cur.erase(cur.begin(), cur.begin() + (long)std::min(i, N));   // (what stays is at most eight records)
        t_merge += now_s() - t_m0;
        if (s_rows.empty()) return;
        const double t_t0 = now_s();
        const ReadView& rd = reads[next_read];
        const size_t step = 32768, n_sl = (s_rows.size() + step - 1) / step, at = parts.size();
        parts.resize(at + n_sl);
        sd::parallel_for((int64_t)n_sl, threads, 1, [&](int64_t x) {
            const size_t r0 = (size_t)x * step, r1 = std::min(s_rows.size(), r0 + step);
            sd::format_rows(parts[at + (size_t)x], rd.name, rd.name_len, tnames, s_rows.data() + r0, r1 - r0,
                            r0 ? s_rows[r0 - 1].end : s_prev_end);
        });
        s_prev_end = s_rows.back().end;
        s_rows.clear();
        t_text += now_s() - t_t0;
    }
    ReadAssembler(const std::vector<ReadView>& r, const std::vector<CRef>& t, const std::vector<int32_t>& n,
                  const std::vector<std::string>& tn, int th, std::string& out)
        : reads(r), table(t), nch(n), tnames(tn), threads(th), tsv(out) {}
    void add(size_t c0, size_t c1, const sd_rec* recs, const int64_t* roff) {
        std::vector<std::vector<sd_rec>> done_rows;   // reads completed by this call, not yet merged / formatted
        std::vector<size_t> done_ids;
        std::vector<std::string> out_parts;           // the text of this call, in read order
        // merge + text (or record stream) of the completed reads gathered so far
        auto flush_done = [&]() {
            if (done_ids.empty()) return;
            const double t_m0 = now_s();
            sd::parallel_for((int64_t)done_ids.size(), threads, 4,
                             [&](int64_t q) { sd::seam_merge(done_rows[(size_t)q]); });
            t_merge += now_s() - t_m0;
            if (rec_out) {
                for (size_t q = 0; q < done_ids.size(); ++q) {
                    const ReadView& rd = reads[done_ids[q]];
                    rec_out->add_read(rd.name, rd.name_len, rd.len, done_rows[q].data(), (int64_t)done_rows[q].size());
                }
            } else {
                // text in slices of 32 k rows, so that a long read is formatted by all host threads as well; a slice only
                // needs the end of the row before it (SaveBatch's prev_end)
                struct Slice { size_t q, r0, r1; };
                std::vector<Slice> slices;
                const size_t step = 32768;
                for (size_t q = 0; q < done_ids.size(); ++q)
                    for (size_t r0 = 0; r0 < done_rows[q].size(); r0 += step)
                        slices.push_back(Slice{q, r0, std::min(done_rows[q].size(), r0 + step)});
                const size_t at = out_parts.size();
                out_parts.resize(at + slices.size());
                const double t_t0 = now_s();
                sd::parallel_for((int64_t)slices.size(), threads, 1, [&](int64_t x) {
                    const Slice& sl = slices[(size_t)x];
                    const ReadView& rd = reads[done_ids[sl.q]];
                    const std::vector<sd_rec>& rows = done_rows[sl.q];
                    sd::format_rows(out_parts[at + (size_t)x], rd.name, rd.name_len, tnames, rows.data() + sl.r0, sl.r1 - sl.r0,
                                    sl.r0 ? rows[sl.r0 - 1].end : 0);
                });
                t_text += now_s() - t_t0;
            }
            done_rows.clear();
            done_ids.clear();
        };
        for (size_t c = c0; c < c1;) {
            // the chunks of this call that belong to the read being assembled: their records are one contiguous range,
            // moved (chunk offsets added, main.cpp:109-111) by all threads when there are many -- a 200-Mb sequence is
            // 40 000 chunks of one read
            const size_t ce = std::min(c1, c + (size_t)(nch[next_read] - chunks_seen));
            const int64_t x0 = roff[c - c0], x1 = roff[ce - c0];
            const size_t base = cur.size();
            cur.resize(base + (size_t)(x1 - x0));
            auto move_chunk = [&](int64_t k) {
                const size_t ck = c + (size_t)k;
                const int32_t add = (int32_t)table[ck].off;
                for (int64_t x = roff[ck - c0]; x < roff[ck - c0 + 1]; ++x) {
                    sd_rec t = recs[x];
                    t.start += add;
                    t.end += add;
                    cur[base + (size_t)(x - x0)] = t;
                }
            };
            if (ce - c >= 512) sd::parallel_for((int64_t)(ce - c), threads, 64, move_chunk);
            else for (size_t k = 0; k < ce - c; ++k) move_chunk((int64_t)k);
            chunks_seen += (int32_t)(ce - c);
            c = ce;
            if (nch[next_read] > kStreamChunks && !rec_out) {   // a huge read: merged and formatted as it arrives
                flush_done();                                   // (the reads before it come first in the text)
                const bool fin = chunks_seen == nch[next_read];
                stream_advance(fin, out_parts);
                if (fin) { ++next_read; chunks_seen = 0; s_prev_end = 0; cur.clear(); }
                continue;
            }
            if (chunks_seen == nch[next_read]) {
                done_rows.emplace_back();
                done_rows.back().swap(cur);
                done_ids.push_back(next_read);
                ++next_read;
                chunks_seen = 0;
            }
        }
        flush_done();
        if (part_sink) {   // the caller gathers (or writes) the pieces itself, in parallel
            for (std::string& part : out_parts) part_sink->push_back(std::move(part));
            return;
        }
        size_t total = tsv.size();
        for (const std::string& part : out_parts) total += part.size();
        tsv.reserve(std::max(total, tsv.capacity()));
        for (const std::string& part : out_parts) tsv += part;
    }
};
}  // namespace

static int decompose_impl(const std::vector<ReadView>& reads, const std::vector<sd::Seq>& monos,
                          const sd_params* p, std::string& tsv, std::string& err, const char* records_out = nullptr,
                          std::vector<std::string>* parts_out = nullptr,   // parts_out: the text in pieces instead of `tsv`
                          const std::function<bool(std::vector<std::string>&)>& flush_parts = nullptr) {   // ... handed over after every batch
    if (monos.empty()) { err = "no monomers"; return SD_ERR_PARAM; }
    for (const ReadView& r : reads)
        if (r.len <= 0) { err = "ERROR: Sequence " + std::string(r.name, r.name_len) + " is empty"; return SD_ERR_EMPTY; }
    TemplateSet ts(monos);
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    build_chunk_table(reads, p, table, nch);
    ReadAssembler as(reads, table, nch, ts.tnames, p->threads, tsv);
    as.part_sink = parts_out;
    sd::RecordsWriter rw;
    if (records_out) {
        const int orc = rw.open(records_out, *p, ts.tnames, err);
        if (orc) return orc;
        as.rec_out = &rw;
    }
    bool flush_ok = true;
    int rc = run_chunk_batches(reads, table, 0, table.size(), ts, p, err,
                               [&](size_t c0, size_t c1, const sd_rec* recs, const int64_t* roff) {
                                   as.add(c0, c1, recs, roff);
                                   if (flush_parts && parts_out && flush_ok) flush_ok = flush_parts(*parts_out);
                               });
    if (rc == SD_OK && !flush_ok) { rc = SD_ERR_IO; err = "short write"; }
    if (records_out && rc == SD_OK) rc = rw.close(err, records_out);
    return rc;
}

int sd_decompose(const char* const* read_names, const char* const* read_seqs,
                 const int64_t* read_lens, int32_t n_reads, const char* const* mono_names,
                 const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono,
                 const sd_params* p, char** tsv, size_t* tsv_len, char* errbuf, size_t errlen) {
    if (!tsv || !tsv_len) return SD_ERR_PARAM;
    *tsv = nullptr;
    *tsv_len = 0;
    std::string err;
    {
        const int vrc = validate_params(p, err);  // before anything walks the chunk table (part_size > 0)
        if (vrc) { set_err(errbuf, errlen, err); return vrc; }
        if (n_reads < 0 || n_mono < 0 || (n_reads && (!read_names || !read_seqs || !read_lens)) ||
            (n_mono && (!mono_names || !mono_seqs || !mono_lens))) {
            set_err(errbuf, errlen, "null input array");
            return SD_ERR_PARAM;
        }
    }
    std::vector<ReadView> reads((size_t)std::max(n_reads, 0));
    std::vector<sd::Seq> monos((size_t)std::max(n_mono, 0));
    {
        // alphabet check of all reads on the host threads; the first offending read (in input
        // order) is reported, as load_fasta does (main.cpp:329-341)
        std::vector<int> bad((size_t)std::max(n_reads, 0), 0);
        sd::parallel_for(n_reads, p->threads, 8, [&](int64_t r) {
            reads[(size_t)r] = ReadView{read_names[r], std::strlen(read_names[r]), read_seqs[r], read_lens[r]};
            std::string e2;
            bad[(size_t)r] = sd::check_alphabet(read_names[r], read_seqs[r], read_lens[r], e2) != SD_OK;
        });
        for (int32_t r = 0; r < n_reads; ++r)
            if (bad[(size_t)r]) {
                int rc = sd::check_alphabet(read_names[r], read_seqs[r], read_lens[r], err);
                set_err(errbuf, errlen, err);
                return rc;
            }
    }
    for (int32_t m = 0; m < n_mono; ++m) {
        monos[(size_t)m].name = mono_names[m];
        monos[(size_t)m].seq.assign(mono_seqs[m], (size_t)mono_lens[m]);
        int rc = sd::check_alphabet(mono_names[m], mono_seqs[m], mono_lens[m], err);
        if (rc) { set_err(errbuf, errlen, err); return rc; }
    }
    std::string out;
    int rc = decompose_impl(reads, monos, p, out, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    char* o = static_cast<char*>(std::malloc(out.size() + 1));
    std::memcpy(o, out.data(), out.size());
    o[out.size()] = 0;
    *tsv = o;
    *tsv_len = out.size();
    return SD_OK;
}

// raw_tsv_out: the text `dp` prints; records_out: the same rows as the binary record stream (sd_records.hpp), no text made
static int decompose_files_impl(const char* reads_fa, const char* monomers_fa, const sd_params* p,
                                const char* raw_tsv_out, const char* records_out, char* errbuf, size_t errlen) {
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (!reads_fa || !monomers_fa || (!raw_tsv_out && !records_out)) return SD_ERR_PARAM;
    sd::FastaFile rf, mf;
    rc = rf.open(reads_fa, p->threads, err);                                  // main.cpp:394
    if (rc == SD_OK) rc = rf.validate(0, rf.recs.size(), p->threads, err);
    if (rc == SD_OK) rc = mf.open(monomers_fa, p->threads, err);              // main.cpp:395
    if (rc == SD_OK) rc = mf.validate(0, mf.recs.size(), p->threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::vector<sd::Seq> monos;
    for (const auto& r : mf.recs) monos.push_back(sd::Seq{std::string(r.name, r.name_len), std::string(r.seq, (size_t)r.len)});
    std::string out;
    std::vector<ReadView> views;
    views.reserve(rf.recs.size());
    for (const auto& r : rf.recs) views.push_back(ReadView{r.name, r.name_len, r.seq, r.len});
    // The text stays in the pieces the threads formatted and goes to the file after every device batch (write_parts: no
    // gather), i.e. while the next batch is on the device -- for a chromosome-sized read too, whose rows are merged and
    // formatted as its chunks arrive (ReadAssembler::stream_advance).
    std::vector<std::string> parts;
    int fd = -1;
    if (raw_tsv_out) {
        fd = ::open(raw_tsv_out, O_RDWR | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) { set_err(errbuf, errlen, std::string("cannot write ") + raw_tsv_out); return SD_ERR_IO; }
    }
    int64_t off = 0;
    auto flush = [&](std::vector<std::string>& ps) {
        const bool ok = fd < 0 || sd::write_parts(fd, off, ps, p->threads);
        ps.clear();
        return ok;
    };
    rc = decompose_impl(views, monos, p, out, err, records_out, &parts, flush);
    const bool closed = fd < 0 || ::close(fd) == 0;
    if (rc) { set_err(errbuf, errlen, err == "short write" ? std::string("short write to ") + raw_tsv_out : err); return rc; }
    if (!closed) { set_err(errbuf, errlen, std::string("short write to ") + raw_tsv_out); return SD_ERR_IO; }
    return SD_OK;
}

int sd_decompose_files(const char* reads_fa, const char* monomers_fa, const sd_params* p,
                       const char* raw_tsv_out, char* errbuf, size_t errlen) {
    if (!raw_tsv_out) return SD_ERR_PARAM;
    return decompose_files_impl(reads_fa, monomers_fa, p, raw_tsv_out, nullptr, errbuf, errlen);
}

int sd_decompose_files_records(const char* reads_fa, const char* monomers_fa, const sd_params* p,
                               const char* records_out, char* errbuf, size_t errlen) {
    if (!records_out) return SD_ERR_PARAM;
    return decompose_files_impl(reads_fa, monomers_fa, p, nullptr, records_out, errbuf, errlen);
}

// ---- the binary record stream as a format of its own (host only) ---------------------------------------------
int sd_write_records(const char* path, const sd_params* p, const char* const* tmpl_names, int32_t n_templates,
                     const char* const* read_names, const int64_t* read_lens, int32_t n_reads, const sd_rec* rows,
                     const int64_t* row_off, char* errbuf, size_t errlen) {
    if (!path || !p || n_templates < 0 || n_reads < 0 || (n_templates && !tmpl_names) || (n_reads && (!read_names || !row_off)))
        return SD_ERR_PARAM;
    std::vector<std::string> tn;
    for (int32_t t = 0; t < n_templates; ++t) tn.emplace_back(tmpl_names[t]);
    for (int32_t r = 0; r < n_reads; ++r) {
        if (row_off[r + 1] < row_off[r] || (row_off[r + 1] > row_off[r] && !rows)) { set_err(errbuf, errlen, "row offsets must not decrease"); return SD_ERR_PARAM; }
        for (int64_t x = row_off[r]; x < row_off[r + 1]; ++x)
            if (rows[x].tmpl < 0 || rows[x].tmpl >= n_templates) { set_err(errbuf, errlen, "record with a template index outside the template table"); return SD_ERR_PARAM; }
    }
    std::string err;
    sd::RecordsWriter rw;
    int rc = rw.open(path, *p, tn, err);
    if (rc == SD_OK) {
        for (int32_t r = 0; r < n_reads; ++r)
            rw.add_read(read_names[r], std::strlen(read_names[r]), read_lens ? read_lens[r] : -1, rows + row_off[r], row_off[r + 1] - row_off[r]);
        rc = rw.close(err, path);
    }
    if (rc) set_err(errbuf, errlen, err);
    return rc;
}

void sd_records_free(sd_records* r) {
    if (!r) return;
    if (r->tmpl_names) for (int32_t t = 0; t < r->n_templates; ++t) std::free(r->tmpl_names[t]);
    if (r->read_names) for (int32_t i = 0; i < r->n_reads; ++i) std::free(r->read_names[i]);
    std::free(r->tmpl_names); std::free(r->read_names); std::free(r->read_lens); std::free(r->row_off); std::free(r->rows);
    std::memset(r, 0, sizeof *r);
}

int sd_read_records(const char* path, sd_records* out, char* errbuf, size_t errlen) {
    if (!path || !out) return SD_ERR_PARAM;
    std::memset(out, 0, sizeof *out);
    sd::RecordsFile f;
    std::string err;
    const int rc = f.load(path, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    auto dup = [](const std::string& x) { char* c = static_cast<char*>(std::malloc(x.size() + 1)); if (c) { std::memcpy(c, x.data(), x.size()); c[x.size()] = 0; } return c; };
    out->ins = f.score[0]; out->del = f.score[1]; out->mismatch = f.score[2]; out->match = f.score[3];
    out->part_size = f.part_size; out->overlap = f.overlap; out->ed_thr = f.ed_thr;
    out->n_templates = (int32_t)f.tnames.size();
    out->n_reads = (int32_t)f.rnames.size();
    out->n_rows = (int64_t)f.rows.size();
    out->tmpl_names = static_cast<char**>(std::calloc(std::max<size_t>(f.tnames.size(), 1), sizeof(char*)));
    out->read_names = static_cast<char**>(std::calloc(std::max<size_t>(f.rnames.size(), 1), sizeof(char*)));
    out->read_lens = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * std::max<size_t>(f.rnames.size(), 1)));
    out->row_off = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (f.rnames.size() + 1)));
    out->rows = static_cast<sd_rec*>(std::malloc(sizeof(sd_rec) * std::max<size_t>(f.rows.size(), 1)));
    bool ok = out->tmpl_names && out->read_names && out->read_lens && out->row_off && out->rows;
    for (size_t t = 0; ok && t < f.tnames.size(); ++t) ok = (out->tmpl_names[t] = dup(f.tnames[t])) != nullptr;
    for (size_t r = 0; ok && r < f.rnames.size(); ++r) ok = (out->read_names[r] = dup(f.rnames[r])) != nullptr;
    if (!ok) { sd_records_free(out); set_err(errbuf, errlen, "out of host memory"); return SD_ERR_INTERNAL; }
    if (!f.rnames.empty()) std::memcpy(out->read_lens, f.read_lens.data(), sizeof(int64_t) * f.rnames.size());
    std::memcpy(out->row_off, f.row_off.data(), sizeof(int64_t) * f.row_off.size());
    if (!f.rows.empty()) std::memcpy(out->rows, f.rows.data(), sizeof(sd_rec) * f.rows.size());
    return SD_OK;
}

// record stream -> the raw TSV SaveBatch prints for the same rows (main.cpp:272-285); slices of 32 k rows on all threads
int sd_records_to_raw_tsv(const char* records_path, const char* raw_tsv_out, int32_t threads, char* errbuf, size_t errlen) {
    if (!records_path || !raw_tsv_out) return SD_ERR_PARAM;
    sd::RecordsFile f;
    std::string err;
    int rc = f.load(records_path, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    struct Slice { size_t r; int64_t a, b; };
    std::vector<Slice> slices;
    for (size_t r = 0; r < f.rnames.size(); ++r)
        for (int64_t a = f.row_off[r]; a < f.row_off[r + 1]; a += 32768)
            slices.push_back(Slice{r, a, std::min<int64_t>(f.row_off[r + 1], a + 32768)});
    std::vector<std::string> parts(slices.size());
    sd::parallel_for((int64_t)slices.size(), std::max(1, (int)threads), 1, [&](int64_t x) {
        const Slice& sl = slices[(size_t)x];
        sd::format_rows(parts[(size_t)x], f.rnames[sl.r].data(), f.rnames[sl.r].size(), f.tnames, f.rows.data() + sl.a,
                        (size_t)(sl.b - sl.a), sl.a > f.row_off[sl.r] ? f.rows[(size_t)sl.a - 1].end : 0);
    });
    const int fd = ::open(raw_tsv_out, O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { set_err(errbuf, errlen, std::string("cannot write ") + raw_tsv_out); return SD_ERR_IO; }
    int64_t off = 0;
    const bool ok = sd::write_parts(fd, off, parts, std::max(1, (int)threads));
    if (::close(fd) != 0 || !ok) { set_err(errbuf, errlen, std::string("short write to ") + raw_tsv_out); return SD_ERR_IO; }
    return SD_OK;
}

// -------------------------------------------------------------------------------------------
// chunk-range form: multi-GPU sharding of one job, one process per GPU
// -------------------------------------------------------------------------------------------
static void text_pool_clear();
void sd_release_cache(void) { pipe_cache_clear(); g_pool.release_all(); g_pinpool.release_all(); g_deferred.drain(); text_pool_clear(); }

int64_t sd_chunk_table_size(const int64_t* read_lens, int32_t n_reads, int32_t part_size, int32_t overlap) {
    if (!read_lens || n_reads < 0 || part_size <= 0 || overlap < 0) return -1;
    int64_t n = 0;
    for (int32_t r = 0; r < n_reads; ++r) n += sd::chunk_plan(read_lens[r], part_size, overlap, [](int64_t, int32_t) {});
    return n;
}

namespace {
// Records of a chunk range as the C-ABI hands them out (malloc'ed records + offsets), filled batch by batch straight
// from the pipeline's pinned buffers: one copy per record (a std::vector + a final copy cost 6 ms per 200-Mb job).
struct RangeCollector {
    sd_rec* recs = nullptr;
    int64_t* off = nullptr;
    size_t cap = 0, n = 0, c_lo, n_chunks;
    bool failed = false;
    RangeCollector(size_t chunks, size_t first) : c_lo(first), n_chunks(chunks) {
        off = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (chunks + 1)));
        cap = std::max<size_t>(4096, chunks * 40);   // ~32 records per 5.5-kb chunk of satellite DNA
        recs = static_cast<sd_rec*>(std::malloc(sizeof(sd_rec) * cap));
        if (!off || !recs) failed = true; else off[0] = 0;
    }
    void add(size_t c0, size_t c1, const sd_rec* r, const int64_t* ro, int threads) {
        if (failed) return;
        const size_t k = (size_t)ro[c1 - c0];
        if (n + k > cap) {
            cap = std::max(n + k, cap + cap / 2);
            sd_rec* q = static_cast<sd_rec*>(std::realloc(recs, sizeof(sd_rec) * cap));
            if (!q) { failed = true; return; }
            recs = q;
        }
        const size_t pieces = (k + 65535) / 65536;
        sd::parallel_for((int64_t)pieces, threads, 1, [&](int64_t x) {
            const size_t a = (size_t)x * 65536, b = std::min(k, a + 65536);
            std::memcpy(recs + n + a, r + a, sizeof(sd_rec) * (b - a));
        });
        for (size_t c = 1; c <= c1 - c0; ++c) off[c0 - c_lo + c] = (int64_t)n + ro[c];
        n += k;
    }
    void release(sd_rec** r, int64_t** o) { *r = recs; *o = off; recs = nullptr; off = nullptr; }
    ~RangeCollector() { std::free(recs); std::free(off); }
};
}  // namespace

int sd_decompose_chunk_range(const char* const* read_seqs, const int64_t* read_lens, int32_t n_reads,
                             const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono,
                             const sd_params* p, int64_t chunk_lo, int64_t chunk_hi, sd_rec** recs,
                             int64_t** rec_off, char* errbuf, size_t errlen) {
    if (!recs || !rec_off || !read_seqs || !read_lens || !mono_seqs || !mono_lens) return SD_ERR_PARAM;
    *recs = nullptr;
    *rec_off = nullptr;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (n_mono <= 0) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    std::vector<ReadView> reads((size_t)std::max(n_reads, 0));
    for (int32_t r = 0; r < n_reads; ++r) {
        if (read_lens[r] <= 0) { set_err(errbuf, errlen, "ERROR: Sequence #" + std::to_string(r) + " is empty"); return SD_ERR_EMPTY; }
        reads[(size_t)r] = ReadView{"", 0, read_seqs[r], read_lens[r]};
    }
    std::vector<sd::Seq> monos((size_t)n_mono);
    for (int32_t m = 0; m < n_mono; ++m) {
        monos[(size_t)m].name = "m" + std::to_string(m);
        monos[(size_t)m].seq.assign(mono_seqs[m], (size_t)mono_lens[m]);
        rc = sd::check_alphabet(monos[(size_t)m].name.c_str(), mono_seqs[m], mono_lens[m], err);
        if (rc) { set_err(errbuf, errlen, err); return rc; }
    }
    TemplateSet ts(monos);
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    build_chunk_table(reads, p, table, nch);
    if (chunk_lo < 0 || chunk_hi < chunk_lo || (size_t)chunk_hi > table.size()) {
        set_err(errbuf, errlen, "chunk range outside the chunk table");
        return SD_ERR_PARAM;
    }
    // only the bases this range touches are validated (every rank validates its own share) -- under the device's work
    auto validate = [&](std::string& verr) -> int {
        std::vector<int> bad((size_t)(chunk_hi - chunk_lo), 0);
        sd::parallel_for(chunk_hi - chunk_lo, p->threads, 64, [&](int64_t i) {
            const CRef& c = table[(size_t)(chunk_lo + i)];
            std::string e2;
            bad[(size_t)i] = sd::check_alphabet("", reads[(size_t)c.read].seq + c.off, c.len, e2) != SD_OK;
        });
        for (size_t i = 0; i < bad.size(); ++i)
            if (bad[i]) {
                const CRef& c = table[(size_t)chunk_lo + i];
                const std::string nm = "#" + std::to_string(c.read);
                return sd::check_alphabet(nm.c_str(), reads[(size_t)c.read].seq + c.off, c.len, verr);
            }
        return SD_OK;
    };
    RangeCollector col((size_t)(chunk_hi - chunk_lo), (size_t)chunk_lo);
    rc = run_chunk_batches(reads, table, (size_t)chunk_lo, (size_t)chunk_hi, ts, p, err,
                           [&](size_t c0, size_t c1, const sd_rec* r, const int64_t* ro) { col.add(c0, c1, r, ro, p->threads); },
                           validate);
    if (rc == SD_OK && col.failed) { rc = SD_ERR_INTERNAL; err = "out of host memory"; }
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    col.release(recs, rec_off);
    return SD_OK;
}

// File form of the chunk-range call for a job sharded over ranks: every rank maps and indexes the FASTA
// (no copy of the sequences), takes the contiguous share block_range(n_chunks, rank, world) of the global
// chunk table and checks the alphabet of the reads that share touches only (main.cpp:329-341 reports the
// first offending read in file order: the caller raises the error of the lowest failing rank).
struct sd_range_asm;
static int range_asm_from_files(sd::FastaFile& rf, sd::FastaFile& mf, const sd_params* p, int64_t lo, int64_t hi,
                                sd_rec* recs, int64_t* off, sd_seam_edge* edge, sd_range_asm** hout, std::string& err);

// edge / hout set: the share's records stay with a range assembler (sd_decompose_files_range_begin)
static int decompose_files_range_impl(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                                      int32_t world, sd_rec** recs, int64_t** rec_off, int64_t* chunk_lo, int64_t* chunk_hi,
                                      int64_t* n_chunks_total, sd_seam_edge* edge, sd_range_asm** hout, char* errbuf, size_t errlen) {
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    sd::FastaFile rf, mf;
    rc = rf.open(reads_fa, p->threads, err);
    if (rc == SD_OK) rc = mf.open(monomers_fa, p->threads, err);
    if (rc == SD_OK) rc = mf.validate(0, mf.recs.size(), p->threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::vector<sd::Seq> monos;
    for (const auto& r : mf.recs) monos.push_back(sd::Seq{std::string(r.name, r.name_len), std::string(r.seq, (size_t)r.len)});
    if (monos.empty()) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    std::vector<ReadView> reads;
    reads.reserve(rf.recs.size());
    for (const auto& r : rf.recs) {
        if (r.len <= 0) { set_err(errbuf, errlen, "ERROR: Sequence " + std::string(r.name, r.name_len) + " is empty"); return SD_ERR_EMPTY; }
        reads.push_back(ReadView{r.name, r.name_len, r.seq, r.len});
    }
    TemplateSet ts(monos);
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    build_chunk_table(reads, p, table, nch);
    const int64_t n = (int64_t)table.size();
    const int64_t base = n / world, extra = n % world;
    const int64_t lo = rank * base + std::min<int64_t>(rank, extra);
    const int64_t hi = lo + base + (rank < extra ? 1 : 0);
    if (chunk_lo) *chunk_lo = lo;
    if (chunk_hi) *chunk_hi = hi;
    if (n_chunks_total) *n_chunks_total = n;
    // the reads this share touches are checked under the device's work (run_chunk_batches: while_busy)
    auto validate = [&](std::string& verr) -> int {
        return hi > lo ? rf.validate((size_t)table[(size_t)lo].read, (size_t)table[(size_t)hi - 1].read + 1, p->threads, verr) : SD_OK;
    };
    RangeCollector col((size_t)(hi - lo), (size_t)lo);
    rc = run_chunk_batches(reads, table, (size_t)lo, (size_t)hi, ts, p, err,
                           [&](size_t c0, size_t c1, const sd_rec* r, const int64_t* ro) { col.add(c0, c1, r, ro, p->threads); },
                           validate);
    if (rc == SD_OK && col.failed) { rc = SD_ERR_INTERNAL; err = "out of host memory"; }
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (hout) {
        sd_rec* r = nullptr;
        int64_t* o = nullptr;
        col.release(&r, &o);
        rc = range_asm_from_files(rf, mf, p, lo, hi, r, o, edge, hout, err);   // (takes r / o over, also when it fails)
        if (rc) { set_err(errbuf, errlen, err); return rc; }
        return SD_OK;
    }
    col.release(recs, rec_off);
    return SD_OK;
}

int sd_decompose_files_range(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                             int32_t world, sd_rec** recs, int64_t** rec_off, int64_t* chunk_lo, int64_t* chunk_hi,
                             int64_t* n_chunks_total, char* errbuf, size_t errlen) {
    if (!recs || !rec_off || !reads_fa || !monomers_fa || world < 1 || rank < 0 || rank >= world) return SD_ERR_PARAM;
    *recs = nullptr;
    *rec_off = nullptr;
    return decompose_files_range_impl(reads_fa, monomers_fa, p, rank, world, recs, rec_off, chunk_lo, chunk_hi, n_chunks_total,
                                      nullptr, nullptr, errbuf, errlen);
}

int sd_decompose_files_range_begin(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                                   int32_t world, sd_seam_edge* edge, sd_range_asm** h, int64_t* chunk_lo, int64_t* chunk_hi,
                                   int64_t* n_chunks_total, char* errbuf, size_t errlen) {
    if (!edge || !h || !reads_fa || !monomers_fa || world < 1 || rank < 0 || rank >= world) return SD_ERR_PARAM;
    *h = nullptr;
    return decompose_files_range_impl(reads_fa, monomers_fa, p, rank, world, nullptr, nullptr, chunk_lo, chunk_hi, n_chunks_total,
                                      edge, h, errbuf, errlen);
}

// Rank 0 of a sharded job: the records of all chunks in table order -> raw TSV file (names and lengths come
// from the FASTA index; host only).
int sd_assemble_files_tsv(const char* reads_fa, const char* monomers_fa, const sd_params* p, const sd_rec* recs,
                          const int64_t* rec_off, int64_t n_chunks, const char* raw_tsv_out, char* errbuf, size_t errlen) {
    if (!reads_fa || !monomers_fa || !rec_off || !raw_tsv_out || (!recs && rec_off[n_chunks] > 0)) return SD_ERR_PARAM;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    sd::FastaFile rf, mf;
    rc = rf.open(reads_fa, p->threads, err);
    if (rc == SD_OK) rc = mf.open(monomers_fa, p->threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::vector<std::string> rnames, mnames;
    std::vector<const char*> rn, mn;
    std::vector<int64_t> rl;
    for (const auto& r : rf.recs) { rnames.emplace_back(r.name, r.name_len); rl.push_back(r.len); }
    for (const auto& r : mf.recs) mnames.emplace_back(r.name, r.name_len);
    for (const std::string& x : rnames) rn.push_back(x.c_str());
    for (const std::string& x : mnames) mn.push_back(x.c_str());
    char* tsv = nullptr;
    size_t len = 0;
    rc = sd_assemble_tsv(rn.data(), rl.data(), (int32_t)rn.size(), mn.data(), (int32_t)mn.size(), p, recs, rec_off, n_chunks,
                         &tsv, &len, errbuf, errlen);
    if (rc) return rc;
    FILE* fp = std::fopen(raw_tsv_out, "wb");
    if (!fp) { std::free(tsv); set_err(errbuf, errlen, std::string("cannot write ") + raw_tsv_out); return SD_ERR_IO; }
    const size_t w = std::fwrite(tsv, 1, len, fp);
    std::free(tsv);
    if (std::fclose(fp) != 0 || w != len) { set_err(errbuf, errlen, std::string("short write to ") + raw_tsv_out); return SD_ERR_IO; }
    return SD_OK;
}

int sd_assemble_tsv(const char* const* read_names, const int64_t* read_lens, int32_t n_reads,
                    const char* const* mono_names, int32_t n_mono, const sd_params* p,
                    const sd_rec* recs, const int64_t* rec_off, int64_t n_chunks, char** tsv,
                    size_t* tsv_len, char* errbuf, size_t errlen) {
    if (!tsv || !tsv_len || !read_names || !read_lens || !mono_names || !rec_off || (!recs && rec_off[n_chunks] > 0))
        return SD_ERR_PARAM;
    *tsv = nullptr;
    *tsv_len = 0;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::vector<ReadView> reads((size_t)std::max(n_reads, 0));
    for (int32_t r = 0; r < n_reads; ++r)
        reads[(size_t)r] = ReadView{read_names[r], std::strlen(read_names[r]), nullptr, read_lens[r]};
    std::vector<std::string> tnames;
    for (int32_t m = 0; m < n_mono; ++m) tnames.emplace_back(mono_names[m]);
    for (int32_t m = 0; m < n_mono; ++m) tnames.push_back(std::string(mono_names[m]) + "'");
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    build_chunk_table(reads, p, table, nch);
    if ((int64_t)table.size() != n_chunks) {
        set_err(errbuf, errlen, "record offsets do not match the chunk table of these reads");
        return SD_ERR_PARAM;
    }
    for (int64_t x = 0; x < rec_off[n_chunks]; ++x)
        if (recs[x].tmpl < 0 || recs[x].tmpl >= 2 * n_mono) {
            set_err(errbuf, errlen, "record with a template index outside the monomer set");
            return SD_ERR_PARAM;
        }
    std::string out;
    std::vector<std::string> parts;
    const double t_a0 = now_s();
    ReadAssembler as(reads, table, nch, tnames, p->threads, out);
    as.part_sink = &parts;
    // in slices, so that the formatting threads always have a few hundred reads to share
    const size_t step = 4096;
    for (size_t c0 = 0; c0 < table.size(); c0 += step) {
        const size_t c1 = std::min(table.size(), c0 + step);
        std::vector<int64_t> ro(c1 - c0 + 1);
        for (size_t c = c0; c <= c1; ++c) ro[c - c0] = rec_off[c] - rec_off[c0];
        as.add(c0, c1, recs + rec_off[c0], ro.data());
    }
    const double t_g0 = now_s();
    // one copy, by all threads, straight into the buffer the caller gets (a 200-Mb sequence is 52 MB of rows: appending
    // the pieces to a string and copying that once more was two thirds of this call)
    std::vector<size_t> at(parts.size() + 1, 0);
    for (size_t i = 0; i < parts.size(); ++i) at[i + 1] = at[i] + parts[i].size();
    const size_t total = at[parts.size()];
    char* o = static_cast<char*>(std::malloc(total + 1));
    if (!o) { set_err(errbuf, errlen, "out of host memory"); return SD_ERR_INTERNAL; }
    sd::parallel_for((int64_t)parts.size(), p->threads, 1, [&](int64_t i) {
        if (!parts[(size_t)i].empty()) std::memcpy(o + at[(size_t)i], parts[(size_t)i].data(), parts[(size_t)i].size());
    });
    o[total] = 0;
    *tsv = o;
    *tsv_len = total;
    if (getenv("SD_TIMING"))
        std::fprintf(stderr, "[sd timing] assemble: %.1f ms to the pieces (seam merge %.1f, text %.1f), gather %.1f ms\n",
                     (t_g0 - t_a0) * 1e3, as.t_merge * 1e3, as.t_text * 1e3, (now_s() - t_g0) * 1e3);
    return SD_OK;
}

// -------------------------------------------------------------------------------------------
// one read over several ranks: every rank assembles its own chunk range (sd_seam.hpp; protocol in sd_hip.h)
// -------------------------------------------------------------------------------------------
static_assert(sizeof(sd_seam_edge) == 160, "sd_seam_edge is exchanged between processes as bytes");

struct sd_range_asm {
    std::vector<std::string> rname_store, tnames;
    std::vector<ReadView> reads;
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    int threads = 1;
    // a crossing piece: records with chunk offsets applied + its scans + the text made ahead
    struct Piece {
        size_t read = 0;
        std::vector<sd_rec> rows;
        sd::SeamPiece sp;
        std::vector<std::string> body;   // text of sp.kept[sp.body_from ..)
        std::string head, tail;          // made by sd_range_assemble_text
    };
    std::unique_ptr<Piece> front, back;  // a share inside one read has only `front` (open at both ends)
    std::vector<std::string> middle;     // text of the reads that lie completely inside the share
    std::vector<const std::string*> order;   // the text, in order (after sd_range_assemble_text)
    bool text_done = false;
    sd_rec* own_recs = nullptr;          // sd_decompose_files_range_begin: the records of the share stay with the handle
    int64_t* own_off = nullptr;          // (sd_range_assemble_records lends them for the gather fall-back)
    int64_t own_chunks = 0;
    ~sd_range_asm() { std::free(own_recs); std::free(own_off); }
    double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    void rows_of(size_t c0, size_t c1, size_t lo, const sd_rec* recs, const int64_t* roff, std::vector<sd_rec>& out) const {
        const int64_t x0 = roff[c0 - lo], x1 = roff[c1 - lo];
        out.resize((size_t)(x1 - x0));
        sd::parallel_for((int64_t)(c1 - c0), threads, 64, [&](int64_t k) {
            const size_t c = c0 + (size_t)k;
            const int32_t add = (int32_t)table[c].off;
            for (int64_t x = roff[c - lo]; x < roff[c - lo + 1]; ++x) {
                sd_rec t = recs[x];
                t.start += add;
                t.end += add;
                out[(size_t)(x - x0)] = t;
            }
        });
    }
    // text of rows[idx[a..b)] in slices, prev_end = end of the row printed before idx[a]
    void format_idx(const Piece& pc, const std::vector<uint32_t>& idx, size_t a, size_t b, int32_t prev_end,
                    std::vector<std::string>& out) const {
        const size_t step = 16384, n_sl = (b - a + step - 1) / step;
        out.assign(n_sl, std::string());
        const ReadView& rd = reads[pc.read];
        sd::parallel_for((int64_t)n_sl, threads, 1, [&](int64_t x) {
            const size_t r0 = a + (size_t)x * step, r1 = std::min(b, r0 + step);
            std::vector<sd_rec> tmp(r1 - r0);
            for (size_t k = r0; k < r1; ++k) tmp[k - r0] = pc.rows[idx[k]];
            sd::format_rows(out[(size_t)x], rd.name, rd.name_len, tnames, tmp.data(), tmp.size(),
                            r0 > a ? pc.rows[idx[r0 - 1]].end : prev_end);
        });
    }
};

static int range_asm_begin(std::unique_ptr<sd_range_asm>& hp, const sd_params* p, int64_t chunk_lo, int64_t chunk_hi,
                           const sd_rec* recs, const int64_t* roff, sd_seam_edge* edge, std::string& err) {
    sd_range_asm& h = *hp;
    const double t0 = now_s();
    h.threads = p->threads;
    build_chunk_table(h.reads, p, h.table, h.nch);
    std::memset(edge, 0, sizeof(*edge));
    if (chunk_lo < 0 || chunk_hi < chunk_lo || (size_t)chunk_hi > h.table.size()) { err = "chunk range outside the chunk table"; return SD_ERR_PARAM; }
    if (chunk_hi == chunk_lo) return SD_OK;   // an empty share: ok stays 0
    const size_t lo = (size_t)chunk_lo, hi = (size_t)chunk_hi;
    const int64_t n_rec = roff[hi - lo];
    const int32_t n_tmpl = (int32_t)h.tnames.size();
    for (int64_t x = 0; x < n_rec; ++x)
        if (recs[x].tmpl < 0 || recs[x].tmpl >= n_tmpl) { err = "record with a template index outside the monomer set"; return SD_ERR_PARAM; }
    std::vector<size_t> cstart(h.reads.size() + 1, 0);
    for (size_t r = 0; r < h.reads.size(); ++r) cstart[r + 1] = cstart[r] + (size_t)h.nch[r];
    const size_t ra = (size_t)h.table[lo].read, rb = (size_t)h.table[hi - 1].read;
    const bool open_front = lo > cstart[ra], open_back = hi < cstart[rb + 1];
    size_t mid_lo = lo, mid_hi = hi;   // chunks of the reads that lie completely inside
    bool ok = true;
    auto make_piece = [&](size_t read, size_t c0, size_t c1, bool of, bool ob) {
        std::unique_ptr<sd_range_asm::Piece> pc(new sd_range_asm::Piece);
        pc->read = read;
        h.rows_of(c0, c1, lo, recs, roff, pc->rows);
        pc->sp.b = pc->rows.data();
        pc->sp.n = pc->rows.size();
        pc->sp.open_front = of;
        pc->sp.open_back = ob;
        if (pc->sp.n < (size_t)sd::SEAM_MIN_PIECE) { ok = false; return pc; }
        pc->sp.scan_assumed();
        const std::vector<uint32_t>& kp = pc->sp.kept;
        if (pc->sp.body_from < kp.size())
            h.format_idx(*pc, kp, pc->sp.body_from, kp.size(), pc->sp.body_from ? pc->rows[kp[pc->sp.body_from - 1]].end : 0, pc->body);
        return pc;
    };
    const double t1 = now_s();
    if (open_front && open_back && ra == rb) {
        h.front = make_piece(ra, lo, hi, true, true);
        mid_lo = mid_hi = hi;
    } else {
        if (open_front) { mid_lo = cstart[ra + 1]; h.front = make_piece(ra, lo, mid_lo, true, false); }
        if (open_back) { mid_hi = cstart[rb]; h.back = make_piece(rb, mid_hi, hi, false, true); }
    }
    const double t2 = now_s();
    if (mid_hi > mid_lo) {
        std::string unused;
        ReadAssembler as(h.reads, h.table, h.nch, h.tnames, h.threads, unused);
        as.part_sink = &h.middle;
        as.next_read = (size_t)h.table[mid_lo].read;
        const size_t step = 4096;
        std::vector<int64_t> ro;
        for (size_t c0 = mid_lo; c0 < mid_hi; c0 += step) {
            const size_t c1 = std::min(mid_hi, c0 + step);
            ro.resize(c1 - c0 + 1);
            for (size_t c = c0; c <= c1; ++c) ro[c - c0] = roff[c - lo] - roff[c0 - lo];
            as.add(c0, c1, recs + roff[c0 - lo], ro.data());
        }
    }
    const double t3 = now_s();
    edge->ok = ok ? 1 : 0;
    edge->has_front = open_front ? 1 : 0;
    edge->has_back = open_back ? 1 : 0;
    edge->through = (open_front && open_back && ra == rb) ? 1 : 0;
    if (ok) {
        if (open_front) {
            const sd_range_asm::Piece& f = *h.front;
            for (int k = 0; k < 8; ++k) { edge->head[k][0] = f.rows[(size_t)k].start; edge->head[k][1] = f.rows[(size_t)k].end; }
        }
        if (open_back) {
            const sd_range_asm::Piece& b = edge->through ? *h.front : *h.back;
            const size_t n = b.rows.size();
            for (int k = 0; k < 8; ++k) { edge->tail[k][0] = b.rows[n - 8 + (size_t)k].start; edge->tail[k][1] = b.rows[n - 8 + (size_t)k].end; }
            for (int e = 0; e < 8; ++e) edge->exit_of[e] = (int8_t)(edge->through ? b.sp.exit_of(e) : (int)(b.sp.exit0 - b.sp.stop()));
        }
    }
    h.st[0] = (now_s() - t0) * 1e3;
    h.st[1] = (t3 - t2) * 1e3;
    h.st[2] = (t2 - t1) * 1e3;
    return SD_OK;
}

int sd_range_assemble_begin(const char* const* read_names, const int64_t* read_lens, int32_t n_reads,
                            const char* const* mono_names, int32_t n_mono, const sd_params* p, int64_t chunk_lo,
                            int64_t chunk_hi, const sd_rec* recs, const int64_t* rec_off, sd_seam_edge* edge,
                            sd_range_asm** hout, char* errbuf, size_t errlen) {
    if (!hout || !edge || !read_names || !read_lens || !mono_names || !rec_off || n_reads < 0 || n_mono <= 0) return SD_ERR_PARAM;
    *hout = nullptr;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::unique_ptr<sd_range_asm> h(new sd_range_asm);
    h->rname_store.reserve((size_t)n_reads);
    for (int32_t r = 0; r < n_reads; ++r) h->rname_store.emplace_back(read_names[r]);
    for (int32_t r = 0; r < n_reads; ++r)
        h->reads.push_back(ReadView{h->rname_store[(size_t)r].c_str(), h->rname_store[(size_t)r].size(), nullptr, read_lens[r]});
    for (int32_t m = 0; m < n_mono; ++m) h->tnames.emplace_back(mono_names[m]);
    for (int32_t m = 0; m < n_mono; ++m) h->tnames.push_back(std::string(mono_names[m]) + "'");
    if (chunk_hi > chunk_lo && !recs && rec_off[chunk_hi - chunk_lo] > 0) return SD_ERR_PARAM;
    rc = range_asm_begin(h, p, chunk_lo, chunk_hi, recs, rec_off, edge, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    *hout = h.release();
    return SD_OK;
}

int sd_range_assemble_begin_files(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                                  int32_t world, const sd_rec* recs, const int64_t* rec_off, sd_seam_edge* edge,
                                  sd_range_asm** hout, char* errbuf, size_t errlen) {
    if (!hout || !edge || !reads_fa || !monomers_fa || !rec_off || world < 1 || rank < 0 || rank >= world) return SD_ERR_PARAM;
    *hout = nullptr;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    sd::FastaFile rf, mf;
    rc = rf.open(reads_fa, p->threads, err);
    if (rc == SD_OK) rc = mf.open(monomers_fa, p->threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::unique_ptr<sd_range_asm> h(new sd_range_asm);
    h->rname_store.reserve(rf.recs.size());
    for (const auto& r : rf.recs) h->rname_store.emplace_back(r.name, r.name_len);
    for (size_t r = 0; r < rf.recs.size(); ++r)
        h->reads.push_back(ReadView{h->rname_store[r].c_str(), h->rname_store[r].size(), nullptr, rf.recs[r].len});
    for (const auto& r : mf.recs) h->tnames.emplace_back(r.name, r.name_len);
    for (const auto& r : mf.recs) h->tnames.push_back(std::string(r.name, r.name_len) + "'");
    if (h->tnames.empty()) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    int64_t n = 0;
    for (const ReadView& r : h->reads) n += sd::chunk_plan(r.len, p->part_size, p->overlap, [](int64_t, int32_t) {});
    const int64_t base = n / world, extra = n % world;
    const int64_t lo = rank * base + std::min<int64_t>(rank, extra);
    const int64_t hi = lo + base + (rank < extra ? 1 : 0);
    rc = range_asm_begin(h, p, lo, hi, recs, rec_off, edge, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    *hout = h.release();
    return SD_OK;
}

static int range_asm_from_files(sd::FastaFile& rf, sd::FastaFile& mf, const sd_params* p, int64_t lo, int64_t hi,
                                sd_rec* recs, int64_t* off, sd_seam_edge* edge, sd_range_asm** hout, std::string& err) {
    std::unique_ptr<sd_range_asm> h(new sd_range_asm);
    h->own_recs = recs;
    h->own_off = off;
    h->own_chunks = hi - lo;
    h->rname_store.reserve(rf.recs.size());
    for (const auto& r : rf.recs) h->rname_store.emplace_back(r.name, r.name_len);
    for (size_t r = 0; r < rf.recs.size(); ++r)
        h->reads.push_back(ReadView{h->rname_store[r].c_str(), h->rname_store[r].size(), nullptr, rf.recs[r].len});
    for (const auto& r : mf.recs) h->tnames.emplace_back(r.name, r.name_len);
    for (const auto& r : mf.recs) h->tnames.push_back(std::string(r.name, r.name_len) + "'");
    const int rc = range_asm_begin(h, p, lo, hi, recs, off, edge, err);
    if (rc) return rc;
    *hout = h.release();
    return SD_OK;
}

int sd_range_assemble_records(sd_range_asm* h, const sd_rec** recs, const int64_t** rec_off, int64_t* n_chunks) {
    if (!h || !recs || !rec_off || !n_chunks || !h->own_off) return SD_ERR_PARAM;
    *recs = h->own_recs;
    *rec_off = h->own_off;
    *n_chunks = h->own_chunks;
    return SD_OK;
}

int sd_range_assemble_text(sd_range_asm* h, const sd_seam_edge* edges, int32_t world, int32_t rank, int64_t* text_bytes,
                           char* errbuf, size_t errlen) {
    if (!h || !edges || !text_bytes || world < 1 || rank < 0 || rank >= world || h->text_done) return SD_ERR_PARAM;
    const double t0 = now_s();
    const sd::SeamEntry en = sd::seam_resolve(edges, world, rank);
    if (!en.ok) { set_err(errbuf, errlen, "a share of this job cannot assemble its own range"); return SD_ERR_UNSUPPORTED; }
    {   // the edges must describe THIS handle at position `rank` (a caller that mixed up the order would get wrong text)
        const sd_range_asm::Piece* bk = h->back ? h->back.get() : (h->front && h->front->sp.open_back ? h->front.get() : nullptr);
        const bool has_front = h->front && h->front->sp.open_front;
        if ((edges[rank].has_front != 0) != has_front || (edges[rank].has_back != 0) != (bk != nullptr)) {
            set_err(errbuf, errlen, "edges[rank] is not this share's edge");
            return SD_ERR_PARAM;
        }
    }
    int64_t printed = 0;
    auto finish = [&](sd_range_asm::Piece& pc, int e, int32_t prev_end, const int32_t (*next_head)[2]) {
        sd::SeamPiece& sp = pc.sp;
        size_t exit_pos = sp.exit0;
        int32_t last_end = prev_end;   // end of the last row printed before the last zone
        if (sp.open_front) {
            std::vector<uint32_t> head, all;
            std::vector<std::string> txt;
            if (sp.head_rows(e, head, all, exit_pos)) {
                h->format_idx(pc, head, 0, head.size(), prev_end, txt);
                printed += (int64_t)head.size();
                if (!pc.body.empty()) last_end = pc.rows[sp.kept.back()].end;
                else if (!head.empty()) last_end = pc.rows[head.back()].end;
            } else {
                // the real scan met the assumed one behind the part made ahead (or never): this piece again
                pc.body.clear();
                h->format_idx(pc, all, 0, all.size(), prev_end, txt);
                printed += (int64_t)all.size();
                h->st[5] = 1;
                if (!all.empty()) last_end = pc.rows[all.back()].end;
            }
            for (const std::string& t : txt) pc.head += t;
        } else if (!sp.kept.empty()) {
            last_end = pc.rows[sp.kept.back()].end;
        }
        if (sp.open_back) {
            int32_t tl[8][2];
            const size_t n = sp.n;
            for (int k = 0; k < 8; ++k) { tl[k][0] = pc.rows[n - 8 + (size_t)k].start; tl[k][1] = pc.rows[n - 8 + (size_t)k].end; }
            std::vector<uint32_t> tk;
            int32_t pe_unused = 0;
            sd::seam_window(tl, next_head, (int)(exit_pos - sp.stop()), pe_unused, [&](int k) { tk.push_back((uint32_t)(n - 8 + (size_t)k)); });
            std::vector<std::string> txt;
            h->format_idx(pc, tk, 0, tk.size(), last_end, txt);
            for (const std::string& t : txt) pc.tail += t;
            printed += (int64_t)tk.size();
        }
    };
    const sd_seam_edge& me = edges[rank];
    const int32_t (*next_head)[2] = rank + 1 < world ? edges[rank + 1].head : nullptr;
    if (h->front) finish(*h->front, en.e, en.prev_end, next_head);
    if (h->back) finish(*h->back, 0, 0, next_head);
    (void)me;
    h->order.clear();
    if (h->front) {
        h->order.push_back(&h->front->head);
        for (const std::string& t : h->front->body) h->order.push_back(&t);
        h->order.push_back(&h->front->tail);
    }
    for (const std::string& t : h->middle) h->order.push_back(&t);
    if (h->back) {
        h->order.push_back(&h->back->head);
        for (const std::string& t : h->back->body) h->order.push_back(&t);
        h->order.push_back(&h->back->tail);
    }
    int64_t total = 0;
    for (const std::string* t : h->order) total += (int64_t)t->size();
    *text_bytes = total;
    h->text_done = true;
    h->st[3] = (now_s() - t0) * 1e3;
    h->st[4] = (double)printed;
    return SD_OK;
}

int sd_range_assemble_write(sd_range_asm* h, const char* path, int64_t offset, int64_t file_bytes, char* errbuf, size_t errlen) {
    if (!h || !path || offset < 0 || !h->text_done) return SD_ERR_PARAM;
    const double t0 = now_s();
    // file_bytes >= 0: every rank creates the file if it is not there and sets its size (the same value on every rank,
    // so the order of the ranks does not matter and no rank waits for another before it writes)
    const int fd = file_bytes >= 0 ? ::open(path, O_WRONLY | O_CREAT, 0666) : ::open(path, O_WRONLY);
    if (fd < 0) { set_err(errbuf, errlen, std::string("cannot write ") + path); return SD_ERR_IO; }
    if (file_bytes >= 0 && ::ftruncate(fd, (off_t)file_bytes) != 0) {
        ::close(fd);
        set_err(errbuf, errlen, std::string("cannot size ") + path);
        return SD_ERR_IO;
    }
    struct Ref { const std::string* s; size_t size() const { return s->size(); } const char* data() const { return s->data(); } bool empty() const { return s->empty(); } };
    std::vector<Ref> parts;
    for (const std::string* t : h->order) parts.push_back(Ref{t});
    int64_t off = offset;
    const bool ok = sd::write_parts(fd, off, parts, h->threads);
    if (::close(fd) != 0 || !ok) { set_err(errbuf, errlen, std::string("short write to ") + path); return SD_ERR_IO; }
    h->st[6] = (now_s() - t0) * 1e3;
    return SD_OK;
}

int sd_range_assemble_copy(sd_range_asm* h, char* buf, int64_t room) {
    if (!h || !h->text_done || (!buf && room > 0)) return SD_ERR_PARAM;
    int64_t at = 0;
    for (const std::string* t : h->order) {
        if (at + (int64_t)t->size() > room) return SD_ERR_PARAM;
        if (!t->empty()) std::memcpy(buf + at, t->data(), t->size());
        at += (int64_t)t->size();
    }
    return SD_OK;
}

void sd_range_assemble_stats(sd_range_asm* h, double out[8]) {
    if (!h || !out) return;
    for (int k = 0; k < 8; ++k) out[k] = h->st[k];
}

void sd_range_assemble_free(sd_range_asm* h) { delete h; }

// -------------------------------------------------------------------------------------------
// streaming form: sequences in host memory -> rows in host memory (AlignReadsSet, main.cpp:67-122,
// without the text), jobs pipelined through the device in sub-batches
// -------------------------------------------------------------------------------------------
namespace {
// Per-read assembly of one job into rows (chunk offsets main.cpp:109-111, seam merge :116, :287-302).
// Batches arrive in chunk-table order; reads that lie completely inside a batch are assembled in
// parallel, a read that spans batches goes through `carry`.
struct RowJob {
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    int32_t n_reads = 0;
    int threads = 1;
    sd_rec* rows = nullptr;       // malloc'ed, handed to the caller by collect
    size_t n_rows = 0, cap_rows = 0;
    int64_t* row_off = nullptr;   // n_reads + 1
    std::vector<sd_rec> carry, tmp;
    size_t next_read = 0;         // first read not complete yet
    int32_t chunks_seen = 0;      // chunks of next_read already in carry
    int batches_left = 0;
    bool oom = false;
    // In-stream identities (sd_ident.hip) follow their records through the merge BY REFERENCE: `per` words per record
    // in up to two arrays (plain / homopolymer-compressed) that stay where the fetch put them (pinned).  per == 0: not
    // tracked.  bid / bidh = the arrays of the batch being added (set by the caller before add).  rsrc[row] >= 0: record
    // index in those arrays; < 0: -1 - k, entry k of xid / xidh -- the words of rows of a read that began in an
    // earlier batch, carried by value.
    int per = 0;
    const uint32_t* bid = nullptr;
    const uint32_t* bidh = nullptr;
    int64_t* rsrc = nullptr;      // malloc'ed with rows
    std::vector<uint32_t> xid, xidh;
    bool ident_ok = true;         // every batch of the rows assembled so far came with identities
    std::vector<uint32_t> carry_id, carry_idh;
    std::vector<int64_t> src_tmp, carry_src;
    ~RowJob() { std::free(rows); std::free(row_off); std::free(rsrc); }
    void reserve(size_t need) {
        if (need <= cap_rows) return;
        size_t nc = std::max<size_t>(need, cap_rows * 2 + 4096);
        sd_rec* q = static_cast<sd_rec*>(std::realloc(rows, nc * sizeof(sd_rec)));
        if (!q) { oom = true; return; }
        rows = q;
        if (per) {
            int64_t* a = static_cast<int64_t*>(std::realloc(rsrc, nc * sizeof(int64_t)));
            if (!a) { oom = true; return; }
            rsrc = a;
        }
        cap_rows = nc;
    }
    void carry_push(const sd_rec& t, int64_t x) {
        carry.push_back(t);
        if (!per) return;
        if (!bid) { ident_ok = false; carry_id.resize(carry.size() * (size_t)per, 0u); carry_idh.resize(carry.size() * (size_t)per, 0u); return; }
        carry_id.insert(carry_id.end(), bid + (size_t)x * (size_t)per, bid + (size_t)(x + 1) * (size_t)per);
        if (bidh) carry_idh.insert(carry_idh.end(), bidh + (size_t)x * (size_t)per, bidh + (size_t)(x + 1) * (size_t)per);
        else carry_idh.resize(carry.size() * (size_t)per, 0u);
    }
    void add(size_t c0, size_t c1, const sd_rec* recs, const int64_t* roff) {
        size_t c = c0;
        if (per && !bid) ident_ok = false;
        // (1) the read that began in an earlier batch
        if (chunks_seen > 0) {
            while (c < c1 && chunks_seen < nch[next_read]) {
                const int32_t add = (int32_t)table[c].off;
                for (int64_t x = roff[c - c0]; x < roff[c - c0 + 1]; ++x) {
                    sd_rec t = recs[x];
                    t.start += add; t.end += add;
                    carry_push(t, x);
                }
                ++c; ++chunks_seen;
            }
            if (chunks_seen < nch[next_read]) return;  // still open
            carry_src.resize(carry.size());
            for (size_t k = 0; k < carry.size(); ++k) carry_src[k] = (int64_t)k;
            const size_t n = sd::seam_merge_inplace(carry.data(), carry_src.data(), carry.size());
            reserve(n_rows + n);
            if (oom) return;
            if (n) std::memcpy(rows + n_rows, carry.data(), n * sizeof(sd_rec));
            if (per)
                for (size_t k = 0; k < n; ++k) {
                    const size_t from = (size_t)carry_src[k] * (size_t)per, xk = xid.size() / (size_t)per;
                    xid.insert(xid.end(), carry_id.begin() + (long)from, carry_id.begin() + (long)(from + (size_t)per));
                    xidh.insert(xidh.end(), carry_idh.begin() + (long)from, carry_idh.begin() + (long)(from + (size_t)per));
                    rsrc[n_rows + k] = -1 - (int64_t)xk;
                }
            n_rows += n;
            row_off[next_read + 1] = (int64_t)n_rows;
            carry.clear();
            carry_id.clear();
            carry_idh.clear();
            chunks_seen = 0;
            ++next_read;
        }
        // (2) reads completely inside [c, c1): parallel
        struct Item { size_t read, ca, cb; size_t n; };
        std::vector<Item> items;
        size_t r = next_read, cc = c;
        while (r < (size_t)n_reads && cc + (size_t)nch[r] <= c1) {
            items.push_back(Item{r, cc, cc + (size_t)nch[r], 0});
            cc += (size_t)nch[r];
            ++r;
        }
        if (!items.empty()) {
            const int64_t lo = roff[c - c0], hi = roff[cc - c0];
            tmp.resize((size_t)(hi - lo));
            if (per) src_tmp.resize((size_t)(hi - lo));
            sd::parallel_for((int64_t)items.size(), threads, 8, [&](int64_t q) {
                Item& it = items[(size_t)q];
                sd_rec* dst = tmp.data() + (roff[it.ca - c0] - lo);
                int64_t* sdst = per ? src_tmp.data() + (roff[it.ca - c0] - lo) : nullptr;
                size_t k = 0;
                for (size_t ch = it.ca; ch < it.cb; ++ch) {
                    const int32_t add = (int32_t)table[ch].off;
                    for (int64_t x = roff[ch - c0]; x < roff[ch - c0 + 1]; ++x) {
                        sd_rec t = recs[x];
                        t.start += add; t.end += add;
                        if (sdst) sdst[k] = x;
                        dst[k++] = t;
                    }
                }
                it.n = sdst ? sd::seam_merge_inplace(dst, sdst, k) : sd::seam_merge_inplace(dst, k);
            });
            size_t total = 0;
            for (const Item& it : items) total += it.n;
            reserve(n_rows + total);
            if (oom) return;
            std::vector<size_t> at(items.size());
            for (size_t q = 0; q < items.size(); ++q) {
                const Item& it = items[q];
                at[q] = n_rows;
                std::memcpy(rows + n_rows, tmp.data() + (roff[it.ca - c0] - lo), it.n * sizeof(sd_rec));
                n_rows += it.n;
                row_off[it.read + 1] = (int64_t)n_rows;
            }
            if (per)   // where the identity words of the kept records are (the words themselves stay in the batch's arrays)
                for (size_t q = 0; q < items.size(); ++q) {
                    const Item& it = items[q];
                    std::memcpy(rsrc + at[q], src_tmp.data() + (roff[it.ca - c0] - lo), it.n * sizeof(int64_t));
                }
            next_read = r;
            c = cc;
        }
        // (3) the read that continues in the next batch
        while (c < c1) {
            const int32_t add = (int32_t)table[c].off;
            for (int64_t x = roff[c - c0]; x < roff[c - c0 + 1]; ++x) {
                sd_rec t = recs[x];
                t.start += add; t.end += add;
                carry_push(t, x);
            }
            ++c; ++chunks_seen;
        }
    }
};
}  // namespace

struct sd_stream {
    sd_params p{};
    std::vector<std::string> mono;       // owned copies
    Pipeline pipe;
    int sub_batches = 1;
    std::vector<std::unique_ptr<RowJob>> jobs;   // FIFO: submitted, not collected yet
    int64_t budget = 0;
    double submit_s = 0, collect_s = 0;
    int64_t n_jobs = 0;
};

int sd_stream_create(sd_stream** out, const sd_params* p, const char* const* mono_seqs,
                     const int32_t* mono_lens, int32_t n_mono, int32_t sub_batches, char* errbuf, size_t errlen) {
    if (!out) return SD_ERR_PARAM;
    *out = nullptr;
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (n_mono <= 0 || !mono_seqs || !mono_lens) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    std::unique_ptr<sd_stream> s(new sd_stream);
    s->p = *p;
    s->sub_batches = std::max(1, (int)sub_batches);
    std::vector<const char*> ms;
    for (int32_t m = 0; m < n_mono; ++m) {
        if (mono_lens[m] <= 0) { set_err(errbuf, errlen, "ERROR: empty monomer sequence"); return SD_ERR_EMPTY; }
        s->mono.emplace_back(mono_seqs[m], (size_t)mono_lens[m]);
    }
    for (const std::string& m : s->mono) ms.push_back(m.data());
    rc = s->pipe.create(p, ms.data(), mono_lens, n_mono);
    if (rc) { set_err(errbuf, errlen, s->pipe.eb); return rc; }
    s->budget = s->pipe.row_budget();
    *out = s.release();
    return SD_OK;
}

void sd_stream_destroy(sd_stream* s) { delete s; }

int sd_stream_submit(sd_stream* s, const char* const* read_seqs, const int64_t* read_lens, int32_t n_reads,
                     char* errbuf, size_t errlen) {
    if (!s || n_reads < 0 || (n_reads && (!read_seqs || !read_lens))) return SD_ERR_PARAM;
    const double t0 = now_s();
    std::unique_ptr<RowJob> job(new RowJob);
    job->n_reads = n_reads;
    job->threads = s->p.threads;
    job->nch.assign((size_t)n_reads, 0);
    for (int32_t r = 0; r < n_reads; ++r) {
        if (read_lens[r] <= 0) { set_err(errbuf, errlen, "ERROR: Sequence #" + std::to_string(r) + " is empty"); return SD_ERR_EMPTY; }
        job->nch[(size_t)r] = sd::chunk_plan(read_lens[r], s->p.part_size, s->p.overlap,
                                             [&](int64_t off, int32_t l) { job->table.push_back(CRef{r, off, l}); });
    }
    job->row_off = static_cast<int64_t*>(std::calloc((size_t)n_reads + 1, sizeof(int64_t)));
    if (!job->row_off) { set_err(errbuf, errlen, "out of host memory"); return SD_ERR_INTERNAL; }
    std::vector<std::pair<size_t, size_t>> batches;
    plan_batches(job->table, 0, job->table.size(), s->budget, s->sub_batches, batches);
    job->batches_left = (int)batches.size();
    RowJob* jp = job.get();
    s->jobs.push_back(std::move(job));
    ++s->n_jobs;
    std::vector<const char*> cptr;
    std::vector<int32_t> clen;
    int rc = SD_OK;
    for (size_t b = 0; b < batches.size() && rc == SD_OK; ++b) {
        const size_t c0 = batches[b].first, c1 = batches[b].second;
        cptr.clear();
        clen.clear();
        for (size_t c = c0; c < c1; ++c) {
            cptr.push_back(read_seqs[jp->table[c].read] + jp->table[c].off);
            clen.push_back(jp->table[c].len);
        }
        rc = s->pipe.push(cptr, clen, [jp, c0, c1](const sd_rec* r, const int64_t* ro, size_t first, size_t n) {
            jp->add(c0 + first, c0 + first + n, r, ro);
            if (c0 + first + n == c1) --jp->batches_left;
        });
    }
    if (rc) {
        set_err(errbuf, errlen, s->pipe.eb);
        (void)s->pipe.drain();   // sinks of older jobs still run; this job is dropped
        for (size_t j = 0; j < s->jobs.size(); ++j)
            if (s->jobs[j].get() == jp) { s->jobs.erase(s->jobs.begin() + (long)j); break; }
    }
    s->submit_s += now_s() - t0;
    return rc;
}

int sd_stream_collect(sd_stream* s, sd_rec** rows, int64_t** row_off, int64_t* n_rows, char* errbuf, size_t errlen) {
    if (!s || !rows || !row_off) return SD_ERR_PARAM;
    *rows = nullptr;
    *row_off = nullptr;
    if (n_rows) *n_rows = 0;
    if (s->jobs.empty()) { set_err(errbuf, errlen, "sd_stream_collect without a submitted job"); return SD_ERR_PARAM; }
    const double t0 = now_s();
    RowJob* job = s->jobs.front().get();
    int rc = SD_OK;
    while (job->batches_left > 0 && rc == SD_OK) {
        if (s->pipe.inflight() == 0) { set_err(errbuf, errlen, "stream lost a batch"); rc = SD_ERR_INTERNAL; break; }
        rc = s->pipe.pop();
        if (rc) set_err(errbuf, errlen, s->pipe.eb);
    }
    if (rc != SD_OK) {
        // batches of this or a later job may still be in flight and their sinks hold pointers to the jobs: wait for
        // every one of them before a job is freed (as sd_stream_submit's error path does), then drop all jobs
        (void)s->pipe.drain();
        s->jobs.clear();
        s->collect_s += now_s() - t0;
        return rc;
    }
    if (rc == SD_OK && job->oom) { set_err(errbuf, errlen, "out of host memory"); rc = SD_ERR_INTERNAL; }
    if (rc == SD_OK) {
        if (!job->rows) job->rows = static_cast<sd_rec*>(std::malloc(sizeof(sd_rec)));
        *rows = job->rows;
        *row_off = job->row_off;
        if (n_rows) *n_rows = (int64_t)job->n_rows;
        job->rows = nullptr;      // ownership moves to the caller (sd_free)
        job->row_off = nullptr;
    }
    s->jobs.erase(s->jobs.begin());
    s->collect_s += now_s() - t0;
    return rc;
}

int sd_stream_stats(sd_stream* s, double out[16]) {
    if (!s || !out) return SD_ERR_PARAM;
    const Pipeline& q = s->pipe;
    const double v[16] = {q.fill_ms, q.trace_ms, q.compact_ms, q.run_ms, (double)q.launches, (double)q.batches,
                          (double)q.rows, q.pack_s * 1e3, q.wait_s * 1e3, q.sink_s * 1e3, s->submit_s * 1e3,
                          s->collect_s * 1e3, (double)s->n_jobs, (double)s->sub_batches, (double)s->budget, 0.0};
    std::memcpy(out, v, sizeof v);
    return SD_OK;
}

int sd_stream_info(sd_stream* s, int64_t info[8]) {
    if (!s) return SD_ERR_PARAM;
    return sd_engine_info(s->pipe.eng[0], info);
}

// Host stages of the path alone, no device (for sizing the host side of a multi-GPU node: SURVEY 8(e) wants
// the host to feed >= 7x one GPU): (a) chunk table + 2-bit packing of the reads into a host buffer, as
// load_chunks_impl does, (b) per-read assembly + raw TSV text of one synthetic record per 171 bases per
// chunk (offsets, seam merge, SaveBatch formatting), as sd_decompose's sink does.
int sd_host_stage_rates(const char* const* read_seqs, const int64_t* read_lens, int32_t n_reads, const sd_params* p,
                        int32_t iters, double out[4]) {
    std::string err;
    if (validate_params(p, err) || !out || n_reads < 0 || (n_reads && (!read_seqs || !read_lens)) || iters < 1) return SD_ERR_PARAM;
    std::vector<ReadView> reads((size_t)n_reads);
    int64_t bp = 0;
    for (int32_t r = 0; r < n_reads; ++r) { reads[(size_t)r] = ReadView{"read", 4, read_seqs[r], read_lens[r]}; bp += read_lens[r]; }
    std::vector<CRef> table;
    std::vector<int32_t> nch;
    build_chunk_table(reads, p, table, nch);
    const size_t C = table.size();
    std::vector<uint32_t> woff(C + 1, 0);
    for (size_t c = 0; c < C; ++c) woff[c + 1] = woff[c] + (uint32_t)((table[c].len + 15) / 16);
    std::vector<uint32_t> words(woff[C] + 1);
    double t0 = now_s();
    for (int it = 0; it < iters; ++it)
        sd::parallel_for((int64_t)C, p->threads, 16, [&](int64_t c) {
            (void)sd::pack_chunk(reads[(size_t)table[(size_t)c].read].seq + table[(size_t)c].off, table[(size_t)c].len,
                                 words.data() + woff[(size_t)c]);
        });
    out[0] = (double)bp * iters / std::max(now_s() - t0, 1e-9);
    // synthetic records: one per 171 bases, chunk-local coordinates
    std::vector<sd_rec> recs;
    std::vector<int64_t> roff(C + 1, 0);
    for (size_t c = 0; c < C; ++c) {
        for (int32_t a = 0; a < table[c].len; a += 171)
            recs.push_back(sd_rec{(int32_t)((a / 171) % 24), a, std::min(a + 170, table[c].len - 1), 100});
        roff[c + 1] = (int64_t)recs.size();
    }
    std::vector<std::string> tnames;
    for (int j = 0; j < 24; ++j) tnames.push_back("M" + std::to_string(j % 12) + (j >= 12 ? "'" : ""));
    size_t text = 0, rows = 0;
    t0 = now_s();
    for (int it = 0; it < iters; ++it) {
        std::string tsv;
        ReadAssembler as(reads, table, nch, tnames, p->threads, tsv);
        const size_t step = 4096;
        for (size_t c0 = 0; c0 < C; c0 += step) {
            const size_t c1 = std::min(C, c0 + step);
            std::vector<int64_t> ro(c1 - c0 + 1);
            for (size_t c = c0; c <= c1; ++c) ro[c - c0] = roff[c] - roff[c0];
            as.add(c0, c1, recs.data() + roff[c0], ro.data());
        }
        text = tsv.size();
        rows = (size_t)std::count(tsv.begin(), tsv.end(), '\n');
    }
    const double dt = std::max(now_s() - t0, 1e-9);
    out[1] = (double)bp * iters / dt;
    out[2] = (double)rows * iters / dt;
    out[3] = (double)text;
    return SD_OK;
}

int32_t sd_pack_bases(const char* seq, int64_t n, uint32_t* words, uint32_t* nmask) {
    if (!seq || n < 0 || n > 0x7fffffff || !words) return -1;
    const bool hn = sd::pack_chunk(seq, (int32_t)n, words);
    if (nmask) {
        std::memset(nmask, 0, sizeof(uint32_t) * (size_t)((n + 31) / 32));
        for (int64_t i = 0; i < n; ++i)
            if (seq[i] == 'N') nmask[i >> 5] |= 1u << (i & 31);
    }
    return hn ? 1 : 0;
}

// sd::write_parts alone (how sd_run_files puts a batch's text into its files): `n_parts` parts of `part_bytes`
// bytes appended to `path` twice (two calls, the second at the first's end offset), then read back and compared.
// fail_reserve != 0 makes the page reservation of the mapped (tmpfs) path fail, as on a full /dev/shm: the text
// must then arrive through the pwritev loop.  out (may be null): [0] bytes written, [1] 1 if the file system is tmpfs / ramfs.
int sd_write_parts_selftest(const char* path, int32_t n_parts, int64_t part_bytes, int32_t threads, int32_t fail_reserve,
                            int64_t out[2]) {
    if (!path || n_parts < 1 || part_bytes < 0 || threads < 1) return SD_ERR_PARAM;
    std::vector<std::string> parts((size_t)n_parts);
    for (int32_t i = 0; i < n_parts; ++i) {
        parts[(size_t)i].resize((size_t)part_bytes);
        for (int64_t b = 0; b < part_bytes; ++b) parts[(size_t)i][(size_t)b] = (char)('a' + (i * 7 + b * 13) % 26);
    }
    const int fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return SD_ERR_IO;
    sd::write_parts_hook().store(fail_reserve ? 1 : 0);
    int64_t off = 0;
    const double t_w0 = now_s();
    bool ok = sd::write_parts(fd, off, parts, threads) && sd::write_parts(fd, off, parts, threads);
    if (getenv("SD_TIMING")) std::fprintf(stderr, "[sd timing] write_parts: %lld bytes in %.2f ms\n", (long long)off, (now_s() - t_w0) * 1e3);
    sd::write_parts_hook().store(0);
    struct statfs fs;
    const bool ram = ::fstatfs(fd, &fs) == 0 && ((unsigned long)fs.f_type == 0x01021994ul || (unsigned long)fs.f_type == 0x858458f6ul);
    struct stat st;
    ok = ok && ::fstat(fd, &st) == 0 && (int64_t)st.st_size == off && off == 2 * (int64_t)n_parts * part_bytes;
    if (ok) {
        std::string back((size_t)part_bytes, '\0');
        for (int rep = 0; rep < 2 && ok; ++rep)
            for (int32_t i = 0; i < n_parts && ok; ++i) {
                const int64_t at = ((int64_t)rep * n_parts + i) * part_bytes;
                ok = ::pread(fd, &back[0], (size_t)part_bytes, (off_t)at) == (ssize_t)part_bytes && back == parts[(size_t)i];
            }
    }
    ::close(fd);
    if (out) { out[0] = off; out[1] = ram ? 1 : 0; }
    return ok ? SD_OK : SD_ERR_IO;
}

// -------------------------------------------------------------------------------------------
// whole CLI job as one native call: FASTA files -> raw TSV + final TSV + _alt TSV, streamed per
// device batch (main.py:186-197 run + :168-184 convert_tsv without the round trip through the raw file)
// -------------------------------------------------------------------------------------------
// rank / world: this process handles the reads [lo, hi) of a split of the read set into `world` contiguous
// groups of about equal chunk counts (world == 1: everything).  *info (may be null): [0] first read, [1] one
// past the last read, [2] reads in the file, [3] chunks of this rank.  A read set that cannot be split by
// reads (one read holds more than half a rank's share, e.g. a single chromosome) gives SD_ERR_UNSUPPORTED
// before anything is written; the caller then shards by chunk range instead.
// stage times of the last sd_run_files / sd_run_files_range call of this process (sd_last_run_stats)
static std::mutex g_last_m;
static double g_last_run[24] = {0};

// The three texts of one hand-over of sd_run_files (raw / final / _alt parts), and the process-wide pool their buffers
// return to (at most four; sd_release_cache() frees them).  Never destroyed at exit (as the other pools).
struct TextJob { std::vector<std::string> raw, fin; std::vector<sd::TextBuf> alt; };
struct TextPool {
    std::mutex m;
    std::deque<TextJob> free_;
    TextJob take() {
        std::lock_guard<std::mutex> g(m);
        TextJob j;
        if (!free_.empty()) { j = std::move(free_.front()); free_.pop_front(); }
        return j;
    }
    void give(TextJob&& j) {
        TextJob drop;   // freed outside the lock
        std::lock_guard<std::mutex> g(m);
        if (free_.size() < 4) free_.push_back(std::move(j)); else drop = std::move(j);
    }
    void clear() {
        std::deque<TextJob> drop;
        std::lock_guard<std::mutex> g(m);
        drop.swap(free_);
    }
};
static TextPool& g_textpool_ref() { static TextPool* p = new TextPool; return *p; }
#define g_textpool g_textpool_ref()
static void text_pool_clear() { g_textpool.clear(); }

static int run_files_impl(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank, int32_t world,
                          const char* raw_tsv_out, const char* final_tsv_out, const char* alt_tsv_out,
                          int32_t min_identity, int32_t second_best, const double* lr_coef, int64_t* info,
                          char* errbuf, size_t errlen, const char* records_out = nullptr) {
    std::string err;
    int rc = validate_params(p, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    if (!reads_fa || !monomers_fa || !raw_tsv_out || !final_tsv_out || !alt_tsv_out || !lr_coef || world < 1 || rank < 0 ||
        rank >= world)
        return SD_ERR_PARAM;
    if (records_out && world != 1) { set_err(errbuf, errlen, "the record stream is written by a single process"); return SD_ERR_PARAM; }
    const bool timing = getenv("SD_TIMING") != nullptr;
    const double t_begin = now_s();
    double t_prev = t_begin;
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double t = now_s();
        std::fprintf(stderr, "[sd timing] %-34s %9.2f ms\n", what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    sd::FastaFile rf, mf;
    const bool progress = (p->reserved[1] & SD_FLAG_PROGRESS) != 0 && rank == 0;
    if (progress)   // main.cpp:393
        std::fprintf(stderr, "Scores: insertion=%d deletion=%d mismatch=%d match=%d\n", p->ins, p->del, p->mismatch, p->match);
    rc = rf.open(reads_fa, p->threads, err);                                  // main.cpp:394
    if (rc == SD_OK && world == 1) rc = rf.validate(0, rf.recs.size(), p->threads, err);   // reads are checked first, as there
    if (rc == SD_OK) rc = mf.open(monomers_fa, p->threads, err);              // main.cpp:395
    if (rc == SD_OK) rc = mf.validate(0, mf.recs.size(), p->threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    std::vector<sd::Seq> monos;
    for (const auto& r : mf.recs) monos.push_back(sd::Seq{std::string(r.name, r.name_len), std::string(r.seq, (size_t)r.len)});
    if (monos.empty()) { set_err(errbuf, errlen, "no monomers"); return SD_ERR_PARAM; }
    std::vector<ReadView> all_reads;
    all_reads.reserve(rf.recs.size());
    for (const auto& r : rf.recs) {
        if (r.len <= 0) { set_err(errbuf, errlen, "ERROR: Sequence " + std::string(r.name, r.name_len) + " is empty"); return SD_ERR_EMPTY; }
        all_reads.push_back(ReadView{r.name, r.name_len, r.seq, r.len});
    }
    {
        // SeqIO.to_dict (main.py:65) refuses repeated read ids
        std::vector<std::pair<std::string, size_t>> nm;
        nm.reserve(all_reads.size());
        for (size_t r = 0; r < all_reads.size(); ++r) nm.emplace_back(std::string(all_reads[r].name, all_reads[r].name_len), r);
        std::sort(nm.begin(), nm.end());
        for (size_t i = 1; i < nm.size(); ++i)
            if (nm[i].first == nm[i - 1].first) { set_err(errbuf, errlen, "Duplicate key '" + nm[i].first + "'"); return SD_ERR_FORMAT; }
    }
    // this rank's reads: contiguous groups of about equal chunk counts
    size_t r_lo = 0, r_hi = all_reads.size();
    if (world > 1) {
        std::vector<int64_t> cum(all_reads.size() + 1, 0);
        int64_t biggest = 0;
        for (size_t r = 0; r < all_reads.size(); ++r) {
            const int64_t k = sd::chunk_plan(all_reads[r].len, p->part_size, p->overlap, [](int64_t, int32_t) {});
            cum[r + 1] = cum[r] + k;
            biggest = std::max(biggest, k);
        }
        const int64_t total = cum[all_reads.size()];
        if (biggest * 2 * world > total) {
            set_err(errbuf, errlen, "read set cannot be split by reads (one read holds more than half a rank's share)");
            return SD_ERR_UNSUPPORTED;
        }
        auto bound = [&](int g) {
            const int64_t want = total * g / world;
            return (size_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
        };
        r_lo = std::min(bound(rank), all_reads.size());
        r_hi = rank + 1 == world ? all_reads.size() : std::min(bound(rank + 1), all_reads.size());
        if (r_hi < r_lo) r_hi = r_lo;
    }
    if (world > 1) rc = rf.validate(r_lo, r_hi, p->threads, err);   // a rank checks the reads it touches (the launcher exchanges failures)
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    // main.cpp:343 (load_fasta): the N warning, once per file, on stderr
    for (const auto& ff : {std::make_pair(&rf, reads_fa), std::make_pair(&mf, monomers_fa)})
        if (ff.first->has_n && rank == 0)
            std::fprintf(stderr, "WARNING: sequences in %s contain N symbol. It will be counted as a separate symbol in scoring!\n", ff.second);
    lap("FASTA index + alphabet check");
    std::vector<ReadView> reads(all_reads.begin() + (long)r_lo, all_reads.begin() + (long)r_hi);
    if (info) { info[0] = (int64_t)r_lo; info[1] = (int64_t)r_hi; info[2] = (int64_t)all_reads.size(); info[3] = 0; }
    TemplateSet ts(monos);
    sd::PostProcessor pp;
    rc = pp.init(monos, min_identity, second_best != 0, lr_coef, p->device, p->threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    // the three outputs as plain descriptors: every batch's text is written by all host threads with pwrite at its
    // offset (sd::write_parts) -- the copy into the page cache is what a 300-MB _alt batch costs
    const int fr = ::open(raw_tsv_out, O_RDWR | O_CREAT | O_TRUNC, 0666);   // O_RDWR: write_parts maps the new range
    const int ff = fr >= 0 ? ::open(final_tsv_out, O_RDWR | O_CREAT | O_TRUNC, 0666) : -1;
    const int fa = ff >= 0 ? ::open(alt_tsv_out, O_RDWR | O_CREAT | O_TRUNC, 0666) : -1;
    int64_t off_r = 0, off_f = 0, off_a = 0;
    auto close_all = [&]() {
        bool ok = true;
        for (int f : {fr, ff, fa}) if (f >= 0 && ::close(f) != 0) ok = false;
        return ok;
    };
    if (fr < 0 || ff < 0 || fa < 0) {
        close_all();
        set_err(errbuf, errlen, std::string("cannot write ") + raw_tsv_out);
        return SD_ERR_IO;
    }
    sd::RecordsWriter rec_w;   // the rows once more as the binary record stream (sd_records.hpp), written as reads complete
    if (records_out) {
        rc = rec_w.open(records_out, *p, ts.tnames, err);
        if (rc) { close_all(); set_err(errbuf, errlen, err); return rc; }
    }
    // Round 6: the pages of the _alt file are reserved WHILE THE DEVICE RUNS THE DP.  A --second-best job writes 2T rows of
    // text per block (280 MB at BASELINE config 4) and the page-cache copy of that text bounded the job: on tmpfs the pages
    // of a new range are zeroed by ONE thread inside fallocate (36-49 ms per 280 MB, sd::write_parts), and that could only
    // begin when the first identities arrived, 20 ms into the job.  The size of the file is known closely from the reads
    // alone -- every base ends up in a block (main.cpp:217-269), a block prints one row per template -- so a helper thread
    // reserves that much in steps of 16 MB from the start of the job (short steps: write_parts' own fallocate of a range
    // that already has its pages, and the page faults of the copying threads, take the inode's lock in between); the file
    // is cut to its real size at the end.  Only with -i 0 (a higher threshold drops rows, main.py:152), only on tmpfs /
    // ramfs (where write_parts copies through a mapping), only for texts of at least 32 MB.
    // The reserved range is mapped ONCE for the job and the helper also fills its page tables (MADV_POPULATE_WRITE on pages
    // that exist is a walk, not an allocation): a hand-over's _alt text is then a plain parallel copy -- the per-hand-over
    // fallocate / mmap / 8 000 minor faults / munmap of write_parts made eight 34-MB writes take 5.5-7 ms each, back to
    // back on the writer thread from the first identities to 30 ms after the last (profiles/r06_c4_second_best_timeline.txt).
    std::thread prealloc;
    std::atomic<bool> pre_stop{false};
    std::atomic<int64_t> pre_done{0};
    char* alt_map = nullptr;
    int64_t alt_map_len = 0;
    int64_t alt_unmapped = 0;   // the mapping below this (page-aligned) offset is gone again
    {
        struct statfs fs;
        const bool ram = ::fstatfs(fa, &fs) == 0 && ((unsigned long)fs.f_type == 0x01021994ul || (unsigned long)fs.f_type == 0x858458f6ul);
        double lmean = 0, nmean = 0;
        for (const sd::Seq& m : monos) { lmean += (double)m.seq.size(); nmean += (double)m.name.size() + 0.5; }   // (half of the templates carry the "'")
        lmean /= std::max<size_t>(1, monos.size());
        nmean /= std::max<size_t>(1, monos.size());
        double est = 0;
        auto digits = [](int64_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; };
        for (const ReadView& r : reads)
            est += ((double)r.len / std::max(1.0, lmean) + 1.0) * (2.0 * (double)monos.size()) *
                   ((double)r.name_len + nmean + 2.0 * digits(r.len) + 5 + 1 + 6);
        const int64_t want = (int64_t)est;
        if (second_best && min_identity <= 0 && ram && want >= (32 << 20) && sd::write_parts_fallocate_ok() && !getenv("SD_ALT_PREALLOC_OFF"))
        {
            void* mp = getenv("SD_ALT_MAP_OFF") ? MAP_FAILED : ::mmap(nullptr, (size_t)want, PROT_READ | PROT_WRITE, MAP_SHARED, fa, 0);
            if (mp != MAP_FAILED) { alt_map = static_cast<char*>(mp); alt_map_len = want; }
            prealloc = std::thread([&, want]() {
                const double tp0 = now_s();
                const int64_t step = 16 << 20;
                for (int64_t at = 0; at < want && !pre_stop.load(std::memory_order_relaxed); at += step) {
                    const int64_t n = std::min(step, want - at);
                    if (::fallocate(fa, 0, (off_t)at, (off_t)n) != 0) break;   // (no space: write_parts reports it)
                    pre_done.store(at + n, std::memory_order_release);
#ifdef MADV_POPULATE_WRITE
                    if (alt_map) (void)::madvise(alt_map + at, (size_t)n, MADV_POPULATE_WRITE);
#endif
                }
                if (getenv("SD_TIMING"))
                    std::fprintf(stderr, "[sd timing] _alt pages reserved ahead: %lld of %lld bytes in %.1f ms (from %.1f ms into the job)\n",
                                 (long long)pre_done.load(), (long long)want, (now_s() - tp0) * 1e3, (tp0 - t_begin) * 1e3);
            });
        }
    }
    auto end_prealloc = [&]() {   // before the files are closed, on every path
        pre_stop.store(true);
        if (prealloc.joinable()) prealloc.join();
        if (pre_done.load() > off_a) (void)!::ftruncate(fa, (off_t)off_a);
        if (alt_map) {   // (what write_alt has not unmapped yet: the page of the file's end and the unused rest of the estimate)
            if (alt_unmapped < alt_map_len) ::munmap(alt_map + alt_unmapped, (size_t)(alt_map_len - alt_unmapped));
            alt_map = nullptr;
        }
    };
    // a hand-over's _alt text: into the job's mapping where its pages are reserved, else as every other text
    auto write_alt = [&](const std::vector<sd::TextBuf>& parts) -> bool {
        std::vector<int64_t> at(parts.size() + 1, off_a);
        for (size_t i = 0; i < parts.size(); ++i) at[i + 1] = at[i] + (int64_t)parts[i].size();
        if (alt_map && at[parts.size()] <= pre_done.load(std::memory_order_acquire)) {
            sd::parallel_for((int64_t)parts.size(), p->threads, 1, [&](int64_t i) {
                const sd::TextBuf& q = parts[(size_t)i];
                if (q.size()) std::memcpy(alt_map + at[(size_t)i], q.data(), q.size());
            });
            off_a = at[parts.size()];
            // The pages behind the text just written leave the mapping at once, on this (the writer's) thread: taking all
            // 70 000 page-table entries of a 280-MB file down at the end of the job was 10-13 ms on the job's critical
            // path -- or, from a detached thread, on the mmap lock of whatever the process did next.
            const int64_t pg = (int64_t)::sysconf(_SC_PAGESIZE);
            const int64_t upto = off_a / pg * pg;
            if (upto > alt_unmapped) {
                ::munmap(alt_map + alt_unmapped, (size_t)(upto - alt_unmapped));
                alt_unmapped = upto;
            }
            return true;
        }
        return sd::write_parts(fa, off_a, parts, p->threads);
    };
    RowJob job;
    job.n_reads = (int32_t)reads.size();
    job.threads = p->threads;
    build_chunk_table(reads, p, job.table, job.nch);
    if (info) info[3] = (int64_t)job.table.size();
    lap("chunk table");
    job.row_off = static_cast<int64_t*>(std::calloc(reads.size() + 1, sizeof(int64_t)));
    if (!job.row_off) {
        end_prealloc();
        close_all();
        set_err(errbuf, errlen, "out of host memory");
        return SD_ERR_INTERNAL;
    }
    // identities of the final TSV in-stream, behind every batch's compaction (sd_ident.hip); template sets the kernel
    // does not take (and SD_IDENT_STREAM=0, developer A/B) leave them to the post-processing as in round 2
    bool stream_ident;
    std::string pkey;
    {
        sd_params pe = *p;
        apply_env_overrides(pe);
        stream_ident = !(pe.reserved[1] & SD_FLAG_NO_STREAM_IDENT);
        pkey = pipe_cache_key(pe, second_best ? '2' : '1', ts.mseq, ts.mlen);   // (host threads do not shape an engine)
    }
    std::unique_ptr<Pipeline> pipe_h = getenv("SD_PIPE_CACHE_OFF") ? nullptr : pipe_cache_take(pkey);
    const bool reused = pipe_h != nullptr;
    if (!pipe_h) pipe_h.reset(new Pipeline);
    Pipeline& pipe = *pipe_h;
    pipe.restart_idle = true;
    pipe.on_engine = [&](sd_engine* e) {
        if (stream_ident && !engine_set_identity(e, pp.interleaved_seqs(), pp.own_interleaved(), second_best != 0)) stream_ident = false;
    };
    if (reused) {
        pipe.begin_job(p, ts.mseq.data(), ts.mlen.data(), (int32_t)ts.mseq.size());
        stream_ident = stream_ident && pipe.ident_ok;
    } else {
        rc = pipe.create(p, ts.mseq.data(), ts.mlen.data(), (int32_t)ts.mseq.size());
        if (rc) err = pipe.eb;
    }
    lap(reused ? "pipeline from the cache" : "engine (HIP runtime start, layout plan, tables, identity masks)");
    if (stream_ident) job.per = second_best ? (int)pp.interleaved_seqs().size() : 1;
    std::vector<std::pair<size_t, size_t>> batches;
    // --second-best makes the host side of a batch (2T identities' worth of text per row) as long as its kernels.  Round 3
    // cut a job that fits ONE batch in up to four, so that the text of a part is written while the next is on the
    // device -- four under-filled fill launches (C4: 47.6 instead of 22.6 ms of fill).  Now the DP of a batch is one
    // launch and its IDENTITIES run in slices of whole reads (sd_engine::slice_end): the host fetches, assembles and
    // formats slice s while the device computes slice s + 1.
    int min_batches = 1;
    const bool slice_ident = second_best && rc == SD_OK && !getenv("SD_IDENT_SLICES_OFF");
    if (second_best && rc == SD_OK && !slice_ident) {
        const size_t nc = job.table.size();
        min_batches = nc >= 2048 ? 4 : nc >= 1024 ? 2 : 1;   // C4 shape, 2 560 chunks: 170 / 159 / 149 / 140 / 134+ ms for 1 / 2 / 3 / 4 / 5+
    }
    // A process's first job pays for every byte it allocates: the driver scrubs memory another process released before
    // it hands it out -- the 17 GB a 50-Mbp job takes as ONE batch cost 0.2-1.2 s, more than the job (0.3 s).  Such a
    // job is cut into eight batches (two run side by side, stream mode 2), so that its buffers are an eighth as large;
    // a pipeline that comes from the cache has its buffers, and a job of many batches allocates full-size ones once.
    if (!reused && rc == SD_OK && !slice_ident) {
        const size_t nc = job.table.size();
        min_batches = std::max(min_batches, nc >= 4096 ? 8 : nc >= 1024 ? 4 : 1);
    }
    if (const char* ev = getenv("SD_MIN_BATCHES")) min_batches = std::max(1, atoi(ev));   // developer A/B
    if (rc == SD_OK) {
        int64_t budget = pipe.row_budget();
        if (!reused) {
            int64_t rows = 0;
            for (const CRef& c : job.table) rows += c.len;
            if (rows > budget) budget = fresh_row_budget(budget, rows);   // many batches: smaller ones, smaller engines
        }
        plan_batches(job.table, 0, job.table.size(), budget, min_batches, batches);
    }
    lap("batch plan");
    const double t_setup = now_s() - t_begin;
    if (progress) std::fprintf(stderr, "Prepared reads\n");   // main.cpp:82
    // The rows of a batch are assembled on the driver thread (they come out of the engine's pinned buffer, which
    // the next load reuses) and handed to a second host thread that turns them into the three texts and writes
    // them, while the driver packs and enqueues the next batch.  At most two batches wait in the hand-over.
    // identities of the rows: the batch's pinned arrays (taken from the pipeline, given back to the pool when the text is
    // written), where each row's words are (src), and the words of carried rows by value (xid / xidh)
    struct Work {
        size_t r0, r1; sd_rec* rows; std::vector<int64_t> off;
        Pipeline::IdentOut ident; int64_t* src; std::vector<uint32_t> xid, xidh; bool have_ident;
    };
    std::mutex wq_m;
    std::condition_variable wq_cv;
    std::deque<Work> wq;
    bool wq_done = false;
    std::atomic<int> sink_rc{SD_OK};
    std::string sink_err;     // written under wq_m by whichever thread fails first (driver or sink thread)
    auto sink_fail = [&](int code, const std::string& msg) {
        std::lock_guard<std::mutex> lk(wq_m);
        if (sink_rc.load() == SD_OK) { sink_err = msg; sink_rc.store(code); }
    };
    double t_fmt = 0, t_post = 0, t_io = 0;
    // The text of a hand-over goes to a third thread that copies it into the files (the page-cache copy of a --second-best
    // job's _alt rows -- 280 MB at C4 -- takes twice as long as formatting them): formatting hand-over s + 1 and writing
    // hand-over s run side by side.  Text buffers circulate between the two threads (a fresh 35-MB vector is page faults).
    // Text buffers circulate between the two threads and stay with the process between jobs (g_textpool: a fresh 35-MB
    // vector is page faults, and giving 300 MB back to the kernel at the end of every job was 16 ms).
    using WriteJob = TextJob;
    std::mutex io_m;
    std::condition_variable io_cv;
    std::deque<WriteJob> io_q;
    bool io_done = false;
    auto io_loop = [&]() {
        sd::HostPool::lane() = 2;
        for (;;) {
            WriteJob j;
            {
                std::unique_lock<std::mutex> lk(io_m);
                io_cv.wait(lk, [&] { return io_done || !io_q.empty(); });
                if (io_q.empty()) return;
                j = std::move(io_q.front());
                io_q.pop_front();
            }
            io_cv.notify_all();
            const double t0 = now_s();
            const int64_t a0 = off_a;
            if (sink_rc.load() == SD_OK &&
                (!sd::write_parts(fr, off_r, j.raw, p->threads) || !sd::write_parts(ff, off_f, j.fin, p->threads) ||
                 !write_alt(j.alt)))
                sink_fail(SD_ERR_IO, std::string("short write to ") + raw_tsv_out);
            t_io += now_s() - t0;
            if (timing)
                std::fprintf(stderr, "[sd timing] write of a hand-over: %.1f MB of _alt rows in %.1f ms, at %.1f ms into the job\n",
                             (double)(off_a - a0) / 1e6, (now_s() - t0) * 1e3, (now_s() - t_begin) * 1e3);
            g_textpool.give(std::move(j));
        }
    };
    auto sink_loop = [&]() {
        sd::HostPool::lane() = 1;   // this thread's parallel loops run on the second pool, beside the driver's
        std::vector<sd::PostRead> preads;
        for (;;) {
            Work w;
            {
                std::unique_lock<std::mutex> lk(wq_m);
                wq_cv.wait(lk, [&] { return wq_done || !wq.empty(); });
                if (wq.empty()) return;
                w = std::move(wq.front());
                wq.pop_front();
            }
            wq_cv.notify_all();
            if (sink_rc.load() == SD_OK) {
                if (progress) {   // main.cpp:115, one line per read, written per hand-over
                    std::string pl;
                    const size_t n_all = reads.size();
                    for (size_t r = w.r0; r < w.r1; ++r) {
                        sd::put_int(pl, (int64_t)((r + 1) * 100 / n_all));
                        pl.append("%: Aligned ");
                        pl.append(reads[r].name, reads[r].name_len);
                        pl.push_back('\n');
                    }
                    (void)std::fwrite(pl.data(), 1, pl.size(), stderr);
                }
                double t0 = now_s();
                // raw TSV (SaveBatch, main.cpp:272-285): slices of <= 32 k rows, so that a chromosome-sized read is
                // formatted by all threads; a slice needs the end of the row before it
                struct Slice { size_t r; int64_t a, b; };
                std::vector<Slice> slices;
                const int64_t* off = w.off.data();   // off[r - r0] .. : rows of read r
                for (size_t r = w.r0; r < w.r1; ++r)
                    for (int64_t a = off[r - w.r0]; a < off[r - w.r0 + 1]; a += 32768)
                        slices.push_back(Slice{r, a, std::min<int64_t>(off[r - w.r0 + 1], a + 32768)});
                WriteJob wj = g_textpool.take();
                std::vector<std::string>& parts = wj.raw;
                std::vector<std::string>& fin_parts = wj.fin;
                std::vector<sd::TextBuf>& alt_parts = wj.alt;
                parts.resize(slices.size());
                for (std::string& q : parts) q.clear();
                sd::parallel_for((int64_t)slices.size(), p->threads, 1, [&](int64_t x) {
                    const Slice& sl = slices[(size_t)x];
                    sd::format_rows(parts[(size_t)x], reads[sl.r].name, reads[sl.r].name_len, ts.tnames, w.rows + sl.a,
                                    (size_t)(sl.b - sl.a), sl.a > off[sl.r - w.r0] ? w.rows[sl.a - 1].end : 0);
                });
                if (records_out)
                    for (size_t r = w.r0; r < w.r1; ++r)
                        rec_w.add_read(reads[r].name, reads[r].name_len, reads[r].len, w.rows + off[r - w.r0], off[r - w.r0 + 1] - off[r - w.r0]);
                t_fmt += now_s() - t0;
                t0 = now_s();
                preads.clear();
                for (size_t r = w.r0; r < w.r1; ++r)
                    preads.push_back(sd::PostRead{reads[r].name, reads[r].name_len, reads[r].seq, reads[r].len});
                std::string e2;
                sd::IdentRef iref;
                if (w.have_ident)
                    iref = sd::IdentRef{w.ident.id, second_best ? w.ident.idh : nullptr, w.src, w.xid.data(), w.xidh.data()};
                const int r2 = pp.process_parts(preads.data(), preads.size(), w.rows, off, fin_parts, alt_parts, e2,
                                                w.have_ident ? &iref : nullptr);
                t_post += now_s() - t0;
                if (r2) {
                    sink_fail(r2, e2);
                } else {
                    std::unique_lock<std::mutex> lk(io_m);
                    io_cv.wait(lk, [&] { return io_q.size() < 2; });
                    io_q.push_back(std::move(wj));
                    lk.unlock();
                    io_cv.notify_all();
                }
            }
            std::free(w.rows);
            std::free(w.src);
            if (!w.ident.own_id) {   // (blocks of a slice go back when the last slice lets go of them)
                g_pinpool.give(w.ident.id, w.ident.id_bytes);
                g_pinpool.give(w.ident.idh, w.ident.idh_bytes);
            }
        }
    };
    std::thread sink_thread(sink_loop);
    std::thread io_thread(io_loop);
    auto sink = [&](size_t c0, size_t c1, const sd_rec* recs, const int64_t* roff) {
        if (sink_rc.load()) return;
        const size_t r0 = job.next_read;
        job.n_rows = 0;
        job.row_off[r0] = 0;
        job.bid = pipe.cur_ident.id;
        job.bidh = pipe.cur_ident.idh;
        job.add(c0, c1, recs, roff);
        if (job.oom) { sink_fail(SD_ERR_INTERNAL, "out of host memory"); return; }
        const size_t r1 = job.next_read;
        if (r1 == r0) return;
        Work w;
        w.r0 = r0;
        w.r1 = r1;
        w.rows = job.rows;
        w.off.assign(job.row_off + r0, job.row_off + r1 + 1);
        // identities that came with the batches of these rows; a batch without them (more records than the outputs
        // had room for) sends the whole hand-over through the text-based identities
        w.have_ident = job.per && job.ident_ok;
        w.src = job.rsrc;
        job.rsrc = nullptr;
        w.xid.swap(job.xid);
        w.xidh.swap(job.xidh);
        job.xid.clear();
        job.xidh.clear();
        w.ident = Pipeline::IdentOut{};
        if (w.have_ident && job.bid) w.ident = pipe.take_ident();   // the rows point into the batch's pinned arrays
        job.ident_ok = job.carry.empty() || job.bid != nullptr;
        job.rows = nullptr;       // the next batch assembles into a fresh (or recycled) buffer
        job.cap_rows = 0;
        job.n_rows = 0;
        std::unique_lock<std::mutex> lk(wq_m);
        wq_cv.wait(lk, [&] { return wq.size() < 2; });
        wq.push_back(std::move(w));
        lk.unlock();
        wq_cv.notify_all();
    };
    std::vector<const char*> cptr;
    std::vector<int32_t> clen;
    for (size_t b = 0; b < batches.size() && rc == SD_OK && sink_rc.load() == SD_OK; ++b) {
        const size_t c0 = batches[b].first, c1 = batches[b].second;
        cptr.clear();
        clen.clear();
        for (size_t c = c0; c < c1; ++c) {
            cptr.push_back(reads[(size_t)job.table[c].read].seq + job.table[c].off);
            clen.push_back(job.table[c].len);
        }
        std::vector<int> slice_end;
        if (slice_ident && stream_ident) {
            // up to eight slices of at least 256 chunks, each ending with a read (a read that ends in a later slice would
            // only be carried; the last slice ends the batch)
            const size_t nb = c1 - c0;
            int n_sl = (int)std::max<size_t>(1, std::min<size_t>(8, nb / 256));
            if (const char* ev = getenv("SD_IDENT_SLICES")) n_sl = std::max(1, std::min(64, atoi(ev)));   // developer A/B
            size_t at = 0;
            for (int sl = 0; sl < n_sl && at < nb; ++sl) {
                size_t want = sl + 1 == n_sl ? nb : std::max(at + 1, nb * (size_t)(sl + 1) / (size_t)n_sl);
                while (want < nb && job.table[c0 + want].read == job.table[c0 + want - 1].read) ++want;
                slice_end.push_back((int)want);
                at = want;
            }
            if (slice_end.empty() || slice_end.back() != (int)nb) slice_end.push_back((int)nb);
        }
        rc = pipe.push(cptr, clen, [&sink, c0](const sd_rec* r, const int64_t* ro, size_t first, size_t n) { sink(c0 + first, c0 + first + n, r, ro); },
                       slice_end);
        if (rc) err = pipe.eb;
    }
    const int rc2 = pipe.drain();
    if (rc == SD_OK && rc2) { rc = rc2; err = pipe.eb; }
    {
        std::lock_guard<std::mutex> lk(wq_m);
        wq_done = true;
    }
    wq_cv.notify_all();
    sink_thread.join();
    {
        std::lock_guard<std::mutex> lk(io_m);
        io_done = true;
    }
    io_cv.notify_all();
    io_thread.join();
    if (rc == SD_OK && sink_rc.load()) { rc = sink_rc.load(); err = sink_err; }
    end_prealloc();
    if (!close_all() && rc == SD_OK) { rc = SD_ERR_IO; err = std::string("short write to ") + raw_tsv_out; }
    if (records_out && rc == SD_OK) rc = rec_w.close(err, records_out);
    if (timing)
        std::fprintf(stderr, "[sd timing] %zu batches: pack+enqueue %.1f ms, wait %.1f ms, raw text %.1f ms, post-processing %.1f ms, "
                     "file writes %.1f ms, total %.1f ms\n", batches.size(), pipe.pack_s * 1e3, pipe.wait_s * 1e3, t_fmt * 1e3,
                     t_post * 1e3, t_io * 1e3, (now_s() - t_begin) * 1e3);
    if (timing)
        std::fprintf(stderr, "[sd timing] of which device / pinned allocations (hipMalloc, hipHostMalloc): %.1f ms\n", (double)g_alloc_ns.load() / 1e6);
    if (timing)
        std::fprintf(stderr, "[sd timing] post-processing: segments %.1f ms, identities %.1f ms, text %.1f ms, concatenation %.1f ms\n",
                     pp.t_prepare * 1e3, pp.t_identity * 1e3, pp.t_format * 1e3, pp.t_concat * 1e3);
    {
        std::lock_guard<std::mutex> lk(g_last_m);
        const double v[24] = {pipe.fill_ms, pipe.trace_ms, pipe.compact_ms, pipe.ident_ms, (double)pipe.ident_pairs,
                              (double)pipe.batches, (double)pipe.rows, pipe.pack_s * 1e3, pipe.wait_s * 1e3, t_fmt * 1e3,
                              t_post * 1e3, t_io * 1e3, pp.t_identity * 1e3, pp.t_format * 1e3, (now_s() - t_begin) * 1e3,
                              (double)g_alloc_ns.load() / 1e6, t_setup * 1e3, pipe.sink_s * 1e3, 0, 0, 0, 0, 0, 0};
        std::memcpy(g_last_run, v, sizeof v);
    }
    if (timing) {
        double nw[4];
        sd::nw_stage_seconds(nw);
        std::fprintf(stderr, "[sd timing] identities on the device: preparation + staging %.1f ms, uploads %.1f ms, launch %.1f ms, "
                     "kernel + downloads %.1f ms\n", nw[0] * 1e3, nw[1] * 1e3, nw[2] * 1e3, nw[3] * 1e3);
    }
    pipe.ident_ok = stream_ident;
    pipe.on_engine = nullptr;   // (it refers to this call's locals)
    if (rc == SD_OK && !getenv("SD_PIPE_CACHE_OFF")) pipe_cache_give(pkey, std::move(pipe_h));
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    return SD_OK;
}

int sd_run_files_records(const char* reads_fa, const char* monomers_fa, const sd_params* p, const char* raw_tsv_out,
                         const char* final_tsv_out, const char* alt_tsv_out, const char* records_out, int32_t min_identity,
                         int32_t second_best, const double* lr_coef, char* errbuf, size_t errlen) {
    return run_files_impl(reads_fa, monomers_fa, p, 0, 1, raw_tsv_out, final_tsv_out, alt_tsv_out, min_identity, second_best,
                          lr_coef, nullptr, errbuf, errlen, records_out);
}

void sd_last_run_stats(double out[24]) {
    std::lock_guard<std::mutex> lk(g_last_m);
    std::memcpy(out, g_last_run, sizeof g_last_run);
}

int sd_run_files(const char* reads_fa, const char* monomers_fa, const sd_params* p, const char* raw_tsv_out,
                 const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity, int32_t second_best,
                 const double* lr_coef, char* errbuf, size_t errlen) {
    return run_files_impl(reads_fa, monomers_fa, p, 0, 1, raw_tsv_out, final_tsv_out, alt_tsv_out, min_identity, second_best,
                          lr_coef, nullptr, errbuf, errlen);
}

int sd_run_files_range(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank, int32_t world,
                       const char* raw_tsv_out, const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity,
                       int32_t second_best, const double* lr_coef, int64_t* info, char* errbuf, size_t errlen) {
    return run_files_impl(reads_fa, monomers_fa, p, rank, world, raw_tsv_out, final_tsv_out, alt_tsv_out, min_identity,
                          second_best, lr_coef, info, errbuf, errlen);
}

// -------------------------------------------------------------------------------------------
// host-only helpers (CPU tests)
// -------------------------------------------------------------------------------------------
int32_t sd_chunk_plan(int64_t read_len, int32_t part_size, int32_t overlap, int64_t* off,
                      int32_t* len, int32_t cap) {
    int32_t k = 0;
    return sd::chunk_plan(read_len, part_size, overlap, [&](int64_t o, int32_t l) {
        if (k < cap) { if (off) off[k] = o; if (len) len[k] = l; }
        ++k;
    });
}

int32_t sd_seam_merge(sd_rec* recs, int32_t n) {
    std::vector<sd_rec> v(recs, recs + (n > 0 ? n : 0));
    sd::seam_merge(v);
    if (!v.empty()) std::memcpy(recs, v.data(), sizeof(sd_rec) * v.size());
    return (int32_t)v.size();
}

int sd_format_rows(const char* read_name, const char* const* tmpl_names, const sd_rec* rows,
                   int32_t n_rows, char** txt, size_t* txt_len) {
    if (!txt || !txt_len) return SD_ERR_PARAM;
    int maxt = -1;
    for (int32_t i = 0; i < n_rows; ++i) maxt = std::max(maxt, (int)rows[i].tmpl);
    std::vector<std::string> tn;
    for (int j = 0; j <= maxt; ++j) tn.emplace_back(tmpl_names[j]);
    std::string o;
    sd::format_rows(o, read_name, std::strlen(read_name), tn, rows, (size_t)std::max(n_rows, 0));
    char* c = static_cast<char*>(std::malloc(o.size() + 1));
    std::memcpy(c, o.data(), o.size());
    c[o.size()] = 0;
    *txt = c;
    *txt_len = o.size();
    return SD_OK;
}

int sd_fasta_load(const char* path, sd_fasta* out, char* errbuf, size_t errlen) {
    if (!out || !path) return SD_ERR_PARAM;
    std::memset(out, 0, sizeof *out);
    sd::FastaFile ff;
    std::string err;
    const int threads = std::max(1, std::min(32, (int)std::thread::hardware_concurrency()));
    int rc = ff.open(path, threads, err);
    if (rc == SD_OK) rc = ff.validate(0, ff.recs.size(), threads, err);
    if (rc) { set_err(errbuf, errlen, err); return rc; }
    const size_t n = ff.recs.size();
    out->n = (int32_t)n;
    out->has_n = ff.has_n ? 1 : 0;
    out->names = static_cast<char**>(std::malloc(sizeof(char*) * std::max<size_t>(n, 1)));
    out->seqs = static_cast<char**>(std::malloc(sizeof(char*) * std::max<size_t>(n, 1)));
    out->lens = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * std::max<size_t>(n, 1)));
    for (size_t i = 0; i < n; ++i) {
        const sd::FastaFile::Rec& r = ff.recs[i];
        out->names[i] = static_cast<char*>(std::malloc(r.name_len + 1));
        std::memcpy(out->names[i], r.name, r.name_len);
        out->names[i][r.name_len] = 0;
        out->seqs[i] = static_cast<char*>(std::malloc((size_t)r.len + 1));
        std::memcpy(out->seqs[i], r.seq, (size_t)r.len);
        out->seqs[i][r.len] = 0;
        out->lens[i] = r.len;
    }
    return SD_OK;
}

void sd_fasta_free(sd_fasta* f) {
    if (!f) return;
    for (int32_t i = 0; i < f->n; ++i) { std::free(f->names[i]); std::free(f->seqs[i]); }
    std::free(f->names);
    std::free(f->seqs);
    std::free(f->lens);
    std::memset(f, 0, sizeof *f);
}

}  // extern "C"
