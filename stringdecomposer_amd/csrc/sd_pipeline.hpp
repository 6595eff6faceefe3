// sd_pipeline.hpp -- internal to libsd_hip.so: the engine structure, the device / pinned buffer pools, the batch pipeline and
// the per-read row assembly, shared by the units that drive the device:
//   sd_engine.hip     engines (create / load / run / fetch), pipeline cache, sd_decompose*, chunk-range calls
//   sd_stream.hip     sd_stream_*: sequences in host memory -> rows in host memory
//   sd_run_files.hip  sd_run_files*: FASTA files -> the three TSV files
// Split from sd_engine.hip in round 6 (it was one 3 900-line unit); the C-ABI is unchanged.
#pragma once

#include <hip/hip_runtime.h>

#include "sd_engine_int.hpp"
#include "sd_convert.hpp"
#include "sd_ident.hpp"
#include "sd_nw.hpp"
#include "sd_device.hpp"
#include "sd_fast.hpp"
#include "sd_kernels.hpp"

namespace sdi {

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
extern DevPool g_pool;                    // (defined in sd_engine.hip)
extern std::atomic<long long> g_alloc_ns;   // time spent in hipMalloc / hipHostMalloc (SD_TIMING report)
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
// AFTER it has destroyed the cached pipelines and their streams.  (Emptying the list at the start of a job, with nothing in
// flight anywhere, was tried and hangs in the same way: what the wait cannot cope with is the NUMBER of queues the process
// holds -- two cached pipelines are a dozen streams -- not work on them.)  Buffers grow by doubling, so the list stays below
// what is in use; a long-lived process that goes through many parameter sets calls sd_release_cache() now and then.
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
inline DeferredFrees& g_deferred_ref() { static DeferredFrees* d = new DeferredFrees; return *d; }
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
extern PinPool g_pinpool;                // (defined in sd_engine.hip)

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

}  // namespace sdi
using namespace sdi;

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
    DevBuf<int> d_icandcnt;        // one counter per identity slice (64)
    DevBuf<uint4> d_ick2;          // checkpoint workspace of the candidate stage (it runs beside the next slice's kernels)
    DevBuf<int> d_ickpos2;
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
    // pruned homopolymer pass of a sliced run: the full alignments of a slice's candidates (a few waves: one wave's latency,
    // 0.4 ms, whatever their number) run on a stream of their own beside the NEXT slice's kernels
    hipStream_t cand_st = nullptr;
    std::vector<hipEvent_t> ev_cand;
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
        for (hipEvent_t e : ev_cand) (void)hipEventDestroy(e);
        if (cand_st) (void)hipStreamDestroy(cand_st);
        g_pinpool.give(h_ident, h_ident_bytes);
        g_pinpool.give(h_identh, h_identh_bytes);
    }

    size_t workspace_bytes() const {
        return d_tmeta.bytes() + d_tend_kd.bytes() + d_tend_j.bytes() + d_ptr.bytes() + d_estate.bytes() + d_grank.bytes() +
               d_ftable.bytes() + d_flane.bytes() + d_fslot.bytes() + d_ftcodes.bytes() + d_fckpt.bytes() +
               d_fckbase.bytes() + d_in.bytes() +
               d_B.bytes() + d_argB.bytes() + d_cnt.bytes() + d_recs.bytes() + d_dense.bytes() +
               d_roff.bytes() + d_recchunk.bytes() + d_ilong.bytes() + d_ick.bytes() + d_ickpos.bytes() +
               d_ident.bytes() + d_identh.bytes() + d_icand.bytes() + d_ick2.bytes() + d_ickpos2.bytes();
    }
};

// ---- engine internals the pipeline drives (sd_engine.hip) ------------------------------------------------------
void apply_env_overrides(sd_params& p);
bool engine_set_identity(sd_engine* e, const std::vector<std::string>& il_seq, const std::vector<int32_t>& own, bool second_best);
int load_chunks_impl(sd_engine* e, const std::vector<const char*>& cptr, const std::vector<int32_t>& clen, hipStream_t st,
                     char* errbuf, size_t errlen);
int engine_run2(sd_engine* e, hipStream_t st, hipStream_t ts, char* errbuf, size_t errlen);
int fetch_begin(sd_engine* e, int64_t& total, char* errbuf, size_t errlen);
int fetch_range(sd_engine* e, int64_t r_lo, int64_t r_hi, uint32_t* id_dst, uint32_t* idh_dst, char* errbuf, size_t errlen);
void engine_grow_ident(sd_engine* e, int64_t total);


// Device pipeline: up to three batches of chunks in flight on three engines (fills alternate between two streams).
// push() packs a batch into the engine's pinned staging buffer, starts its H2D copy and enqueues its
// kernels (all asynchronous); pop() waits for the oldest batch, brings its records into pinned host
// memory and hands them to that batch's sink.  While the device works on batch b the host packs and
// enqueues b+1 and then assembles b; kernels of consecutive batches sit on different streams, so the
// tail of one launch overlaps the head of the next.
// recs of the chunks [first, first + n) of a batch (word 0 = the first record of chunk `first`), their offsets (n + 1,
// relative to recs); a batch arrives in one call (first = 0) or, with identity slices, in one call per slice
using RecSink = std::function<void(const sd_rec*, const int64_t*, size_t, size_t)>;
namespace sdi {

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
    int64_t homo_pairs = 0, homo_full_pairs = 0;   // homopolymer-compressed pairs of the job / those aligned in full (pruned pass)
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
        homo_pairs = homo_full_pairs = 0;
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
            // pruned homopolymer pass: the pairs that were aligned in full (one counter per identity slice)
            if (e->ident_valid && e->ident_mode == 2 && e->ia_homo.cand_list && e->d_icandcnt.p) {
                int cnt[64];
                if (hipMemcpy(cnt, e->d_icandcnt.p, sizeof cnt, hipMemcpyDeviceToHost) == hipSuccess)
                    for (size_t sl = 0; sl < std::max<size_t>(1, e->sliced_run ? e->slice_end.size() : 1) && sl < 64; ++sl) homo_full_pairs += cnt[sl];
                homo_pairs += total * (int64_t)e->iT;
            }
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

// Cuts the chunks [c_lo, c_hi) of a table into device batches of consecutive chunks: at most `budget` rows each (a chunk
// that alone exceeds it is a batch of its own), about `min_batches` or more of them, and of about EQUAL rows -- the last
// batch of a job must not be a small remainder (a launch with less than one chunk per resident wave takes as long as a
// full round).  Round 6: the share is recomputed from what is left after every batch, and the number of batches counts
// with the rows a budget can really hold (a budget is rarely a multiple of the chunk length): rounds 1-5 cut 289 chunks
// of 5.5 kb at a budget of 100 000 rows into sixteen batches of 18 and ONE of 1 (sd_pipeline_logic_selftest).
inline void plan_batches(const std::vector<CRef>& table, size_t c_lo, size_t c_hi, int64_t budget, int min_batches,
                  std::vector<std::pair<size_t, size_t>>& out) {
    out.clear();
    int64_t left = 0, lmax = 1;
    for (size_t c = c_lo; c < c_hi; ++c) { left += table[c].len; lmax = std::max<int64_t>(lmax, table[c].len); }
    budget = std::max<int64_t>(budget, 1);
    const int64_t holds = std::max<int64_t>(1, budget - (lmax - 1));   // rows a batch is sure to take before the next chunk no longer fits
    int64_t nb = std::max<int64_t>(std::max(min_batches, 1), (left + holds - 1) / holds);
    for (size_t c0 = c_lo; c0 < c_hi;) {
        const int64_t target = (left + nb - 1) / nb;
        int64_t rows = 0;
        size_t c1 = c0;
        while (c1 < c_hi && (c1 == c0 || (rows < target && rows + table[c1].len <= budget))) rows += table[c1++].len;
        out.emplace_back(c0, c1);
        left -= rows;
        nb = std::max<int64_t>(std::max<int64_t>(nb - 1, 1), (left + holds - 1) / holds);
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
inline int64_t fresh_row_budget(int64_t budget, int64_t job_rows) {
    int64_t cap = job_rows > 20000000000ll ? budget : job_rows > 3000000000ll ? ((int64_t)28 << 20) : ((int64_t)14 << 20);
    if (const char* ev = getenv("SD_FRESH_ROWS")) { const long long v = atoll(ev); if (v > 0) cap = v; else return budget; }
    if (job_rows <= cap) return std::min(budget, cap);                 // (a job of one batch: the callers' own rules)
    return std::min(budget, std::max<int64_t>(cap / 2, std::min(cap, job_rows / 8)));
}
}  // namespace sdi

// ---- pipelines kept between jobs (sd_engine.hip) ----------------------------------------------------------------
std::string pipe_cache_key(const sd_params& pe, char kind, const std::vector<const char*>& mseq, const std::vector<int32_t>& mlen);
std::unique_ptr<Pipeline> pipe_cache_take(const std::string& key);
void pipe_cache_give(const std::string& key, std::unique_ptr<Pipeline> q);
void pipe_cache_clear();
void text_pool_clear();      // (sd_run_files.hip: the text buffers sd_run_files keeps between jobs)

// -------------------------------------------------------------------------------------------
// streaming form: sequences in host memory -> rows in host memory (AlignReadsSet, main.cpp:67-122,
// without the text), jobs pipelined through the device in sub-batches
// -------------------------------------------------------------------------------------------
namespace sdi {
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
}  // namespace sdi
