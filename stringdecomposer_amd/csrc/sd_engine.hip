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
#include "sd_engine_int.hpp"
#include "sd_pipeline.hpp"

namespace sdi {
DevPool g_pool;
std::atomic<long long> g_alloc_ns{0};   // time spent in hipMalloc / hipHostMalloc (SD_TIMING report)
PinPool g_pinpool;
}  // namespace sdi


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

}  // namespace

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

namespace {

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
extern "C++" bool engine_set_identity(sd_engine* e, const std::vector<std::string>& il_seq, const std::vector<int32_t>& own,
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
                e->d_icandcnt.alloc(64);
                a.cand_list = e->d_icand.p;
                a.cand_cnt = e->d_icandcnt.p;
                a.grid_cand = e->n_cu * 3;
                e->d_ick2.alloc((size_t)a.grid_cand * 256 * (size_t)a.cap_short * (size_t)a.K);
                e->d_ickpos2.alloc((size_t)a.grid_cand * 256 * (size_t)a.cap_short);
                a.ck_cand = e->d_ick2.p;
                a.ckpos_cand = e->d_ickpos2.p;
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

extern "C++" int load_chunks_impl(sd_engine* e, const std::vector<const char*>& cptr,
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
extern "C++" int engine_run2(sd_engine* e, hipStream_t st, hipStream_t ts, char* errbuf, size_t errlen) {
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
                    // the candidate stage of slice s on its own stream, beside the kernels of slice s + 1 (per-slice counter,
                    // per-slice region of the list, its own checkpoint workspace)
                    const bool side = e->ident_mode == 2 && e->ia_homo.cand_list && e->slice_end.size() <= 64 && !getenv("SD_IDENT_CAND_INLINE");
                    if (side) {
                        if (!e->cand_st) SD_HIP(hipStreamCreateWithFlags(&e->cand_st, hipStreamNonBlocking));
                        while (e->ev_cand.size() < e->slice_end.size()) {
                            hipEvent_t ev;
                            SD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                            e->ev_cand.push_back(ev);
                        }
                        SD_HIP(hipMemsetAsync(e->d_icandcnt.p, 0, 64 * sizeof(int), ts));
                    }
                    int c_lo = 0;
                    bool on_side = false;
                    for (size_t sl = 0; sl < e->slice_end.size(); ++sl) {
                        const int c_hi = e->slice_end[sl];
                        e->ia_plain.rec_lo = e->ia_homo.rec_lo = e->d_roff.p + c_lo;
                        e->ia_plain.rec_hi = e->ia_homo.rec_hi = e->d_roff.p + c_hi;
                        sd::launch_ident(ts, e->ia_plain);
                        bool rec_on_side = false;
                        if (e->ident_mode == 2) {
                            e->ia_homo.cand_cnt = e->ia_homo.cand_list ? e->d_icandcnt.p + (side ? sl : 0) : nullptr;
                            if (side) {
                                if (sd::launch_ident_pruned_front(ts, e->ia_homo)) {
                                    SD_HIP(hipEventRecord(e->ev_cand[sl], ts));
                                    SD_HIP(hipStreamWaitEvent(e->cand_st, e->ev_cand[sl], 0));
                                    sd::launch_ident_pruned_back(e->cand_st, e->ia_homo);
                                    rec_on_side = on_side = true;
                                }
                            } else {
                                sd::launch_ident_pruned(ts, e->ia_homo);
                            }
                        }
                        SD_HIP(hipEventRecord(e->ev_slice[sl], rec_on_side ? e->cand_st : ts));
                        c_lo = c_hi;
                    }
                    if (on_side) SD_HIP(hipStreamWaitEvent(ts, e->ev_slice.back(), 0));   // what follows on ts follows the last candidates
                } else {
                    e->ia_plain.rec_lo = e->ia_homo.rec_lo = e->ia_plain.rec_hi = e->ia_homo.rec_hi = nullptr;
                    sd::launch_ident(ts, e->ia_plain);
                    if (e->ident_mode == 2) {
                        e->ia_homo.cand_cnt = e->ia_homo.cand_list ? e->d_icandcnt.p : nullptr;
                        sd::launch_ident_pruned(ts, e->ia_homo);
                    }
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
extern "C++" int fetch_begin(sd_engine* e, int64_t& total, char* errbuf, size_t errlen) {
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
extern "C++" int fetch_range(sd_engine* e, int64_t r_lo, int64_t r_hi, uint32_t* id_dst, uint32_t* idh_dst, char* errbuf, size_t errlen) {
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
extern "C++" void engine_grow_ident(sd_engine* e, int64_t total) {
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
}  // namespace
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

// -------------------------------------------------------------------------------------------
// chunk-range form: multi-GPU sharding of one job, one process per GPU
// -------------------------------------------------------------------------------------------
// Host only (CPU test): the two pieces of pipeline logic that decide WHICH engines a job meets and HOW it is cut, checked
// against their contracts without a device --
//   pipe_cache_key: two jobs share cached engines exactly when every parameter that shapes an engine and the monomer set
//     agree; the host-thread count does not shape one; ("AC", "G") and ("A", "CG") are different sets;
//   plan_batches / fresh_row_budget: consecutive, complete, in order; no batch above the budget unless one chunk alone
//     is; at least half of min_batches when there are that many chunks (a share carries a slack of one chunk); no small
//     remainder (the smallest batch of a job of equal chunks, 16 or more per batch, holds at least half the rows of the largest).
// Returns SD_OK or SD_ERR_INTERNAL with the broken property in errbuf.
int sd_pipeline_logic_selftest(char* errbuf, size_t errlen) {
    auto fail = [&](const std::string& m) { set_err(errbuf, errlen, m); return SD_ERR_INTERNAL; };
    sd_params a;
    sd_params_default(&a);
    const char* m1[] = {"ACGT", "GG"};
    const std::vector<const char*> ms(m1, m1 + 2);
    const std::vector<int32_t> ml = {4, 2};
    const std::string k0 = pipe_cache_key(a, '1', ms, ml);
    {
        sd_params b = a;
        b.threads = 17;
        if (pipe_cache_key(b, '1', ms, ml) != k0) return fail("cache key depends on the host-thread count");
    }
    int32_t* fields[] = {&a.ins, &a.del, &a.mismatch, &a.match, &a.part_size, &a.overlap, &a.ed_thr, &a.device, &a.kernel,
                         &a.max_batch_rows, &a.reserved[0], &a.reserved[1], &a.reserved[2], &a.reserved[3], &a.reserved[4]};
    for (size_t f = 0; f < sizeof fields / sizeof fields[0]; ++f) {
        const int32_t keep = *fields[f];
        *fields[f] = keep + 1;
        const std::string k = pipe_cache_key(a, '1', ms, ml);
        *fields[f] = keep;
        if (k == k0) return fail("cache key ignores parameter field " + std::to_string(f));
    }
    if (pipe_cache_key(a, '2', ms, ml) == k0) return fail("cache key ignores the job kind");
    {
        const char* m2[] = {"ACGTG", "G"};
        if (pipe_cache_key(a, '1', std::vector<const char*>(m2, m2 + 2), {5, 1}) == k0) return fail("cache key: monomer boundaries");
        const char* m3[] = {"GG", "ACGT"};
        if (pipe_cache_key(a, '1', std::vector<const char*>(m3, m3 + 2), {2, 4}) == k0) return fail("cache key: monomer order");
    }
    if (pipe_cache_take(std::string("no such key\x01")) != nullptr) return fail("cache returned a pipeline for an unknown key");
    // batch plans over random chunk tables
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](uint64_t n) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint64_t)(rng % n); };
    for (int trial = 0; trial < 400; ++trial) {
        const size_t n = 1 + (size_t)rnd(300);
        const bool equal = rnd(3) == 0;
        std::vector<CRef> table(n);
        for (size_t c = 0; c < n; ++c) table[c] = CRef{(int32_t)c, 0, equal ? 5500 : (int32_t)(1 + rnd(5500))};
        const size_t lo = (size_t)rnd(n), hi = lo + 1 + (size_t)rnd(n - lo);
        const int64_t budget = 1 + (int64_t)rnd(200000);
        const int minb = 1 + (int)rnd(9);
        std::vector<std::pair<size_t, size_t>> out;
        plan_batches(table, lo, hi, budget, minb, out);
        size_t at = lo;
        int64_t small = INT64_MAX, big = 0;
        for (const auto& b : out) {
            if (b.first != at || b.second <= b.first) return fail("batches are not consecutive and non-empty");
            int64_t rows = 0;
            for (size_t c = b.first; c < b.second; ++c) rows += table[c].len;
            if (rows > budget && b.second - b.first > 1) return fail("a batch of several chunks exceeds the budget");
            small = std::min(small, rows);
            big = std::max(big, rows);
            at = b.second;
        }
        if (at != hi) return fail("batches do not cover the range");
        // (the shares carry a slack of one chunk, so a batch may take one chunk more than its share: at least half of what was asked for)
        if (equal && budget >= (int64_t)5500 * (int64_t)(hi - lo) && 2 * out.size() < (size_t)std::min<int>(minb, (int)(hi - lo)))
            return fail("far fewer batches than min_batches although the chunks allow them");
        // (with a handful of chunks per batch the one-chunk slack itself is a large fraction: 10 chunks in 4 batches are 3, 3, 3, 1)
        if (equal && out.size() > 1 && hi - lo >= 16 * out.size() && small * 2 < big) return fail("a small remainder batch");
    }
    if (fresh_row_budget(100 << 20, 1000) > (100 << 20)) return fail("fresh_row_budget raised the budget");
    for (int64_t rows : {(int64_t)1 << 20, (int64_t)60 << 20, (int64_t)500 << 20, (int64_t)5000 << 20, (int64_t)30000 << 20}) {
        const int64_t b = fresh_row_budget((int64_t)64 << 20, rows);
        if (b <= 0 || b > ((int64_t)64 << 20)) return fail("fresh_row_budget outside (0, budget]");
    }
    return SD_OK;
}

void sd_release_cache(void) { pipe_cache_clear(); g_pool.release_all(); g_pinpool.release_all(); g_deferred.drain(); text_pool_clear(); }


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


}  // extern "C"
