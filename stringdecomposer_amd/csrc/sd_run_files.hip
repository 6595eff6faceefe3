// sd_run_files.hip -- sd_run_files*: FASTA files on disk -> final_decomposition{_raw,,_alt}.tsv in one native call
// (main.py:186-197 run() + :168-184 convert_tsv), streamed through the device batch by batch.
// Split from sd_engine.hip in round 6; the C-ABI is unchanged.
#include <sys/mman.h>
#include <sys/vfs.h>

#include "sd_pipeline.hpp"

extern "C" {

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
extern "C++" void text_pool_clear() { g_textpool.clear(); }

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
                    // (page tables first, then the range is handed to the writer: write_alt unmaps what it has written, and
                    // an madvise still walking a range that has left the mapping could meet somebody else's pages there)
#ifdef MADV_POPULATE_WRITE
                    if (alt_map) (void)::madvise(alt_map + at, (size_t)n, MADV_POPULATE_WRITE);
#endif
                    pre_done.store(at + n, std::memory_order_release);
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
                              (double)g_alloc_ns.load() / 1e6, t_setup * 1e3, pipe.sink_s * 1e3, (double)pipe.homo_pairs, (double)pipe.homo_full_pairs, 0, 0, 0, 0};
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

}  // extern "C"
