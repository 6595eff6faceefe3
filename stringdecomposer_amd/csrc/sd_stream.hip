// sd_stream.hip -- sd_stream_*: the streaming form of the job (sequences in host memory -> rows in host memory,
// AlignReadsSet of main.cpp:67-122 without the text), jobs pipelined through the device in sub-batches.
// Split from sd_engine.hip in round 6; the C-ABI is unchanged.
#include "sd_pipeline.hpp"

extern "C" {

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

}  // extern "C"
