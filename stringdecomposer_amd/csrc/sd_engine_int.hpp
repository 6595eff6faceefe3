// sd_engine_int.hpp -- internal to libsd_hip.so: what the translation units behind the C-ABI (include/sd_hip.h) share.
//   sd_engine.hip     engines, batch pipeline, pipeline cache, the job entry points that drive the device
//   sd_range_asm.hip  host only: raw TSV from records (sd_assemble_*), the rank-local assembly of a chunk range (sd_range_assemble_*)
//   sd_host_api.hip   host only: record stream files, FASTA / chunk-plan / seam-merge / formatting helpers, host self-tests
// Types and small stateless helpers only; everything has internal linkage per unit except the two functions declared at
// the end.  (Round 6: sd_engine.hip was one 3 900-line unit; the host-only third of it no longer rebuilds with the engine.)
#pragma once

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sd_hip.h"
#include "sd_host.hpp"
#include "sd_records.hpp"
#include "sd_seam.hpp"

namespace {

inline void set_err(char* buf, size_t len, const std::string& m) {
    if (buf && len) {
        std::snprintf(buf, len, "%s", m.c_str());
    }
}

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline int validate_params(const sd_params* p, std::string& err) {
    if (!p) { err = "null params"; return SD_ERR_PARAM; }
    if (p->part_size <= 0) { err = "part_size must be > 0"; return SD_ERR_PARAM; }
    if (p->overlap < 0) { err = "overlap must be >= 0"; return SD_ERR_PARAM; }
    return SD_OK;
}

struct ReadView {  // borrowed for the duration of the call
    const char* name;
    size_t name_len;
    const char* seq;
    int64_t len;
};
struct CRef { int32_t read; int64_t off; int32_t len; };

// Global chunk table (main.cpp:70-81) of a read set; nch[r] = chunks of read r.
inline void build_chunk_table(const std::vector<ReadView>& reads, const sd_params* p, std::vector<CRef>& table,
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

// Per-read assembly (main.cpp:104-117) of per-chunk records arriving in chunk order: chunk offsets,
// seam merge, raw TSV text.  Reads complete in input order.
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

// the rank-local assembly's handle from two mapped FASTA files (sd_range_asm.hip); sd_decompose_files_range_begin
// (sd_engine.hip) makes it right behind the DP of the share
struct sd_range_asm;
int range_asm_from_files(sd::FastaFile& rf, sd::FastaFile& mf, const sd_params* p, int64_t lo, int64_t hi,
                         sd_rec* recs, int64_t* off, sd_seam_edge* edge, sd_range_asm** hout, std::string& err);
