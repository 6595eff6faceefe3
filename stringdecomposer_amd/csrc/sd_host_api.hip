// sd_host_api.hip -- host only (no device code): the C-ABI entry points that never touch the GPU.
//   sd_write_records / sd_read_records / sd_records_to_raw_tsv   the binary record stream as a format of its own (sd_records.hpp)
//   sd_chunk_table_size, sd_chunk_plan, sd_seam_merge, sd_format_rows, sd_fasta_load / _free   (main.cpp:67-81, 287-302, 272-285, 314-346)
//   sd_host_stage_rates, sd_write_parts_selftest                  measurements / self-tests of the host stages
// Split from sd_engine.hip in round 6; the C-ABI is unchanged.
#include "sd_engine_int.hpp"

extern "C" {

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


int64_t sd_chunk_table_size(const int64_t* read_lens, int32_t n_reads, int32_t part_size, int32_t overlap) {
    if (!read_lens || n_reads < 0 || part_size <= 0 || overlap < 0) return -1;
    int64_t n = 0;
    for (int32_t r = 0; r < n_reads; ++r) n += sd::chunk_plan(read_lens[r], part_size, overlap, [](int64_t, int32_t) {});
    return n;
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
