// sd_range_asm.hip -- host only (no device code): raw TSV text from per-chunk records.
//   sd_assemble_tsv / sd_assemble_files_tsv   all records of a job -> the reference's stdout (main.cpp:104-120, 272-302)
//   sd_range_assemble_*                       one rank's chunk range of a job sharded over processes: the seam merge across
//                                             range boundaries from exchanged 160-byte edges (sd_seam.hpp)
// Split from sd_engine.hip in round 6; the C-ABI is unchanged.
#include "sd_engine_int.hpp"

extern "C" {

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

extern "C++" int range_asm_from_files(sd::FastaFile& rf, sd::FastaFile& mf, const sd_params* p, int64_t lo, int64_t hi,
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


}  // extern "C"
