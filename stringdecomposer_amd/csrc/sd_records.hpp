// sd_records.hpp -- the on-disk binary record stream ("optional binary record stream to skip
// text round-trips").  One file holds what <out>_raw.tsv holds -- the rows of SaveBatch (main.cpp:272-285) per read,
// in read order -- as 16-byte sd_rec records instead of text; the raw TSV is a pure function of it
// (sd_records_to_raw_tsv), so a downstream tool can take the decomposition without parsing 60 bytes of text per row.
//
// Layout (all integers little-endian; every section starts on a multiple of 8 bytes):
//   header      char[8] "SDRECS1\n" | u32 header_bytes (offset of the first read block) | u32 flags (0)
//               | i32 ins, del, mismatch, match, part_size, overlap, ed_thr | u32 n_templates
//               | per template: u32 len, bytes   (column 2 of the raw TSV: monomer names, then the names + "'",
//                 main.cpp:364-371) | zero padding
//   read block  u32 name_len, u32 0 | i64 read_len (-1: unknown) | i64 n_rows | name bytes, zero padding
//               | sd_rec rows[n_rows] (i32 tmpl, start, end, score: columns 2-5; columns 6-7 are derived)
//   trailer     u32 0xFFFFFFFF, u32 0 | i64 n_reads | i64 n_rows_total
// A writer appends read blocks as reads complete (sd_run_files streams them batch by batch); a file without the
// trailer is a truncated one and the reader says so.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/sd_hip.h"

namespace sd {

static const char kRecMagic[8] = {'S', 'D', 'R', 'E', 'C', 'S', '1', '\n'};

struct RecordsWriter {
    int fd = -1;
    std::string buf;
    int64_t n_reads = 0, n_rows = 0;
    int32_t n_tmpl = 0;
    bool failed = false;

    RecordsWriter() = default;
    RecordsWriter(const RecordsWriter&) = delete;
    RecordsWriter& operator=(const RecordsWriter&) = delete;
    ~RecordsWriter() { if (fd >= 0) ::close(fd); }

    template <class T> void put(T v) { buf.append(reinterpret_cast<const char*>(&v), sizeof v); }
    void pad8() { while (buf.size() & 7) buf.push_back('\0'); }
    bool flush() {
        size_t done = 0;
        while (done < buf.size() && !failed) {
            const ssize_t w = ::write(fd, buf.data() + done, buf.size() - done);
            if (w <= 0) failed = true; else done += (size_t)w;
        }
        buf.clear();
        return !failed;
    }
    int open(const char* path, const sd_params& p, const std::vector<std::string>& tnames, std::string& err) {
        fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) { err = std::string("cannot write ") + path; return SD_ERR_IO; }
        n_tmpl = (int32_t)tnames.size();
        buf.append(kRecMagic, 8);
        put<uint32_t>(0);   // header_bytes, patched below
        put<uint32_t>(0);
        for (int32_t v : {p.ins, p.del, p.mismatch, p.match, p.part_size, p.overlap, p.ed_thr}) put<int32_t>(v);
        put<uint32_t>((uint32_t)tnames.size());
        for (const std::string& t : tnames) { put<uint32_t>((uint32_t)t.size()); buf.append(t); }
        pad8();
        const uint32_t hb = (uint32_t)buf.size();
        std::memcpy(&buf[8], &hb, 4);
        return SD_OK;
    }
    // rows of one read, read-global coordinates, in position order (what SaveBatch prints for it)
    void add_read(const char* name, size_t name_len, int64_t read_len, const sd_rec* rows, int64_t n) {
        put<uint32_t>((uint32_t)name_len);
        put<uint32_t>(0);
        put<int64_t>(read_len);
        put<int64_t>(n);
        buf.append(name, name_len);
        pad8();
        if (n > 0) buf.append(reinterpret_cast<const char*>(rows), sizeof(sd_rec) * (size_t)n);
        ++n_reads;
        n_rows += n;
        if (buf.size() >= ((size_t)4 << 20)) flush();
    }
    int close(std::string& err, const char* path) {
        put<uint32_t>(0xFFFFFFFFu);
        put<uint32_t>(0);
        put<int64_t>(n_reads);
        put<int64_t>(n_rows);
        flush();
        const bool ok = ::close(fd) == 0 && !failed;
        fd = -1;
        if (!ok) { err = std::string("short write to ") + path; return SD_ERR_IO; }
        return SD_OK;
    }
};

// The whole file, parsed and checked (bounds, template indices, the trailer's totals).
struct RecordsFile {
    int32_t score[4] = {0, 0, 0, 0}, part_size = 0, overlap = 0, ed_thr = 0;
    std::vector<std::string> tnames, rnames;
    std::vector<int64_t> read_lens, row_off;   // row_off: n_reads + 1
    std::vector<sd_rec> rows;

    int load(const char* path, std::string& err) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path; return SD_ERR_IO; }
        struct stat sb;
        std::string d;
        if (::fstat(fd, &sb) == 0) d.resize((size_t)sb.st_size);
        size_t got = 0;
        while (got < d.size()) {
            const ssize_t r = ::read(fd, &d[got], d.size() - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        ::close(fd);
        if (got != d.size()) { err = std::string("cannot read ") + path; return SD_ERR_IO; }
        return parse(d.data(), d.size(), path, err);
    }
    int parse(const char* d, size_t N, const char* what, std::string& err) {
        auto bad = [&](const std::string& m) { err = std::string(what) + ": " + m; return SD_ERR_FORMAT; };
        size_t at = 0;
        auto need = [&](size_t k) { return at + k <= N; };
        auto u32 = [&]() { uint32_t v; std::memcpy(&v, d + at, 4); at += 4; return v; };
        auto i32 = [&]() { int32_t v; std::memcpy(&v, d + at, 4); at += 4; return v; };
        auto i64 = [&]() { int64_t v; std::memcpy(&v, d + at, 8); at += 8; return v; };
        if (N < 48 || std::memcmp(d, kRecMagic, 8) != 0) return bad("not a record stream (bad magic)");
        at = 8;
        const uint32_t hb = u32();
        (void)u32();
        for (int k = 0; k < 4; ++k) score[k] = i32();
        part_size = i32(); overlap = i32(); ed_thr = i32();
        const uint32_t nt = u32();
        for (uint32_t t = 0; t < nt; ++t) {
            if (!need(4)) return bad("truncated template table");
            const uint32_t l = u32();
            if (!need(l)) return bad("truncated template table");
            tnames.emplace_back(d + at, l);
            at += l;
        }
        at = (at + 7) & ~(size_t)7;
        if (at != hb || hb > N) return bad("header size does not match the template table");
        row_off.assign(1, 0);
        for (;;) {
            if (!need(8)) return bad("truncated: no trailer (the writer did not finish)");
            const uint32_t nl = u32();
            (void)u32();
            if (nl == 0xFFFFFFFFu) {
                if (!need(16)) return bad("truncated trailer");
                const int64_t nr = i64(), nrow = i64();
                if (nr != (int64_t)rnames.size() || nrow != (int64_t)rows.size()) return bad("trailer totals do not match the read blocks");
                if (at != N) return bad("bytes after the trailer");
                return SD_OK;
            }
            if (!need(16)) return bad("truncated read block");
            const int64_t rl = i64(), n = i64();
            if (n < 0 || !need(nl)) return bad("truncated read block");
            rnames.emplace_back(d + at, nl);
            at = (at + nl + 7) & ~(size_t)7;
            if ((uint64_t)n > (N - std::min(at, N)) / sizeof(sd_rec)) return bad("truncated read block");
            const size_t r0 = rows.size();
            rows.resize(r0 + (size_t)n);
            if (n) std::memcpy(&rows[r0], d + at, sizeof(sd_rec) * (size_t)n);
            at += sizeof(sd_rec) * (size_t)n;
            for (size_t x = r0; x < rows.size(); ++x)
                if (rows[x].tmpl < 0 || (uint32_t)rows[x].tmpl >= nt) return bad("record with a template index outside the template table");
            read_lens.push_back(rl);
            row_off.push_back((int64_t)rows.size());
        }
    }
};

}  // namespace sd
