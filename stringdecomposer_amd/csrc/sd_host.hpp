// sd_host.hpp -- host side of the hot path: FASTA ingest, reverse complements, chunk table,
// 2-bit packing, per-read assembly (offset + seam merge) and raw-TSV formatting.
//
// Behaviour follows ablab/stringdecomposer v1.1.2 stringdecomposer/src/main.cpp (cited per
// function); the code is written from scratch for this library.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/sd_hip.h"

namespace sd {

struct Seq {
    std::string name;
    std::string seq;
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

// load_fasta (main.cpp:314-346): record name = first whitespace token of the header, sequence
// lines appended verbatim (no upper-casing, '\r' kept), alphabet {A,C,G,T,N} enforced afterwards.
inline int load_fasta(const std::string& path, std::vector<Seq>& out, bool& has_n, std::string& err) {
    out.clear();
    has_n = false;
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open " + path; return SD_ERR_IO; }
    std::string data;
    {
        char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, fp)) > 0) data.append(buf, got);
    }
    std::fclose(fp);
    size_t pos = 0;
    const size_t N = data.size();
    while (pos < N) {
        size_t eol = data.find('\n', pos);
        if (eol == std::string::npos) eol = N;
        const char* ln = data.data() + pos;
        const size_t L = eol - pos;
        if (L > 0 && ln[0] == '>') {
            size_t a = 1;
            while (a < L && is_ws(ln[a])) ++a;
            size_t b = a;
            while (b < L && !is_ws(ln[b])) ++b;
            if (a == b) { err = "FASTA header without a name"; return SD_ERR_FORMAT; }
            out.push_back(Seq{std::string(ln + a, b - a), std::string()});
        } else if (L > 0) {
            if (out.empty()) { err = "FASTA does not start with a header"; return SD_ERR_FORMAT; }
            out.back().seq.append(ln, L);
        }
        pos = eol + 1;
    }
    for (const Seq& s : out) {
        for (char c : s.seq) {
            if (c == 'A' || c == 'C' || c == 'G' || c == 'T') continue;
            if (c == 'N') { has_n = true; continue; }
            err = "ERROR: Sequence " + s.name + " contains undefined symbol (not ACGT): " + c;
            return SD_ERR_SYMBOL;
        }
    }
    return SD_OK;
}

// alphabet check for in-memory sequences (same rule and message as load_fasta)
inline int check_alphabet(const char* name, const char* s, int64_t n, std::string& err) {
    for (int64_t i = 0; i < n; ++i) {
        char c = s[i];
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') continue;
        err = std::string("ERROR: Sequence ") + (name ? name : "?") +
              " contains undefined symbol (not ACGT): " + c;
        return SD_ERR_SYMBOL;
    }
    return SD_OK;
}

// reverse_complement (main.cpp:348-362)
inline bool reverse_complement(const std::string& s, std::string& out) {
    out.resize(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i], r;
        switch (c) {
            case 'A': r = 'T'; break;
            case 'T': r = 'A'; break;
            case 'G': r = 'C'; break;
            case 'C': r = 'G'; break;
            case 'N': r = 'N'; break;
            default: return false;
        }
        out[i] = r;
    }
    return true;
}

// base code used on the device: A,C,G,T = 0..3, N = 4
inline int base_code(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

// chunk plan (main.cpp:70-81): for i = 0, part, 2*part ... < len keep [i, i+min(part+overlap,len-i))
// iff len-i >= overlap or len < overlap.
template <class F>
inline int chunk_plan(int64_t len, int part, int overlap, F&& emit) {
    int cnt = 0;
    for (int64_t i = 0; i < len; i += part) {
        if (len - i >= overlap || len < overlap) {
            int64_t l = len - i;
            if ((int64_t)part + overlap < l) l = (int64_t)part + overlap;
            emit(i, (int32_t)l);
            ++cnt;
        }
    }
    return cnt;
}

// PostProcessing (main.cpp:287-302), literal: after a drop the element behind the dropped run is
// appended without being compared with its own successors.
inline void seam_merge(std::vector<sd_rec>& b) {
    std::vector<sd_rec> res;
    res.reserve(b.size());
    size_t i = 0;
    const size_t N = b.size();
    while (i < N) {
        const size_t lim = i + 7 < N ? i + 7 : N;
        for (size_t j = i + 1; j < lim; ++j) {
            if ((b[i].end - b[j].start) * 2 > (b[j].end - b[j].start)) {
                res.push_back(b[i]);
                i = j + 1;
                break;
            }
        }
        if (i < N) res.push_back(b[i]);
        ++i;
    }
    b.swap(res);
}

// decimal text of an int, appended (std::to_string(int), main.cpp:277-281)
inline void put_int(std::string& o, int64_t v) {
    char buf[24];
    int p = 24;
    bool neg = v < 0;
    uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    do { buf[--p] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) buf[--p] = '-';
    o.append(buf + p, 24 - p);
}

// SaveBatch (main.cpp:272-285).  The reference prints to_string(float identity) == "%f"; the
// identity is an integer-valued float (|v| < 1e6 < 2^24 by the range check in the engine), so
// "%f" is exactly "<int>.000000".
inline void format_rows(std::string& o, const char* read_name, size_t read_name_len,
                        const std::vector<std::string>& tnames, const sd_rec* rows, size_t n,
                        int prev_end = 0) {  // prev_end: end of the row before rows[0] (0 at a read's start)
    for (size_t x = 0; x < n; ++x) {
        const sd_rec& r = rows[x];
        o.append(read_name, read_name_len);
        o.push_back('\t');
        o.append(tnames[(size_t)r.tmpl]);
        o.push_back('\t');
        put_int(o, r.start);
        o.push_back('\t');
        put_int(o, r.end);
        o.push_back('\t');
        put_int(o, r.score);
        o.append(".000000\t", 8);
        put_int(o, (int64_t)r.start - prev_end);
        o.push_back('\t');
        put_int(o, (int64_t)r.end - r.start);
        o.push_back('\n');
        prev_end = r.end;
    }
}

// Minimal fork-join over [0, n): `threads` host threads pull blocks of `grain` indices.
template <class F>
inline void parallel_for(int64_t n, int threads, int64_t grain, F&& body) {
    if (n <= 0) return;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && threads > hw) threads = hw;
    if (threads <= 1 || n <= grain) {
        for (int64_t i = 0; i < n; ++i) body(i);
        return;
    }
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t b = next.fetch_add(grain);
            if (b >= n) break;
            const int64_t e = b + grain < n ? b + grain : n;
            for (int64_t i = b; i < e; ++i) body(i);
        }
    };
    std::vector<std::thread> th;
    const int64_t blocks = (n + grain - 1) / grain;
    const int nt = (int)(blocks < threads ? blocks : threads);
    for (int k = 1; k < nt; ++k) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

// 2-bit packing of one chunk (16 bases per dword); returns true if the chunk contains N
inline bool pack_chunk(const char* s, int32_t l, uint32_t* out) {
    bool has_n = false;
    const int32_t full = l & ~15;
    for (int32_t i = 0; i < full; i += 16) {
        uint32_t w = 0;
        for (int k = 0; k < 16; ++k) {
            const int code = base_code(s[i + k]);
            has_n |= code == 4;
            w |= (uint32_t)(code & 3) << (2 * k);
        }
        out[i >> 4] = w;
    }
    if (full < l) {
        uint32_t w = 0;
        for (int32_t i = full; i < l; ++i) {
            const int code = base_code(s[i]);
            has_n |= code == 4;
            w |= (uint32_t)(code & 3) << (2 * (i & 15));
        }
        out[full >> 4] = w;
    }
    return has_n;
}

}  // namespace sd
