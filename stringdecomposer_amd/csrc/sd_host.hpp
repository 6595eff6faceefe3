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
#include <condition_variable>
#include <functional>
#include <mutex>
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
// appended without being compared with its own successors.  In place: the write position never
// passes the read position (an iteration appends at most b[i] and b[j+1] while i moves to j+2).
inline size_t seam_merge_inplace(sd_rec* b, size_t N) {
    size_t w = 0, i = 0;
    while (i < N) {
        const size_t lim = i + 7 < N ? i + 7 : N;
        for (size_t j = i + 1; j < lim; ++j) {
            if ((b[i].end - b[j].start) * 2 > (b[j].end - b[j].start)) {
                b[w++] = b[i];
                i = j + 1;
                break;
            }
        }
        if (i < N) b[w++] = b[i];
        ++i;
    }
    return w;
}
inline void seam_merge(std::vector<sd_rec>& b) { b.resize(seam_merge_inplace(b.data(), b.size())); }

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

// Process-wide pool of host worker threads.  The packer, the assembler and the formatter run many
// short parallel loops per device batch (a few hundred microseconds each); creating 32-64 threads for
// every loop cost more than the loops, so the threads persist and sleep between loops.  One loop at a
// time (a second caller waits its turn); a loop started from inside a pool worker runs serially.
class HostPool {
  public:
    static HostPool& get() {
        static HostPool* p = new HostPool;  // never destroyed: workers may outlive static destructors
        return *p;
    }
    // runs `work()` on the calling thread and on up to helpers pool threads; returns when all are back
    template <class F>
    void run(int helpers, F&& work) {
        if (helpers <= 0 || in_worker()) { work(); return; }
        std::unique_lock<std::mutex> job(job_m_);   // one loop at a time
        {
            std::unique_lock<std::mutex> lk(m_);
            grow(helpers);
            fn_ = [&work]() { work(); };
            slots_ = helpers;
            ++gen_;
        }
        cv_work_.notify_all();
        in_worker() = true;   // a loop started by `work` itself runs serially (job_m_ is not recursive)
        work();
        in_worker() = false;
        std::unique_lock<std::mutex> lk(m_);
        slots_ = 0;                                 // nobody joins any more
        cv_done_.wait(lk, [&] { return running_ == 0; });
        fn_ = nullptr;
    }

  private:
    static bool& in_worker() { static thread_local bool w = false; return w; }
    void grow(int want) {  // m_ held
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && want > hw) want = hw;
        while ((int)th_.size() < want) {
            th_.emplace_back([this] { loop(); });
            th_.back().detach();
        }
    }
    void loop() {
        in_worker() = true;
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (slots_ <= 0) continue;
            --slots_;
            ++running_;
            std::function<void()> f = fn_;
            lk.unlock();
            f();
            lk.lock();
            if (--running_ == 0) cv_done_.notify_all();
        }
    }
    std::mutex job_m_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> th_;
    std::function<void()> fn_;
    int slots_ = 0, running_ = 0;
    uint64_t gen_ = 0;
};

// Fork-join over [0, n): up to `threads` host threads (the caller + pool workers) pull blocks of
// `grain` indices.
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
    const int64_t blocks = (n + grain - 1) / grain;
    const int nt = (int)(blocks < threads ? blocks : threads);
    HostPool::get().run(nt - 1, work);
}

// 2-bit packing of one chunk (16 bases per dword, base i at bits 2*(i&15)); returns true if the chunk
// contains N (whose 2-bit code is 0; the N mask carries it).  Eight bases per 64-bit step: for the
// validated alphabet ((c >> 1) ^ (c >> 2)) & 3 maps A,C,G,T to 0,1,2,3 and N to 0.
inline uint32_t pack8(uint64_t x, uint64_t& nacc) {
    const uint64_t y = x ^ 0x4E4E4E4E4E4E4E4Eull;                       // a zero byte where the base is 'N'
    nacc |= (y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull;
    uint64_t t = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
    t = (t | (t >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    t = (t | (t >> 24)) & 0xFFFFull;
    return (uint32_t)t;
}
inline bool pack_chunk(const char* s, int32_t l, uint32_t* out) {
    uint64_t nacc = 0;
    const int32_t full = l & ~15;
    for (int32_t i = 0; i < full; i += 16) {
        uint64_t a, b;
        std::memcpy(&a, s + i, 8);
        std::memcpy(&b, s + i + 8, 8);
        out[i >> 4] = pack8(a, nacc) | (pack8(b, nacc) << 16);
    }
    bool has_n = nacc != 0;
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
