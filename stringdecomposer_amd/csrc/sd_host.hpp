// sd_host.hpp -- host side of the hot path: FASTA ingest, reverse complements, chunk table,
// 2-bit packing, per-read assembly (offset + seam merge) and raw-TSV formatting.
//
// Behaviour follows ablab/stringdecomposer v1.1.2 stringdecomposer/src/main.cpp (cited per
// function); the code is written from scratch for this library.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <sys/vfs.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/sd_hip.h"

namespace sd {

// edlib computes the path of an alignment by its block traceback while the traceback data fits 1 MB and by
// Hirschberg's split of the target otherwise (edlib.cpp:1186-1192): true for the second case (~19.6 kb against a
// 171-bp monomer).  The host identities (sd_post.hip) follow both; the device kernels implement the traceback only.
inline bool edlib_splits(int64_t qlen, int64_t tlen) {
    return 20ll * ((qlen + 63) / 64) * tlen + 8ll * tlen >= 1024 * 1024;
}

struct Seq {
    std::string name;
    std::string seq;
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
inline bool cpu_has_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}
// number of leading bytes of s[0, n) (a multiple of 32) that are all in {A, C, G, T, N}: one byte shuffle looks the
// only valid byte with a given low nibble up (1 -> 'A', 3 -> 'C', 7 -> 'G', 4 -> 'T', 14 -> 'N'), one compare tests it
__attribute__((target("avx2"))) inline int64_t valid_prefix_avx2(const char* s, int64_t n, bool* saw_n = nullptr) {
    const __m256i vn = _mm256_set1_epi8('N');
    __m256i nacc = _mm256_setzero_si256();
    const __m256i lut = _mm256_setr_epi8(0, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 'N', 0,
                                         0, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 'N', 0);
    const __m256i lo = _mm256_set1_epi8(0x0f);
    int64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i));
        // (a byte with its top bit set selects 0 in the shuffle and 0 never equals it; a zero byte is caught below)
        const __m256i want = _mm256_shuffle_epi8(lut, _mm256_and_si256(x, lo));
        const __m256i ok = _mm256_andnot_si256(_mm256_cmpeq_epi8(x, _mm256_setzero_si256()), _mm256_cmpeq_epi8(x, want));
        if (_mm256_movemask_epi8(ok) != -1) break;
        nacc = _mm256_or_si256(nacc, _mm256_cmpeq_epi8(x, vn));
    }
    if (saw_n && _mm256_movemask_epi8(nacc) != 0) *saw_n = true;
    return i;
}
#endif

// alphabet check for in-memory sequences (same rule and message as load_fasta)
inline int check_alphabet(const char* name, const char* s, int64_t n, std::string& err) {
    int64_t i0 = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (n >= 64 && cpu_has_avx2()) i0 = valid_prefix_avx2(s, n);   // the scalar loop names the offending byte
#endif
    for (int64_t i = i0; i < n; ++i) {
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
// the same merge with a parallel array that follows the records (where a kept record came from)
inline size_t seam_merge_inplace(sd_rec* b, int64_t* src, size_t N) {
    size_t w = 0, i = 0;
    while (i < N) {
        const size_t lim = i + 7 < N ? i + 7 : N;
        for (size_t j = i + 1; j < lim; ++j) {
            if ((b[i].end - b[j].start) * 2 > (b[j].end - b[j].start)) {
                src[w] = src[i];
                b[w++] = b[i];
                i = j + 1;
                break;
            }
        }
        if (i < N) { src[w] = src[i]; b[w++] = b[i]; }
        ++i;
    }
    return w;
}

// Text of a batch: a byte vector whose resize() does not zero-fill (the parts of a batch are copied into it by
// all threads; a std::string would first write 200 MB of zeros for the _alt rows of a --second-best batch).
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
using TextBuf = std::vector<char, NoInitAlloc<char>>;
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;   // resize() leaves the new elements unwritten

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

// "%.2f" of a double, appended: the correctly rounded (ties to even, on the exact binary value) two-decimal
// text that printf and Python's "{:.2f}".format give (main.py:157-165), without printf: the value is
// m * 2^-sh exactly, so floor(100 v) and the remainder come from one 128-bit product.  snprintf takes
// 0.4-0.6 us per call and the final TSV holds four such fields per row (the _alt TSV one per row).
inline void put_fixed2(std::string& o, double v) {
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    const int ex = (int)((bits >> 52) & 0x7ff);
    uint64_t m = bits & ((1ull << 52) - 1);
    if (ex == 0x7ff || ex >= 1023 + 40) {   // inf / nan / |v| >= 2^40: not a percentage; let printf do it
        char b[400];
        const int n = std::snprintf(b, sizeof b, "%.2f", v);
        o.append(b, (size_t)n);
        return;
    }
    int sh;   // |v| = m * 2^-sh
    if (ex) { m |= 1ull << 52; sh = 1075 - ex; } else sh = 1074;
    const unsigned __int128 num = (unsigned __int128)m * 100u;   // < 2^60
    uint64_t q = 0;
    if (sh < 64) {   // sh >= 13 here
        const uint64_t n64 = (uint64_t)num;
        q = n64 >> sh;
        const uint64_t rem = n64 & ((1ull << sh) - 1), half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;
    }                // sh >= 64: 100 |v| < 2^-4, rounds to 0
    char buf[32];
    int p = 32;
    buf[--p] = (char)('0' + q % 10); q /= 10;
    buf[--p] = (char)('0' + q % 10); q /= 10;
    buf[--p] = '.';
    do { buf[--p] = (char)('0' + q % 10); q /= 10; } while (q);
    if (bits >> 63) buf[--p] = '-';
    o.append(buf + p, (size_t)(32 - p));
}

// the same text written at `w` (room for 48 bytes), returns the end
inline char* put_fixed2_at(char* w, double v) {
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    const int ex = (int)((bits >> 52) & 0x7ff);
    uint64_t m = bits & ((1ull << 52) - 1);
    if (ex == 0x7ff || ex >= 1023 + 40) return w + std::snprintf(w, 48, "%.2f", v);
    int sh;
    if (ex) { m |= 1ull << 52; sh = 1075 - ex; } else sh = 1074;
    const unsigned __int128 num = (unsigned __int128)m * 100u;
    uint64_t q = 0;
    if (sh < 64) {
        const uint64_t n64 = (uint64_t)num;
        q = n64 >> sh;
        const uint64_t rem = n64 & ((1ull << sh) - 1), half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;
    }
    char buf[32];
    int p = 32;
    buf[--p] = (char)('0' + q % 10); q /= 10;
    buf[--p] = (char)('0' + q % 10); q /= 10;
    buf[--p] = '.';
    do { buf[--p] = (char)('0' + q % 10); q /= 10; } while (q);
    if (bits >> 63) buf[--p] = '-';
    std::memcpy(w, buf + p, (size_t)(32 - p));
    return w + (32 - p);
}

// SaveBatch (main.cpp:272-285).  The reference prints to_string(float identity) == "%f"; the
// identity is an integer-valued float (|v| < 1e6 < 2^24 by the range check in the engine), so
// "%f" is exactly "<int>.000000".
// decimal text of an int written at w, returns the end (room for 12 bytes)
inline char* put_int_at(char* w, int64_t v) {
    char buf[24];
    int p = 24;
    const bool neg = v < 0;
    uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    do { buf[--p] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) buf[--p] = '-';
    std::memcpy(w, buf + p, (size_t)(24 - p));
    return w + (24 - p);
}

inline void format_rows(std::string& o, const char* read_name, size_t read_name_len,
                        const std::vector<std::string>& tnames, const sd_rec* rows, size_t n,
                        int prev_end = 0) {  // prev_end: end of the row before rows[0] (0 at a read's start)
    // Written through a pointer into room reserved for the worst case (a std::string that grows field by field spent
    // 580 ns per row on appends and reallocations: 61 ms for the 1.17 M rows of a 200-Mb sequence on 16 threads).
    size_t tmax = 0;
    for (const std::string& t : tnames) tmax = std::max(tmax, t.size());
    const size_t row_max = read_name_len + tmax + 5 * 12 + 8 + 7;   // two names, five ints, ".000000", separators
    const size_t at0 = o.size();
    o.resize(at0 + n * row_max);
    char* w = &o[at0];
    for (size_t x = 0; x < n; ++x) {
        const sd_rec& r = rows[x];
        std::memcpy(w, read_name, read_name_len); w += read_name_len;
        *w++ = '\t';
        const std::string& tn = tnames[(size_t)r.tmpl];
        std::memcpy(w, tn.data(), tn.size()); w += tn.size();
        *w++ = '\t';
        w = put_int_at(w, r.start);
        *w++ = '\t';
        w = put_int_at(w, r.end);
        *w++ = '\t';
        w = put_int_at(w, r.score);
        std::memcpy(w, ".000000\t", 8); w += 8;
        w = put_int_at(w, (int64_t)r.start - prev_end);
        *w++ = '\t';
        w = put_int_at(w, (int64_t)r.end - r.start);
        *w++ = '\n';
        prev_end = r.end;
    }
    o.resize((size_t)(w - o.data()));
}

// Process-wide pool of host worker threads.  The packer, the assembler and the formatter run many
// short parallel loops per device batch (a few hundred microseconds each); creating 32-64 threads for
// every loop cost more than the loops, so the threads persist and sleep between loops.  One loop at a
// time (a second caller waits its turn); a loop started from inside a pool worker runs serially.
class HostPool {
  public:
    // Two pools: a loop runs on the pool of the calling thread's lane.  Lane 1 is the sink thread of sd_run_files
    // (text + post-processing of batch b), lane 0 everything else (the driver packing batch b+1): with one pool
    // the two took turns -- "one loop at a time" -- and each waited for the other's loops.
    static int& lane() { static thread_local int l = 0; return l; }
    static HostPool& get() {
        static HostPool* p[3] = {new HostPool, new HostPool, new HostPool};  // never destroyed: workers may outlive static destructors
        return *p[(unsigned)lane() % 3u];   // lane 2: the file writer of sd_run_files
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

// out = concatenation of parts, copied by up to `threads` threads (parts are typically 0.1-1 MB each)
template <class Part>
inline void gather_text(const std::vector<Part>& parts, int threads, TextBuf& out) {
    std::vector<size_t> off(parts.size() + 1, 0);
    for (size_t i = 0; i < parts.size(); ++i) off[i + 1] = off[i] + parts[i].size();
    out.resize(off[parts.size()]);
    parallel_for((int64_t)parts.size(), threads, 4, [&](int64_t i) {
        if (!parts[(size_t)i].empty()) std::memcpy(out.data() + off[(size_t)i], parts[(size_t)i].data(), parts[(size_t)i].size());
    });
}

// test hook of write_parts: a non-zero value makes the page reservation of the mapped path "fail"
inline std::atomic<int>& write_parts_hook() {
    static std::atomic<int> h{0};
    return h;
}
inline bool write_parts_fallocate_ok() { return write_parts_hook().load(std::memory_order_relaxed) == 0; }

// parts -> the file `fd` at `off` (advanced by the total).  The cost of a TSV write is the copy into the page cache
// (300 MB of _alt rows per --second-best batch).  On tmpfs, where write(2) serialises on the inode, large texts are copied
// by all threads through a shared mapping of the file's new range (65 instead of 88 ms per 297 MB); on a disk file system
// a page fault of a mapped write allocates blocks one page at a time (measured: 100 ms per 25 MB on the GPU box's /tmp),
// so everything else is ONE pwritev stream in order, as an fwrite would do.
// A store into a sparse tmpfs mapping that the file system cannot back (full or small /dev/shm) raises SIGBUS and kills
// the process, so the mapped copy is taken only when fallocate has really reserved the pages of the new range; the
// pwritev loop below reports the same condition as a short write (-> SD_ERR_IO).  fallocate zeroes the pages on ONE
// thread (36-49 ms per 280 MB at C4: what bounds that job).  Round 5 measured the alternative that reports failure
// without a signal and runs on all threads -- every copying thread populating the pages of its part with
// madvise(MADV_POPULATE_WRITE) -- on the GPU box: 66-87 ms per 294 MB against 35-44 (fallocate) and 43-54 (pwritev), C4
// 106-123 ms per step against 76-80: concurrent page allocation in one tmpfs file contends harder than one thread
// zeroing; not kept.  SD_WRITE_PATH=1: pwritev only (A/B).
// sd_write_parts_test_hook() != 0 makes fallocate "fail" (CPU test of the fall-back).
template <class Part>
inline bool write_parts(int fd, int64_t& off, const std::vector<Part>& parts, int threads) {
    std::vector<int64_t> at(parts.size() + 1, off);
    for (size_t i = 0; i < parts.size(); ++i) at[i + 1] = at[i] + (int64_t)parts[i].size();
    const int64_t total = at[parts.size()] - off;
    if (total == 0) return true;
    struct statfs fs;
    const bool ram = ::fstatfs(fd, &fs) == 0 && ((unsigned long)fs.f_type == 0x01021994ul /* tmpfs */ ||
                                                 (unsigned long)fs.f_type == 0x858458f6ul /* ramfs */);
    static const int wmode = [] { const char* e = getenv("SD_WRITE_PATH"); return e ? atoi(e) : 0; }();   // developer A/B: 1 = pwritev only
    if (ram && wmode != 1 && total >= (4 << 20) && write_parts_fallocate_ok() &&
        ::fallocate(fd, 0, (off_t)off, (off_t)total) == 0) {
        // pages allocated in one call: the copies below then take minor faults only (300 MB on the GPU box: 52-63 ms
        // against 83-104 ms with every page allocated by the fault of a copying thread; tools/scratch/tmpfs_write.cpp)
        const long pg = ::sysconf(_SC_PAGESIZE);
        const int64_t m0 = off / pg * pg;
        // (MAP_POPULATE -- the page tables filled by one call instead of a minor fault per page in the copying threads --
        // was measured at C4: 66-69 ms of writes per 280 MB against 36-37: the populate is single-threaded)
        void* mp = ::mmap(nullptr, (size_t)(off + total - m0), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)m0);
        if (mp != MAP_FAILED) {
            char* base = static_cast<char*>(mp) - m0;   // base + file offset
            parallel_for((int64_t)parts.size(), threads, 1, [&](int64_t i) {
                const Part& p = parts[(size_t)i];
                if (p.size()) std::memcpy(base + at[(size_t)i], p.data(), p.size());
            });
            ::munmap(mp, (size_t)(off + total - m0));
            off += total;
            return true;
        }
    }
    // in order, up to 1024 parts per call
    size_t i = 0;
    int64_t pos = off;
    while (i < parts.size()) {
        struct iovec iov[1024];
        int n = 0;
        size_t j = i;
        for (; j < parts.size() && n < 1024; ++j)
            if (parts[j].size()) { iov[n].iov_base = const_cast<char*>(parts[j].data()); iov[n].iov_len = parts[j].size(); ++n; }
        int64_t want = 0;
        for (int k = 0; k < n; ++k) want += (int64_t)iov[k].iov_len;
        int k0 = 0;
        while (want > 0) {
            const ssize_t got = ::pwritev(fd, iov + k0, n - k0, (off_t)pos);
            if (got <= 0) return false;
            pos += got;
            want -= got;
            ssize_t g = got;
            while (g > 0 && k0 < n) {   // skip what was written (a short write ends inside an element)
                if ((size_t)g >= iov[k0].iov_len) { g -= (ssize_t)iov[k0].iov_len; ++k0; }
                else { iov[k0].iov_base = static_cast<char*>(iov[k0].iov_base) + g; iov[k0].iov_len -= (size_t)g; g = 0; }
            }
        }
        i = j;
    }
    off = at[parts.size()];
    return true;
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
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// 32 bases per step with AVX2: the same two-bit code per byte, four codes folded into a byte by two multiply-adds
// (weights 1, 4 and 1, 16), the eight bytes of a step gathered by one byte shuffle.  Returns the bases consumed (a
// multiple of 32); *has_n as in pack_chunk.  4-5 times the SWAR loop per thread: what a rank with two host threads
// (eight ranks on a 16-CPU box) needs to keep the packing of a 50-Mbp step under 2 ms.
__attribute__((target("avx2"))) inline int32_t pack_chunk_avx2(const char* s, int32_t l, uint32_t* out, bool* has_n) {
    const __m256i m3 = _mm256_set1_epi8(3), vn = _mm256_set1_epi8('N');
    const __m256i w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i nacc = _mm256_setzero_si256();
    int32_t i = 0;
    for (; i + 32 <= l; i += 32) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i));
        nacc = _mm256_or_si256(nacc, _mm256_cmpeq_epi8(x, vn));
        const __m256i t = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(x, 1), _mm256_srli_epi16(x, 2)), m3);
        const __m256i b = _mm256_madd_epi16(_mm256_maddubs_epi16(t, w14), w116);   // one byte of output per 32-bit lane
        const __m256i c = _mm256_shuffle_epi8(b, gather);
        out[(i >> 4)] = (uint32_t)_mm256_extract_epi32(c, 0);
        out[(i >> 4) + 1] = (uint32_t)_mm256_extract_epi32(c, 4);
    }
    *has_n = _mm256_movemask_epi8(nacc) != 0;
    return i;
}
#endif

inline bool pack_chunk(const char* s, int32_t l, uint32_t* out) {
    uint64_t nacc = 0;
    int32_t done = 0;
    bool has_n0 = false;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (l >= 64 && cpu_has_avx2()) done = pack_chunk_avx2(s, l, out, &has_n0);
#endif
    const int32_t full = l & ~15;
    for (int32_t i = done; i < full; i += 16) {
        uint64_t a, b;
        std::memcpy(&a, s + i, 8);
        std::memcpy(&b, s + i + 8, 8);
        out[i >> 4] = pack8(a, nacc) | (pack8(b, nacc) << 16);
    }
    bool has_n = nacc != 0 || has_n0;
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

// ---------------------------------------------------------------------------------------------
// FastaFile: load_fasta (main.cpp:314-346) for files of any size.  The file is mapped, record starts
// ('>' at the beginning of a line) are found by all host threads, and every record is then parsed on
// its own: name = first whitespace token of the header, sequence = the following lines appended
// verbatim.  A sequence that sits on ONE line is used where it lies in the mapping (no copy); a
// multi-line sequence is concatenated into a buffer of its own.  The alphabet {A,C,G,T,N} is enforced
// per record (validate()); the first offending record in file order is reported with the reference's
// text.  For a job sharded over ranks only the records a rank's chunk range touches need validating.
// ---------------------------------------------------------------------------------------------
class FastaFile {
  public:
    struct Rec {
        const char* name; size_t name_len;
        const char* seq; int64_t len;     // into the mapping or into `owned`
    };
    std::vector<Rec> recs;
    bool has_n = false;

    FastaFile() = default;
    FastaFile(const FastaFile&) = delete;
    FastaFile& operator=(const FastaFile&) = delete;
    ~FastaFile() { close_(); }

    int open(const std::string& path, int threads, std::string& err) {
        close_();
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) { err = "cannot open " + path; return SD_ERR_IO; }
        struct stat sb;
        if (fstat(fd_, &sb) != 0) { err = "cannot open " + path; return SD_ERR_IO; }
        size_ = (size_t)sb.st_size;
        if (size_ == 0) return SD_OK;
        void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) {  // not mappable (a pipe, a special file): read it
            map_ = nullptr;
            fallback_.resize(size_);
            size_t got = 0;
            while (got < size_) {
                const ssize_t r = ::pread(fd_, &fallback_[got], size_ - got, (off_t)got);
                if (r <= 0) { err = "cannot read " + path; return SD_ERR_IO; }
                got += (size_t)r;
            }
            data_ = fallback_.data();
        } else {
            map_ = m;
            data_ = static_cast<const char*>(m);
            (void)madvise(m, size_, MADV_SEQUENTIAL);
        }
        return index(threads, err);
    }
    // alphabet check of records [r0, r1) (main.cpp:329-344); sets has_n
    int validate(size_t r0, size_t r1, int threads, std::string& err) {
        r1 = std::min(r1, recs.size());
        if (r0 >= r1) return SD_OK;
        // pieces of <= 4 MB so that one long sequence is checked by all threads
        struct Piece { size_t rec; int64_t off, len; };
        std::vector<Piece> pieces;
        for (size_t r = r0; r < r1; ++r)
            for (int64_t o = 0; o < recs[r].len; o += (4 << 20))
                pieces.push_back(Piece{r, o, std::min<int64_t>(4 << 20, recs[r].len - o)});
        std::vector<int64_t> bad(pieces.size(), -1);
        std::vector<uint8_t> hn(pieces.size(), 0);
        parallel_for((int64_t)pieces.size(), threads, 1, [&](int64_t x) {
            const Piece& pc = pieces[(size_t)x];
            const char* q = recs[pc.rec].seq + pc.off;
            uint8_t n = 0;
            int64_t i0 = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
            if (pc.len >= 64 && cpu_has_avx2()) {
                bool sn = false;
                i0 = valid_prefix_avx2(q, pc.len, &sn);
                if (sn) n = 1;
            }
#endif
            for (int64_t i = i0; i < pc.len; ++i) {
                const char c = q[i];
                if (c == 'A' || c == 'C' || c == 'G' || c == 'T') continue;
                if (c == 'N') { n = 1; continue; }
                bad[(size_t)x] = i;
                break;
            }
            hn[(size_t)x] = n;
        });
        for (size_t x = 0; x < pieces.size(); ++x) {
            if (bad[x] >= 0) {
                const Rec& rc = recs[pieces[x].rec];
                err = "ERROR: Sequence " + std::string(rc.name, rc.name_len) + " contains undefined symbol (not ACGT): " +
                      rc.seq[pieces[x].off + bad[x]];
                return SD_ERR_SYMBOL;
            }
            if (hn[x]) has_n = true;
        }
        return SD_OK;
    }

  private:
    int fd_ = -1;
    void* map_ = nullptr;
    const char* data_ = nullptr;
    size_t size_ = 0;
    std::string fallback_;
    std::vector<std::string> owned_;   // concatenated multi-line sequences

    void close_() {
        if (map_) munmap(map_, size_);
        if (fd_ >= 0) ::close(fd_);
        map_ = nullptr;
        fd_ = -1;
        data_ = nullptr;
        size_ = 0;
        recs.clear();
        owned_.clear();
        fallback_.clear();
    }
    int index(int threads, std::string& err) {
        const char* d = data_;
        const size_t N = size_;
        // (1) record starts: '>' at offset 0 or right after a newline
        const size_t slab = (size_t)8 << 20;
        const int64_t n_slabs = (int64_t)((N + slab - 1) / slab);
        std::vector<std::vector<size_t>> found((size_t)n_slabs);
        parallel_for(n_slabs, threads, 1, [&](int64_t sl) {
            const size_t b = (size_t)sl * slab, e = std::min(N, b + slab);
            std::vector<size_t>& out = found[(size_t)sl];
            const char* p = d + b;
            const char* end = d + e;
            while (p < end) {
                p = static_cast<const char*>(std::memchr(p, '>', (size_t)(end - p)));
                if (!p) break;
                const size_t pos = (size_t)(p - d);
                if (pos == 0 || d[pos - 1] == '\n') out.push_back(pos);
                ++p;
            }
        });
        std::vector<size_t> starts;
        for (auto& v : found) starts.insert(starts.end(), v.begin(), v.end());
        {
            // anything before the first header must be blank lines (the reference would append it to a
            // record that does not exist, main.cpp:327)
            const size_t first = starts.empty() ? N : starts[0];
            for (size_t i = 0; i < first; ++i)
                if (d[i] != '\n') { err = "FASTA does not start with a header"; return SD_ERR_FORMAT; }
        }
        const size_t R = starts.size();
        recs.assign(R, Rec{nullptr, 0, nullptr, 0});
        owned_.assign(R, std::string());
        std::vector<uint8_t> bad(R, 0);
        // (2) every record on its own
        parallel_for((int64_t)R, threads, 4, [&](int64_t r) {
            const size_t b = starts[(size_t)r], e = (size_t)r + 1 < R ? starts[(size_t)r + 1] : N;
            const char* nl = static_cast<const char*>(std::memchr(d + b, '\n', e - b));
            const size_t hdr_end = nl ? (size_t)(nl - d) : e;
            size_t a = b + 1;
            while (a < hdr_end && is_ws(d[a])) ++a;
            size_t z = a;
            while (z < hdr_end && !is_ws(d[z])) ++z;
            if (a == z) { bad[(size_t)r] = 1; return; }
            Rec& rc = recs[(size_t)r];
            rc.name = d + a;
            rc.name_len = z - a;
            size_t pos = hdr_end < e ? hdr_end + 1 : e;
            // sequence lines: [pos, e)
            // fast path: exactly one non-empty line
            size_t lines = 0, total = 0, first_b = 0, first_l = 0;
            for (size_t q = pos; q < e;) {
                const char* n2 = static_cast<const char*>(std::memchr(d + q, '\n', e - q));
                const size_t le = n2 ? (size_t)(n2 - d) : e;
                if (le > q) {
                    if (lines == 0) { first_b = q; first_l = le - q; }
                    ++lines;
                    total += le - q;
                }
                q = le + 1;
            }
            if (lines <= 1) {
                rc.seq = d + first_b;
                rc.len = (int64_t)first_l;
            } else {
                std::string& o = owned_[(size_t)r];
                o.reserve(total);
                for (size_t q = pos; q < e;) {
                    const char* n2 = static_cast<const char*>(std::memchr(d + q, '\n', e - q));
                    const size_t le = n2 ? (size_t)(n2 - d) : e;
                    if (le > q) o.append(d + q, le - q);
                    q = le + 1;
                }
                rc.seq = o.data();
                rc.len = (int64_t)o.size();
            }
        });
        for (size_t r = 0; r < R; ++r)
            if (bad[r]) { err = "FASTA header without a name"; return SD_ERR_FORMAT; }
        return SD_OK;
    }
};

}  // namespace sd
