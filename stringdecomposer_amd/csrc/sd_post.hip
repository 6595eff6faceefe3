// sd_post.hip -- host-side post-processing kernel of the drop-in CLI: unit-cost global alignment
// identity of read segments against templates (what stringdecomposer/main.py:29-60 gets from
// python-edlib).  Host code only (compiled by hipcc for a uniform build); multi-threaded over pairs.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/sd_hip.h"
#include "sd_host.hpp"

namespace {

// longest sequence of the host identities (memory: the block traceback of a pair stays below edlib's 1 MB, the split
// form keeps one column)
constexpr int32_t SD_NW_HOST_MAX = 1 << 27;

// Unit-cost global alignment by Myers' bit-vector algorithm (J. ACM 46(3), 1999; block form of
// Hyyro 2003): query along the bit rows, one column per target symbol, the vertical delta vectors
// (Pv, Mv) of every column are kept so that any D[i][j] = j + popcount(Pv_j & low_i) -
// popcount(Mv_j & low_i) can be read back.  The traceback then applies the same rule as a
// full-matrix walk from the bottom-right corner: up ('I', consume query) > left ('D', consume
// target) > diagonal ('=' / 'X').  Alignment columns = edit distance + matches.
using sd::edlib_splits;

struct BitNW {
    std::vector<uint64_t> peq;  // [classes][K]
    std::vector<uint64_t> pv, mv;  // [(tlen + 1)][K]
    std::vector<uint64_t> cp, cm;  // one column (column scores)
    std::vector<int32_t> colL, colR;
    std::string rq, rt;
    uint16_t cls[256];
    // symbol classes of one pair: every byte of the query its own class, bytes only the target has share class 0
    // (they match nothing) -- edlib's alphabet is the set of bytes that occur
    int classes(const char* q, int qlen) {
        std::memset(cls, 0, sizeof cls);
        int n = 1;
        for (int i = 0; i < qlen; ++i) {
            uint16_t& c = cls[(uint8_t)q[i]];
            if (!c) c = (uint16_t)n++;
        }
        return n;
    }
    void build_peq(const char* q, int qlen, int K, int ncls) {
        peq.assign((size_t)ncls * K, 0);
        for (int i = 0; i < qlen; ++i) peq[(size_t)cls[(uint8_t)q[i]] * K + (i >> 6)] |= 1ull << (i & 63);
    }
    // one column step of the unit-cost global alignment (Myers / Hyyro block form), in place or into (np, nm)
    static inline void column_step(const uint64_t* eqs, const uint64_t* pp, const uint64_t* pm, uint64_t* np, uint64_t* nm,
                                   int K) {
        int hin = 1;  // global alignment: D[0][j] - D[0][j-1] = 1
        for (int b = 0; b < K; ++b) {
            uint64_t Eq = eqs[b];
            const uint64_t Pv = pp[b], Mv = pm[b];
            const uint64_t Xv = Eq | Mv;
            if (hin < 0) Eq |= 1ull;
            const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint64_t Ph = Mv | ~(Xh | Pv);
            uint64_t Mh = Pv & Xh;
            int hout = 0;
            if (Ph >> 63) hout = 1;
            if (Mh >> 63) hout = -1;
            Ph <<= 1;
            Mh <<= 1;
            if (hin < 0) Mh |= 1ull;
            if (hin > 0) Ph |= 1ull;
            np[b] = Mh | ~(Xv | Ph);
            nm[b] = Ph & Xv;
            hin = hout;
        }
    }
    // out[i] = distance of q[0..i) to t[0..tlen), i = 0..qlen (memory: one column)
    void column_scores(const char* q, int qlen, const char* t, int tlen, std::vector<int32_t>& out) {
        const int K = (qlen + 63) >> 6;
        const int ncls = classes(q, qlen);
        build_peq(q, qlen, K, ncls);
        cp.assign((size_t)K, ~0ull);
        cm.assign((size_t)K, 0);
        for (int j = 1; j <= tlen; ++j)
            column_step(&peq[(size_t)cls[(uint8_t)t[j - 1]] * K], cp.data(), cm.data(), cp.data(), cm.data(), K);
        out.resize((size_t)qlen + 1);
        int v = tlen;
        out[0] = v;
        for (int i = 0; i < qlen; ++i) {
            const uint64_t bit = 1ull << (i & 63);
            v += ((cp[(size_t)(i >> 6)] & bit) ? 1 : 0) - ((cm[(size_t)(i >> 6)] & bit) ? 1 : 0);
            out[(size_t)i + 1] = v;
        }
    }
    // Unit-cost global alignment by Myers' bit-vector algorithm (J. ACM 46(3), 1999; block form of Hyyro 2003):
    // query along the bit rows, one column per target symbol, the vertical delta vectors (Pv, Mv) of every column
    // are kept so that any D[i][j] = j + popcount(Pv_j & low_i) - popcount(Mv_j & low_i) can be read back.  The
    // traceback then applies the same rule as a full-matrix walk from the bottom-right corner: up ('I', consume
    // query) > left ('D', consume target) > diagonal ('=' / 'X') -- edlib.cpp:945-1130.  Returns the distance; adds
    // the '=' columns of the path to `matches`.
    int traceback(const char* q, int qlen, const char* t, int tlen, int32_t& matches) {
        const int K = (qlen + 63) >> 6;
        const int ncls = classes(q, qlen);
        build_peq(q, qlen, K, ncls);
        pv.resize((size_t)(tlen + 1) * K);
        mv.resize((size_t)(tlen + 1) * K);
        for (int b = 0; b < K; ++b) { pv[b] = ~0ull; mv[b] = 0; }
        for (int j = 1; j <= tlen; ++j)
            column_step(&peq[(size_t)cls[(uint8_t)t[j - 1]] * K], &pv[(size_t)(j - 1) * K], &mv[(size_t)(j - 1) * K],
                        &pv[(size_t)j * K], &mv[(size_t)j * K], K);
        auto D = [&](int i, int j) -> int {  // i rows of the query consumed, j target symbols
            const uint64_t* P = &pv[(size_t)j * K];
            const uint64_t* M = &mv[(size_t)j * K];
            int v = j;
            const int full = i >> 6, rem = i & 63;
            for (int b = 0; b < full; ++b) v += __builtin_popcountll(P[b]) - __builtin_popcountll(M[b]);
            if (rem) {
                const uint64_t low = (1ull << rem) - 1;
                v += __builtin_popcountll(P[full] & low) - __builtin_popcountll(M[full] & low);
            }
            return v;
        };
        int i = qlen, j = tlen, m = 0;
        int cur = D(i, j);
        const int dist = cur;
        while (i > 0 || j > 0) {
            int up = -1;
            if (i > 0) {  // D[i-1][j] from the vertical delta of row i in column j
                const uint64_t bit = 1ull << ((i - 1) & 63);
                const size_t w = (size_t)j * K + ((i - 1) >> 6);
                up = cur - ((pv[w] & bit) ? 1 : 0) + ((mv[w] & bit) ? 1 : 0);
            }
            if (i > 0 && up + 1 == cur) { --i; cur = up; continue; }          // 'I'
            if (j > 0) {
                const int left = D(i, j - 1);
                if (left + 1 == cur) { --j; cur = left; continue; }          // 'D'
            }
            const int dg = D(i - 1, j - 1);                                    // '=' or 'X'
            if (dg == cur) ++m;
            --i; --j; cur = dg;
        }
        matches += m;
        return dist;
    }
    // The path edlib reports for a (sub)problem whose optimum `best` is known (obtainAlignment, edlib.cpp:1164-1213):
    // small problems by the block traceback; once its data would reach 1 MB the TARGET is split in halves
    // (obtainAlignmentHirschberg, edlib.cpp:1234-1400): with L[i] = distance of q[0..i) to the left half and R[k] =
    // distance of the last k query symbols to the right half, the split row is the SMALLEST x in 0..qlen-2 with
    // L[x+1] + R[qlen-x-1] == best, else x = -1 if lw + R[qlen] == best, else x = qlen-1 if L[qlen] + rw == best
    // (edlib.cpp:1315-1349; its banded columns hold every cell of an optimal path exactly, so the full columns
    // find the same row), and the paths of (q[0..x], left half) and (q[x+1..), right half) are concatenated.
    bool path(const char* q, int qlen, const char* t, int tlen, int best, int32_t& matches) {
        if (qlen == 0 || tlen == 0) return true;
        if (!edlib_splits(qlen, tlen)) return traceback(q, qlen, t, tlen, matches) == best;
        const int lw = tlen / 2, rw = tlen - lw;
        std::vector<int32_t> L, R;
        column_scores(q, qlen, t, lw, L);
        std::string rq2(q, q + qlen), rt2(t + lw, t + tlen);
        std::reverse(rq2.begin(), rq2.end());
        std::reverse(rt2.begin(), rt2.end());
        column_scores(rq2.data(), qlen, rt2.data(), rw, R);
        int x = -2, ls = 0, rs = 0;
        for (int i = 0; i <= qlen - 2; ++i)
            if (L[(size_t)i + 1] + R[(size_t)(qlen - i - 1)] == best) { x = i; ls = L[(size_t)i + 1]; rs = R[(size_t)(qlen - i - 1)]; break; }
        if (x == -2 && lw + R[(size_t)qlen] == best) { x = -1; ls = lw; rs = R[(size_t)qlen]; }
        if (x == -2 && L[(size_t)qlen] + rw == best) { x = qlen - 1; ls = L[(size_t)qlen]; rs = rw; }
        if (x == -2) return false;
        const int ul = x + 1;
        return path(q, ul, t, lw, ls, matches) && path(q + ul, qlen - ul, t + lw, rw, rs, matches);
    }
    void run(const char* q, int qlen, const char* t, int tlen, int32_t& dist, int32_t& matches,
             int32_t& columns) {
        dist = -1;
        matches = 0;
        columns = 0;
        if (qlen <= 0 || tlen <= 0) return;
        int32_t m = 0;
        int d;
        if (!edlib_splits(qlen, tlen)) {
            d = traceback(q, qlen, t, tlen, m);
        } else {
            column_scores(q, qlen, t, tlen, colL);
            d = colL[(size_t)qlen];
            if (!path(q, qlen, t, tlen, d, m)) return;   // cannot happen: d is the optimum
        }
        dist = d;
        matches = m;
        columns = d + m;
    }
};

}  // namespace

extern "C" int sd_nw_identity_batch(const char* const* queries, const int32_t* qlens,
                                    const char* const* targets, const int32_t* tlens,
                                    int64_t n_pairs, int32_t threads, int32_t* dist,
                                    int32_t* matches, int32_t* columns) {
    if (n_pairs < 0 || !queries || !targets || !qlens || !tlens || !matches || !columns) return SD_ERR_PARAM;
    for (int64_t i = 0; i < n_pairs; ++i)
        if (qlens[i] > SD_NW_HOST_MAX || tlens[i] > SD_NW_HOST_MAX) return SD_ERR_UNSUPPORTED;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_pairs));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        BitNW nw;
        for (;;) {
            const int64_t b = next.fetch_add(16);
            if (b >= n_pairs) break;
            const int64_t e = std::min<int64_t>(b + 16, n_pairs);
            for (int64_t i = b; i < e; ++i) {
                int32_t d, m, c;
                nw.run(queries[i], qlens[i], targets[i], tlens[i], d, m, c);
                if (dist) dist[i] = d;
                matches[i] = m;
                columns[i] = c;
            }
        }
    };
    if (nt == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; ++k) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    return SD_OK;
}


// Homopolymer compression (main.py:87-92).
static void homo_compress(const char* s, int64_t n, std::string& out) {
    out.clear();
    char prev = 0;
    for (int64_t i = 0; i < n; ++i)
        if (i == 0 || s[i] != prev) { out.push_back(s[i]); prev = s[i]; }
}

extern "C" int sd_identity_segments(const char* seq, int64_t seqlen, const int64_t* starts,
                                    const int64_t* ends, int64_t n_seg, const char* const* tmpl,
                                    const int32_t* tlen, int32_t T, const int32_t* pair_tmpl,
                                    int32_t homo, int32_t threads, int32_t* dist, int32_t* matches,
                                    int32_t* columns) {
    if (n_seg < 0 || T < 0 || !seq || (n_seg && (!starts || !ends)) || (T && (!tmpl || !tlen)) ||
        !matches || !columns)
        return SD_ERR_PARAM;
    for (int64_t s = 0; s < n_seg; ++s) {
        if (starts[s] < 0 || ends[s] >= seqlen || ends[s] - starts[s] + 1 > SD_NW_HOST_MAX) return SD_ERR_PARAM;
    }
    for (int t = 0; t < T; ++t)
        if (tlen[t] > SD_NW_HOST_MAX) return SD_ERR_UNSUPPORTED;
    if (pair_tmpl)
        for (int64_t s = 0; s < n_seg; ++s)
            if (pair_tmpl[s] < 0 || pair_tmpl[s] >= T) return SD_ERR_PARAM;
    std::vector<std::string> hm;
    if (homo) {
        hm.resize(T);
        for (int t = 0; t < T; ++t) homo_compress(tmpl[t], tlen[t], hm[t]);
    }
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_seg));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        BitNW nw;
        std::string hs;
        for (;;) {
            const int64_t s = next.fetch_add(1);
            if (s >= n_seg) break;
            const char* q = seq + starts[s];
            int qlen = (int)std::max<int64_t>(0, ends[s] - starts[s] + 1);
            if (homo) {
                homo_compress(q, qlen, hs);
                q = hs.data();
                qlen = (int)hs.size();
            }
            const int t0 = pair_tmpl ? pair_tmpl[s] : 0, t1 = pair_tmpl ? pair_tmpl[s] + 1 : T;
            for (int t = t0; t < t1; ++t) {
                int32_t d, m, c;
                if (homo) nw.run(q, qlen, hm[t].data(), (int)hm[t].size(), d, m, c);
                else nw.run(q, qlen, tmpl[t], tlen[t], d, m, c);
                const int64_t o = pair_tmpl ? s : s * T + t;
                if (dist) dist[o] = d;
                matches[o] = m;
                columns[o] = c;
            }
        }
    };
    if (nt == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; ++k) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    return SD_OK;
}


// Text of _alt.tsv rows (main.py:161-165) of one read or of a batch of reads (row_read = index of each
// block's read in read_names; NULL = all rows belong to read_names[0]): for every kept block, one line per monomer
// name:  read \t name \t start \t end \t "{:.2f}".format(identity) \t ('*' for the block's own monomer,
// '-' otherwise).  printf("%.2f") and Python's format both print the correctly rounded decimal of the
// double.  Multi-threaded over blocks.
extern "C" int sd_format_alt_rows(const char* const* read_names, int32_t n_reads, const int32_t* row_read,
                                  const char* const* key_names, int32_t n_keys, const int64_t* starts,
                                  const int64_t* ends, const int32_t* own_key, const double* vals,
                                  int64_t n_rows, int32_t threads, char** txt, size_t* txt_len) {
    if (!txt || !txt_len || !read_names || n_reads < 1 || n_rows < 0 || n_keys < 0 ||
        (n_rows && n_keys && (!key_names || !starts || !ends || !own_key || !vals)))
        return SD_ERR_PARAM;
    *txt = nullptr;
    *txt_len = 0;
    if (row_read)
        for (int64_t r = 0; r < n_rows; ++r)
            if (row_read[r] < 0 || row_read[r] >= n_reads) return SD_ERR_PARAM;
    std::vector<std::string> heads;
    for (int r = 0; r < n_reads; ++r) heads.push_back(std::string(read_names[r]) + "\t");
    std::vector<std::string> keys;
    for (int k = 0; k < n_keys; ++k) keys.emplace_back(key_names[k]);
    const int64_t grain = 256;
    const int64_t n_blocks = (n_rows + grain - 1) / grain;
    std::vector<std::string> parts((size_t)n_blocks);
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_blocks));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t b = next.fetch_add(1);
            if (b >= n_blocks) break;
            std::string& o = parts[(size_t)b];
            const int64_t r1 = std::min(n_rows, (b + 1) * grain);
            for (int64_t r = b * grain; r < r1; ++r) {
                char mid[64];
                const int ml = std::snprintf(mid, sizeof mid, "\t%lld\t%lld\t", (long long)starts[r], (long long)ends[r]);
                const std::string& head = heads[row_read ? (size_t)row_read[r] : 0];
                for (int k = 0; k < n_keys; ++k) {
                    o += head;
                    o += keys[(size_t)k];
                    o.append(mid, (size_t)ml);
                    sd::put_fixed2(o, vals[(size_t)r * n_keys + k]);
                    o += (k == own_key[r]) ? "\t*\n" : "\t-\n";
                }
            }
        }
    };
    if (nt == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; ++k) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    size_t total = 0;
    for (const std::string& p : parts) total += p.size();
    char* out = static_cast<char*>(std::malloc(total + 1));
    if (!out) return SD_ERR_PARAM;
    size_t pos = 0;
    for (const std::string& p : parts) { std::memcpy(out + pos, p.data(), p.size()); pos += p.size(); }
    out[total] = 0;
    *txt = out;
    *txt_len = total;
    return SD_OK;
}
