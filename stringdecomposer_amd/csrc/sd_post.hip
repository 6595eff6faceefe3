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

// Unit-cost global alignment by Myers' bit-vector algorithm (J. ACM 46(3), 1999; block form of
// Hyyro 2003): query along the bit rows, one column per target symbol, the vertical delta vectors
// (Pv, Mv) of every column are kept so that any D[i][j] = j + popcount(Pv_j & low_i) -
// popcount(Mv_j & low_i) can be read back.  The traceback then applies the same rule as a
// full-matrix walk from the bottom-right corner: up ('I', consume query) > left ('D', consume
// target) > diagonal ('=' / 'X').  Alignment columns = edit distance + matches.
struct BitNW {
    std::vector<uint64_t> peq;  // [5][K]
    std::vector<uint64_t> pv, mv;  // [(tlen + 1)][K]
    static int code(char c) {
        switch (c) {
            case 'A': return 0;
            case 'C': return 1;
            case 'G': return 2;
            case 'T': return 3;
            case 'N': return 4;
            default: return 5;
        }
    }
    void run(const char* q, int qlen, const char* t, int tlen, int32_t& dist, int32_t& matches,
             int32_t& columns) {
        dist = -1;
        matches = 0;
        columns = 0;
        if (qlen <= 0 || tlen <= 0) return;
        const int K = (qlen + 63) >> 6;
        // symbols outside ACGTN can only match themselves: handled by a slow exact path
        for (int i = 0; i < qlen; ++i)
            if (code(q[i]) > 4) { slow(q, qlen, t, tlen, dist, matches, columns); return; }
        for (int j = 0; j < tlen; ++j)
            if (code(t[j]) > 4) { slow(q, qlen, t, tlen, dist, matches, columns); return; }
        peq.assign((size_t)5 * K, 0);
        for (int i = 0; i < qlen; ++i) peq[(size_t)code(q[i]) * K + (i >> 6)] |= 1ull << (i & 63);
        pv.resize((size_t)(tlen + 1) * K);
        mv.resize((size_t)(tlen + 1) * K);
        for (int b = 0; b < K; ++b) { pv[b] = ~0ull; mv[b] = 0; }
        for (int j = 1; j <= tlen; ++j) {
            const uint64_t* eqs = &peq[(size_t)code(t[j - 1]) * K];
            const uint64_t* pp = &pv[(size_t)(j - 1) * K];
            const uint64_t* pm = &mv[(size_t)(j - 1) * K];
            uint64_t* np = &pv[(size_t)j * K];
            uint64_t* nm = &mv[(size_t)j * K];
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
        dist = cur;
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
        matches = m;
        columns = dist + m;
    }
    // full-matrix version for sequences with symbols outside ACGTN (never produced by the pipeline)
    std::vector<uint16_t> Dm;
    void slow(const char* q, int qlen, const char* t, int tlen, int32_t& dist, int32_t& matches,
              int32_t& columns) {
        const size_t W = (size_t)tlen + 1;
        Dm.resize((size_t)(qlen + 1) * W);
        for (int c = 0; c <= tlen; ++c) Dm[(size_t)c] = (uint16_t)c;
        for (int r = 1; r <= qlen; ++r) {
            uint16_t* cur = &Dm[(size_t)r * W];
            const uint16_t* up = cur - W;
            cur[0] = (uint16_t)r;
            for (int c = 1; c <= tlen; ++c) {
                uint16_t v = (uint16_t)(up[c - 1] + (q[r - 1] == t[c - 1] ? 0 : 1));
                if ((uint16_t)(up[c] + 1) < v) v = (uint16_t)(up[c] + 1);
                if ((uint16_t)(cur[c - 1] + 1) < v) v = (uint16_t)(cur[c - 1] + 1);
                cur[c] = v;
            }
        }
        int r = qlen, c = tlen, m = 0, cols = 0;
        while (r > 0 || c > 0) {
            const uint16_t cur = Dm[(size_t)r * W + c];
            if (r > 0 && (uint16_t)(Dm[(size_t)(r - 1) * W + c] + 1) == cur) { --r; }
            else if (c > 0 && (uint16_t)(Dm[(size_t)r * W + c - 1] + 1) == cur) { --c; }
            else { if (Dm[(size_t)(r - 1) * W + c - 1] == cur) ++m; --r; --c; }
            ++cols;
        }
        dist = Dm[(size_t)qlen * W + tlen];
        matches = m;
        columns = cols;
    }
};

}  // namespace

extern "C" int sd_nw_identity_batch(const char* const* queries, const int32_t* qlens,
                                    const char* const* targets, const int32_t* tlens,
                                    int64_t n_pairs, int32_t threads, int32_t* dist,
                                    int32_t* matches, int32_t* columns) {
    if (n_pairs < 0 || !queries || !targets || !qlens || !tlens || !matches || !columns) return SD_ERR_PARAM;
    for (int64_t i = 0; i < n_pairs; ++i)
        if (qlens[i] > 65000 || tlens[i] > 65000) return SD_ERR_UNSUPPORTED;
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
        if (starts[s] < 0 || ends[s] >= seqlen || ends[s] - starts[s] + 1 > 65000) return SD_ERR_PARAM;
    }
    for (int t = 0; t < T; ++t)
        if (tlen[t] > 65000) return SD_ERR_UNSUPPORTED;
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
