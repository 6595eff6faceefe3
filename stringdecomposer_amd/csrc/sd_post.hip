// sd_post.hip -- host-side post-processing kernel of the drop-in CLI: unit-cost global alignment
// identity of read segments against templates (what stringdecomposer/main.py:29-60 gets from
// python-edlib).  Host code only (compiled by hipcc for a uniform build); multi-threaded over pairs.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/sd_hip.h"

namespace {

// Full-matrix NW (uint16 cells suffice: distance <= max(qlen, tlen) < 65536), traceback from the
// bottom-right corner: up ('I') > left ('D') > diagonal ('=' / 'X').
void nw_identity(const char* q, int qlen, const char* t, int tlen, std::vector<uint16_t>& D,
                 int32_t& dist, int32_t& matches, int32_t& columns) {
    dist = -1;
    matches = 0;
    columns = 0;
    if (qlen <= 0 || tlen <= 0) return;
    const size_t W = (size_t)tlen + 1;
    D.resize((size_t)(qlen + 1) * W);
    for (int c = 0; c <= tlen; ++c) D[(size_t)c] = (uint16_t)c;
    for (int r = 1; r <= qlen; ++r) {
        uint16_t* cur = &D[(size_t)r * W];
        const uint16_t* up = cur - W;
        cur[0] = (uint16_t)r;
        const char qc = q[r - 1];
        // diagonal/up terms vectorise; the left dependency is a cheap second pass
        for (int c = 1; c <= tlen; ++c) {
            const uint16_t d = (uint16_t)(up[c - 1] + (qc == t[c - 1] ? 0 : 1));
            const uint16_t u = (uint16_t)(up[c] + 1);
            cur[c] = d < u ? d : u;
        }
        for (int c = 1; c <= tlen; ++c) {
            const uint16_t l = (uint16_t)(cur[c - 1] + 1);
            if (l < cur[c]) cur[c] = l;
        }
    }
    int r = qlen, c = tlen, m = 0, cols = 0;
    while (r > 0 || c > 0) {
        const uint16_t cur = D[(size_t)r * W + c];
        if (r > 0 && (uint16_t)(D[(size_t)(r - 1) * W + c] + 1) == cur) { --r; }
        else if (c > 0 && (uint16_t)(D[(size_t)r * W + c - 1] + 1) == cur) { --c; }
        else { if (D[(size_t)(r - 1) * W + c - 1] == cur) ++m; --r; --c; }
        ++cols;
    }
    dist = D[(size_t)qlen * W + tlen];
    matches = m;
    columns = cols;
}

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
        std::vector<uint16_t> D;
        for (;;) {
            const int64_t b = next.fetch_add(16);
            if (b >= n_pairs) break;
            const int64_t e = std::min<int64_t>(b + 16, n_pairs);
            for (int64_t i = b; i < e; ++i) {
                int32_t d, m, c;
                nw_identity(queries[i], qlens[i], targets[i], tlens[i], D, d, m, c);
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
