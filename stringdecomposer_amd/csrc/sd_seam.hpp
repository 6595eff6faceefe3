// sd_seam.hpp -- the seam merge (main.cpp:287-302) of ONE read whose chunks are spread over several ranks, without
// sending the rows anywhere (host only).
//
// The merge is a scan over the read's records b[0..N) with a single index as its state: at POSITION i it looks at
// b[i+1..i+6]; if b[i] covers more than half of some b[j] there, it keeps b[i], keeps b[j+1] unchecked and goes on at
// j + 2; otherwise it keeps b[i] and goes on at i + 1.  So a rank that holds b[lo..hi) can run the scan on its own
// records as soon as it knows (1) at which of its first eight records the scan enters and (2) the next rank's first
// eight records (a decision looks at most six records ahead, j + 2 lands at most eight ahead).  Both are small:
//
//   * every rank publishes an `sd_seam_edge` (include/sd_hip.h): start / end of its first and last eight records, and
//     for each of the eight possible entry positions the position at which the scan reaches its last eight records
//     (`exit_of`: eight scans that almost always join the first one after a few records);
//   * with all edges (one all-gather of ~170 bytes per rank) every rank follows the chain rank 0 -> itself through the
//     16-record windows [last eight of q | first eight of q + 1] and so learns its own entry position and the end of
//     the last record kept before it (SaveBatch prints start - previous end, main.cpp:279-283);
//   * rows and text are then made by each rank for its own records only (the record at index x is printed by the rank
//     that holds x), and the texts of the ranks, in rank order, are the text of the whole merge.
//
// Scans from different entry positions are compared by POSITION: two scans that reach the same position are the
// same scan from there on, and all records they keep from there on have indices >= that position (a jump from i
// keeps j + 1 and lands on j + 2), which is what lets a rank format its text from an assumed entry (position 0)
// while the edges travel and repair only the first rows afterwards.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/sd_hip.h"

namespace sd {

constexpr int SEAM_ZONE = 8;        // records of a piece that the neighbouring rank's scan can reach
constexpr int SEAM_MIN_PIECE = 32;  // a crossing piece shorter than this is not shared (the caller gathers instead)

inline bool seam_covers(const int32_t* bi, const int32_t* bj) {   // {start, end}: main.cpp:292
    return (bi[1] - bj[0]) * 2 > (bj[1] - bj[0]);
}
inline bool seam_covers(const sd_rec& bi, const sd_rec& bj) {
    return (bi.end - bj.start) * 2 > (bj.end - bj.start);
}

// One decision of the scan at position i over b[0..N): keep(x) for every record kept, returns the next position.
template <class Rows, class Keep>
inline size_t seam_step(const Rows& b, size_t N, size_t i, Keep&& keep) {
    const size_t lim = i + 7 < N ? i + 7 : N;
    for (size_t j = i + 1; j < lim; ++j)
        if (seam_covers(b[i], b[j])) {
            keep(i);
            if (j + 1 < N) keep(j + 1);
            return j + 2;
        }
    keep(i);
    return i + 1;
}

// A piece of a read on one rank: records b[0..n) of the read's record list (chunk offsets applied).
//   open_front: the read began on an earlier rank (the entry position is decided there)
//   open_back:  the read goes on on the next rank (the last eight positions are decided with its first records)
struct SeamPiece {
    const sd_rec* b = nullptr;
    size_t n = 0;
    bool open_front = false, open_back = false;
    // scan from position 0 ("assumed"): which indices are positions, what is kept, where it reaches the last zone
    std::vector<uint8_t> is_pos;
    std::vector<uint32_t> kept;     // indices kept by positions < exit0, ascending
    size_t exit0 = 0;               // open_back: first position >= n - SEAM_ZONE; else n
    size_t body_from = 0;           // kept[body_from..) is the part formatted ahead (positions >= body_pos)
    size_t body_pos = 0;            // a position of the assumed scan; everything from here on is final unless the real
    size_t body_guard = 0;          // scan joins later than body_guard (the position before body_pos)

    size_t stop() const { return open_back ? n - SEAM_ZONE : n; }

    void scan_assumed() {
        is_pos.assign(n + 1, 0);
        kept.clear();
        size_t i = 0;
        const size_t lim = stop();
        while (i < lim) {
            is_pos[i] = 1;
            i = seam_step(b, n, i, [&](size_t x) { kept.push_back((uint32_t)x); });
        }
        exit0 = i;
        body_from = 0;
        body_pos = 0;
        body_guard = 0;
        if (open_front) {
            // the text from a position >= 64 on is made ahead; positions before it wait for the real entry
            size_t pos = 0, prev = 0, k = 0;
            while (pos < lim && pos < 64) {
                prev = pos;
                pos = seam_step(b, n, pos, [&](size_t) { ++k; });
            }
            body_pos = pos < lim ? pos : exit0;
            body_guard = prev;
            if (pos >= lim) k = kept.size();
            body_from = k;
        }
    }
    // where the scan entering at position e reaches the last zone, as an offset 0..7 into it (open_back pieces)
    int exit_of(int e) const {
        size_t i = (size_t)e;
        const size_t lim = stop();
        while (i < lim && !is_pos[i]) i = seam_step(b, n, i, [](size_t) {});
        if (i < lim) i = exit0;
        return (int)(i - lim);
    }
    // The rows this rank prints BEFORE the part made ahead, for the real entry position e (row e - 1 was kept
    // unchecked by the previous rank's last jump).  Returns false when the real scan joins the assumed one too late
    // (then `all` holds every kept row of positions < stop() and `exit_pos` the real exit).
    bool head_rows(int e, std::vector<uint32_t>& head, std::vector<uint32_t>& all, size_t& exit_pos) const {
        head.clear();
        all.clear();
        if (e >= 1) head.push_back((uint32_t)(e - 1));
        size_t i = (size_t)e;
        const size_t lim = stop();
        while (i < lim && !is_pos[i]) i = seam_step(b, n, i, [&](size_t x) { head.push_back((uint32_t)x); });
        // joined the assumed scan at position i (or ran into the last zone / the end on its own)
        if (i <= body_guard) {
            for (size_t k = 0; k < body_from; ++k)
                if (kept[k] >= i) head.push_back(kept[k]);
            exit_pos = exit0;
            return true;
        }
        all = head;
        if (i < lim) {
            for (uint32_t x : kept)
                if (x >= i) all.push_back(x);
            exit_pos = exit0;
        } else {
            exit_pos = i;
        }
        return false;
    }
};

// The 16-record window between rank q and q + 1: from position x (0..7) of q's last zone.  Returns the entry
// position 0..7 on q + 1; keeps(idx) for the kept records of q's zone (idx 0..7); prev_end = end of the last kept record
// of q's zone.
template <class Keep>
inline int seam_window(const int32_t tail[8][2], const int32_t head[8][2], int x, int32_t& prev_end, Keep&& keep) {
    int32_t w[16][2];
    std::memcpy(w, tail, sizeof(int32_t) * 16);
    std::memcpy(w + 8, head, sizeof(int32_t) * 16);
    size_t i = (size_t)x;
    struct View { const int32_t (*w)[2]; const int32_t* operator[](size_t k) const { return w[k]; } } v{w};
    while (i < 8)
        i = seam_step(v, 16, i, [&](size_t k) {
            if (k < 8) { keep((int)k); prev_end = w[k][1]; }
        });
    return (int)i - 8;
}

struct SeamEntry { int e = 0; int32_t prev_end = 0; bool ok = true; };

// Entry of rank `rank` from the edges of all ranks.  ok = false: the edges do not describe a shareable job.
inline SeamEntry seam_resolve(const sd_seam_edge* edges, int world, int rank) {
    SeamEntry s;
    for (int q = 0; q < world; ++q)
        if (!edges[q].ok) { s.ok = false; return s; }
    if (edges[0].has_front || edges[world - 1].has_back) { s.ok = false; return s; }
    // the whole chain, whatever `rank` is: every rank reaches the same verdict about the same edges
    int e = 0;
    int32_t pe = 0;
    for (int q = 0; q + 1 < world; ++q) {
        if (q == rank) { s.e = e; s.prev_end = pe; }
        if (edges[q].through && !(edges[q].has_front && edges[q].has_back)) { s.ok = false; return s; }
        if (!edges[q].has_back) {
            if (edges[q + 1].has_front) { s.ok = false; return s; }
            e = 0;
            pe = 0;
            continue;
        }
        if (!edges[q + 1].has_front) { s.ok = false; return s; }
        const int x = edges[q].exit_of[edges[q].through ? e : 0];
        if (x < 0 || x > 7) { s.ok = false; return s; }
        e = seam_window(edges[q].tail, edges[q + 1].head, x, pe, [](int) {});
    }
    if (rank == world - 1) { s.e = e; s.prev_end = pe; }
    return s;
}

}  // namespace sd
