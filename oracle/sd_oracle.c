/*
 * sd_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see sd_oracle.h).
 *
 * Plain-C restatement of ablab/stringdecomposer v1.1.2, stringdecomposer/src/main.cpp.
 * Written from the behaviour of that file (citations are to its lines); deliberately naive:
 * it materialises the whole dp[i][j][k] matrix like the reference does and walks it with the
 * reference's own equality tests, so that tie-breaking is identical by construction.
 * Parity pinned against oracle/_ref/dp (the real reference binary) -- see tests/test_oracle.py.
 */
#define _GNU_SOURCE
#include "sd_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SDO_INF (-1000000) /* main.cpp:156 */

void sdo_free(void* p) { free(p); }

/* ------------------------------------------------------------------------------------------ */
/* AlignPartClassicDP, main.cpp:151-270                                                        */
/* ------------------------------------------------------------------------------------------ */
int sdo_align_chunk(const char* read, int n, const char* const* tmpl, const int* tlen, int T,
                    sdo_scoring sc, sdo_rec* out, int cap, int32_t* Bout, int32_t* argBout,
                    int64_t* ptr_hist) {
    const int ins = sc.ins, del = sc.del, match = sc.match, mismatch = sc.mismatch;
    const int INF = SDO_INF;
    if (n <= 0 || T <= 0) return -1;

    /* flat layout: row i has sumL template cells followed by the "between monomers" slot
     * dp[i][T][0] (main.cpp:166-167). */
    int64_t* toff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(T + 1));
    toff[0] = 0;
    for (int j = 0; j < T; ++j) toff[j + 1] = toff[j] + tlen[j];
    const int64_t sumL = toff[T];
    const int64_t W = sumL + 1;
    int32_t* dp = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * (size_t)n);
    if (!dp) { free(toff); return -1; }
#define DP(i, j, k) dp[(int64_t)(i) * W + toff[(j)] + (k)]
#define BT(i) dp[(int64_t)(i) * W + sumL]
    /* main.cpp:158-169: everything starts at INF */
    for (int64_t q = 0; q < W * (int64_t)n; ++q) dp[q] = INF;

    /* row 0, main.cpp:171-182 */
    for (int j = 0; j < T; ++j) {
        DP(0, j, 0) = (tmpl[j][0] == read[0]) ? match : mismatch;
        for (int k = 1; k < tlen[j]; ++k) {
            int mm = (tmpl[j][k] == read[0]) ? match : mismatch;
            int a = DP(0, j, k - 1) + del;
            int b = del * (k - 1) + mm; /* note k-1, main.cpp:180 */
            DP(0, j, k) = a > b ? a : b;
        }
    }
    /* rows 1..n-1, main.cpp:183-208 */
    for (int i = 1; i < n; ++i) {
        for (int j = 0; j < T; ++j) {
            int v = DP(i - 1, j, tlen[j] - 1);
            if (v > BT(i)) BT(i) = v;
        }
        const int Bi = BT(i);
        for (int j = 0; j < T; ++j) {
            for (int k = 0; k < tlen[j]; ++k) {
                int score = INF;
                int mm = (tmpl[j][k] == read[i]) ? match : mismatch;
                if (Bi > INF) {
                    int c = Bi + mm + k * del;
                    if (c > score) score = c;
                }
                if (k > 0) {
                    if (DP(i - 1, j, k - 1) > INF) {
                        int c = DP(i - 1, j, k - 1) + mm;
                        if (c > score) score = c;
                    }
                    if (DP(i - 1, j, k) > INF) {
                        int c = DP(i - 1, j, k) + ins;
                        if (c > score) score = c;
                    }
                    if (DP(i, j, k - 1) > INF) {
                        int c = DP(i, j, k - 1) + del;
                        if (c > score) score = c;
                    }
                }
                DP(i, j, k) = score;
            }
        }
    }
    /* main.cpp:209-216: first strict maximum over template ends of the last row */
    int max_score = INF, best_m = T;
    for (int j = 0; j < T; ++j) {
        int v = DP(n - 1, j, tlen[j] - 1);
        if (max_score < v) { max_score = v; best_m = j; }
    }

    if (Bout) {
        Bout[0] = INF;
        for (int i = 1; i < n; ++i) Bout[i] = BT(i);
        Bout[n] = max_score;
    }
    if (argBout) {
        argBout[0] = -1;
        for (int i = 1; i < n; ++i) {
            int a = -1;
            for (int p = 0; p < T; ++p)
                if (DP(i - 1, p, tlen[p] - 1) == BT(i)) { a = p; break; }
            argBout[i] = a;
        }
        argBout[n] = best_m;
    }
    if (ptr_hist) {
        ptr_hist[0] = ptr_hist[1] = ptr_hist[2] = ptr_hist[3] = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < T; ++j)
                for (int k = 0; k < tlen[j]; ++k) {
                    int v = DP(i, j, k), code;
                    int mm = (tmpl[j][k] == read[i]) ? match : mismatch;
                    if (k != 0 && v == DP(i, j, k - 1) + del) code = 0;
                    else if (i != 0 && v == DP(i - 1, j, k) + ins) code = 1;
                    else if (i != 0 && k != 0 && v == DP(i - 1, j, k - 1) + mm) code = 2;
                    else code = 3;
                    ptr_hist[code]++;
                }
    }

    /* traceback, main.cpp:217-269 */
    int cnt = 0, overflow = 0;
    if (best_m == T) { /* unreachable for finite inputs; the reference would index out of range */
        free(dp); free(toff);
        return -1;
    }
    long long i = n - 1, j = best_m, k = tlen[best_m] - 1;
    int changed = 1;
    sdo_rec cur;
    memset(&cur, 0, sizeof cur);
    while (i >= 0) {
        if (j != T && k == tlen[j] - 1 && changed) { /* main.cpp:224-227 */
            cur.tmpl = (int32_t)j;
            cur.start = (int32_t)i;
            cur.end = (int32_t)i;
            cur.score = (float)DP(i, j, k);
            changed = 0;
        }
        if (j == T) { /* main.cpp:228-240 */
            if (i != 0) {
                int moved = 0;
                for (int p = 0; p < T; ++p) {
                    if (DP(i - 1, p, tlen[p] - 1) == BT(i)) {
                        --i; j = p; k = tlen[p] - 1; moved = 1;
                        break;
                    }
                }
                if (!moved) break; /* cannot happen: B_i is the max of exactly these cells */
            } else {
                --i;
            }
        } else {
            int v = DP(i, j, k);
            if (k != 0 && v == DP(i, j, k - 1) + del) { /* main.cpp:242 */
                --k;
            } else if (i != 0 && v == DP(i - 1, j, k) + ins) { /* main.cpp:245 */
                --i;
            } else {
                int mm = (tmpl[j][k] == read[i]) ? match : mismatch;
                if (i != 0 && k != 0 && v == DP(i - 1, j, k - 1) + mm) { /* main.cpp:249 */
                    --i; --k;
                } else {
                    changed = 1;
                    if (i != 0 && BT(i) + k * del + mm == v) { /* main.cpp:253 */
                        cur.start = (int32_t)i;
                        cur.score = cur.score - (float)BT(i);
                        if (cnt < cap) out[cnt] = cur; else overflow = 1;
                        ++cnt;
                        j = T; k = 0;
                    } else { /* main.cpp:258-262 */
                        cur.start = (int32_t)i;
                        if (cnt < cap) out[cnt] = cur; else overflow = 1;
                        ++cnt;
                        --i;
                    }
                }
            }
        }
    }
#undef DP
#undef BT
    free(dp);
    free(toff);
    if (overflow) return -1;
    /* reverse, main.cpp:268 */
    for (int a = 0, b = cnt - 1; a < b; ++a, --b) {
        sdo_rec t = out[a]; out[a] = out[b]; out[b] = t;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* PostProcessing, main.cpp:287-302 (literal: the element after a dropped run is appended       */
/* without being compared with its successors)                                                  */
/* ------------------------------------------------------------------------------------------ */
int sdo_postprocess(sdo_rec* b, int n) {
    sdo_rec* res = (sdo_rec*)malloc(sizeof(sdo_rec) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    size_t i = 0, N = (size_t)n;
    while (i < N) {
        size_t lim = i + 7 < N ? i + 7 : N;
        for (size_t j = i + 1; j < lim; ++j) {
            if ((b[i].end - b[j].start) * 2 > (b[j].end - b[j].start)) {
                res[m++] = b[i];
                i = j + 1;
                break;
            }
        }
        if (i < N) res[m++] = b[i];
        ++i;
    }
    memcpy(b, res, sizeof(sdo_rec) * (size_t)m);
    free(res);
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* chunking, main.cpp:70-81                                                                     */
/* ------------------------------------------------------------------------------------------ */
int sdo_chunk_plan(int64_t len, int part, int overlap, int64_t* off, int32_t* clen, int cap) {
    int cnt = 0;
    for (int64_t i = 0; i < len; i += part) {
        if (len - i >= overlap || len < overlap) {
            int64_t l = len - i;
            if ((int64_t)part + overlap < l) l = (int64_t)part + overlap;
            if (cnt < cap) { off[cnt] = i; clen[cnt] = (int32_t)l; }
            ++cnt;
        }
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* load_fasta, main.cpp:314-346                                                                 */
/* ------------------------------------------------------------------------------------------ */
void sdo_free_fasta(sdo_fasta* f) {
    if (!f) return;
    for (int i = 0; i < f->n; ++i) { free(f->names[i]); free(f->seqs[i]); }
    free(f->names); free(f->seqs); free(f->lens);
    memset(f, 0, sizeof *f);
}

static void set_err(char* err, size_t errlen, const char* msg) {
    if (err && errlen) { snprintf(err, errlen, "%s", msg); }
}

int sdo_load_fasta(const char* path, sdo_fasta* out, char* err, size_t errlen) {
    memset(out, 0, sizeof *out);
    FILE* fp = fopen(path, "rb");
    if (!fp) {
        /* the reference's ifstream silently yields zero sequences; report it instead */
        char m[512]; snprintf(m, sizeof m, "cannot open %s", path);
        set_err(err, errlen, m);
        return 2;
    }
    int cap = 0;
    size_t* caps = NULL;
    char* line = NULL;
    size_t lcap = 0;
    ssize_t got;
    int rc = 0;
    while ((got = getline(&line, &lcap, fp)) >= 0) {
        size_t L = (size_t)got;
        if (L && line[L - 1] == '\n') line[--L] = 0; /* std::getline strips only '\n' */
        if (L > 0 && line[0] == '>') {
            /* name = first whitespace-delimited token of the header (main.cpp:321-325) */
            size_t a = 1;
            while (a < L && (line[a] == ' ' || line[a] == '\t' || line[a] == '\r' || line[a] == '\v' || line[a] == '\f')) ++a;
            size_t b = a;
            while (b < L && !(line[b] == ' ' || line[b] == '\t' || line[b] == '\r' || line[b] == '\v' || line[b] == '\f')) ++b;
            if (b == a) { set_err(err, errlen, "FASTA header without a name"); rc = 3; break; }
            if (out->n == cap) {
                cap = cap ? cap * 2 : 16;
                out->names = (char**)realloc(out->names, sizeof(char*) * (size_t)cap);
                out->seqs = (char**)realloc(out->seqs, sizeof(char*) * (size_t)cap);
                out->lens = (int64_t*)realloc(out->lens, sizeof(int64_t) * (size_t)cap);
                caps = (size_t*)realloc(caps, sizeof(size_t) * (size_t)cap);
            }
            out->names[out->n] = strndup(line + a, b - a);
            out->seqs[out->n] = (char*)malloc(16);
            out->seqs[out->n][0] = 0;
            out->lens[out->n] = 0;
            caps[out->n] = 16;
            out->n++;
        } else {
            if (out->n == 0) {
                if (L == 0) continue; /* s[0] of an empty string is '\0' in the reference: no-op... */
                set_err(err, errlen, "FASTA does not start with a header"); rc = 3; break;
            }
            int q = out->n - 1; /* sequence lines are appended verbatim (main.cpp:327) */
            size_t need = (size_t)out->lens[q] + L + 1;
            if (need > caps[q]) {
                while (caps[q] < need) caps[q] *= 2;
                out->seqs[q] = (char*)realloc(out->seqs[q], caps[q]);
            }
            memcpy(out->seqs[q] + out->lens[q], line, L);
            out->lens[q] += (int64_t)L;
            out->seqs[q][out->lens[q]] = 0;
        }
    }
    free(line);
    free(caps);
    fclose(fp);
    if (rc) { sdo_free_fasta(out); return rc; }
    /* alphabet check, main.cpp:329-341 */
    for (int q = 0; q < out->n; ++q) {
        for (int64_t p = 0; p < out->lens[q]; ++p) {
            char c = out->seqs[q][p];
            if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N')) {
                char m[1024];
                snprintf(m, sizeof m, "ERROR: Sequence %s contains undefined symbol (not ACGT): %c",
                         out->names[q], c);
                set_err(err, errlen, m);
                sdo_free_fasta(out);
                return 255; /* exit(-1) */
            }
            if (c == 'N') out->has_n = 1;
        }
    }
    return 0;
}

/* reverse_complement, main.cpp:348-362 */
int sdo_reverse_complement(const char* s, int64_t n, char* dst) {
    for (int64_t i = 0; i < n; ++i) {
        char c = s[n - 1 - i], r;
        switch (c) {
            case 'A': r = 'T'; break;
            case 'T': r = 'A'; break;
            case 'G': r = 'C'; break;
            case 'C': r = 'G'; break;
            case 'N': r = 'N'; break;
            default: return -1;
        }
        dst[i] = r;
    }
    dst[n] = 0;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* AlignReadsSet (main.cpp:67-122) + SaveBatch (main.cpp:272-285)                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { char* p; size_t len, cap; } sbuf;
static void sb_put(sbuf* s, const char* t, size_t n) {
    if (s->len + n + 1 > s->cap) {
        while (s->len + n + 1 > s->cap) s->cap = s->cap ? s->cap * 2 : 1 << 16;
        s->p = (char*)realloc(s->p, s->cap);
    }
    memcpy(s->p + s->len, t, n);
    s->len += n;
    s->p[s->len] = 0;
}

/* MonomerEditDistance (main.cpp:128-133): edlib HW ("infix") mode = unit-cost edit distance of the
 * whole template against the best-matching substring of the chunk (free gaps before and after
 * the match in the chunk).  Plain DP restatement of that definition. */
int sdo_hw_edit_distance(const char* tmpl, int m, const char* text, int n) {
    int* col = (int*)malloc(sizeof(int) * (size_t)(m + 1));
    for (int i = 0; i <= m; ++i) col[i] = i;
    int best = col[m];
    for (int j = 1; j <= n; ++j) {
        int diag = col[0]; /* D[0][j-1] = 0 */
        col[0] = 0;
        for (int i = 1; i <= m; ++i) {
            int up = col[i - 1], left = col[i];
            int v = diag + (tmpl[i - 1] == text[j - 1] ? 0 : 1);
            if (up + 1 < v) v = up + 1;
            if (left + 1 < v) v = left + 1;
            diag = left;
            col[i] = v;
        }
        if (col[m] < best) best = col[m];
    }
    free(col);
    return best;
}

/* FilterMonomersForRead (main.cpp:135-149): order the templates by (distance, index), keep the
 * first one and every other one with distance <= ed_thr, in that order.  Returns the count;
 * keep[] receives the original template indices in filtered order. */
int sdo_filter_templates(const char* chunk, int n, const char* const* tmpl, const int* tlen, int T,
                         int ed_thr, int* keep) {
    int* dist = (int*)malloc(sizeof(int) * (size_t)T);
    int* ord = (int*)malloc(sizeof(int) * (size_t)T);
    for (int j = 0; j < T; ++j) { dist[j] = sdo_hw_edit_distance(tmpl[j], tlen[j], chunk, n); ord[j] = j; }
    for (int a = 1; a < T; ++a) { /* insertion sort by (dist, index); indices start ordered */
        int x = ord[a], b = a - 1;
        while (b >= 0 && dist[ord[b]] > dist[x]) { ord[b + 1] = ord[b]; --b; }
        ord[b + 1] = x;
    }
    int cnt = 0;
    keep[cnt++] = ord[0];
    for (int a = 1; a < T; ++a)
        if (dist[ord[a]] <= ed_thr) keep[cnt++] = ord[a];
    free(dist);
    free(ord);
    return cnt;
}

int sdo_decompose_ex(const char* const* read_names, const char* const* read_seqs, int n_reads,
                     const char* const* mono_names, const char* const* mono_seqs, int n_mono,
                     int threads, int part_size, int overlap, sdo_scoring sc, int ed_thr, char** tsv,
                     size_t* tsv_len, char* err, size_t errlen) {
    *tsv = NULL; *tsv_len = 0;
    if (part_size <= 0 || overlap < 0 || n_mono <= 0) {
        set_err(err, errlen, "bad parameters");
        return 4;
    }
    /* add_reverse_complement, main.cpp:364-371 */
    const int T = 2 * n_mono;
    char** tn = (char**)calloc((size_t)T, sizeof(char*));
    char** ts = (char**)calloc((size_t)T, sizeof(char*));
    int* tl = (int*)calloc((size_t)T, sizeof(int));
    int rc = 0;
    for (int j = 0; j < n_mono; ++j) {
        size_t L = strlen(mono_seqs[j]);
        if (L == 0) { set_err(err, errlen, "empty monomer sequence"); rc = 5; }
        tn[j] = strdup(mono_names[j]);
        ts[j] = strdup(mono_seqs[j]);
        tl[j] = (int)L;
        size_t nl = strlen(mono_names[j]);
        tn[n_mono + j] = (char*)malloc(nl + 2);
        memcpy(tn[n_mono + j], mono_names[j], nl);
        tn[n_mono + j][nl] = '\'';
        tn[n_mono + j][nl + 1] = 0;
        ts[n_mono + j] = (char*)malloc(L + 1);
        tl[n_mono + j] = (int)L;
        if (sdo_reverse_complement(mono_seqs[j], (int64_t)L, ts[n_mono + j]) != 0) {
            set_err(err, errlen, "map::at"); rc = 255;
        }
    }
    /* chunk table, main.cpp:70-81 */
    int64_t total_chunks = 0;
    int* nch = (int*)calloc((size_t)(n_reads > 0 ? n_reads : 1), sizeof(int));
    for (int r = 0; r < n_reads && !rc; ++r) {
        int64_t len = (int64_t)strlen(read_seqs[r]);
        nch[r] = sdo_chunk_plan(len, part_size, overlap, NULL, NULL, 0);
        if (nch[r] == 0) { /* the reference dereferences batch[0] of an empty batch (main.cpp:115) */
            char m[1024];
            snprintf(m, sizeof m, "ERROR: Sequence %s is empty", read_names[r]);
            set_err(err, errlen, m);
            rc = 6;
        }
        total_chunks += nch[r];
    }
    int64_t* c_off = NULL; int32_t* c_len = NULL; int* c_read = NULL;
    sdo_rec** c_recs = NULL; int* c_cnt = NULL;
    if (!rc) {
        size_t C = (size_t)(total_chunks > 0 ? total_chunks : 1);
        c_off = (int64_t*)malloc(sizeof(int64_t) * C);
        c_len = (int32_t*)malloc(sizeof(int32_t) * C);
        c_read = (int*)malloc(sizeof(int) * C);
        c_recs = (sdo_rec**)calloc(C, sizeof(sdo_rec*));
        c_cnt = (int*)calloc(C, sizeof(int));
        int64_t q = 0;
        for (int r = 0; r < n_reads; ++r) {
            int64_t len = (int64_t)strlen(read_seqs[r]);
            sdo_chunk_plan(len, part_size, overlap, c_off + q, c_len + q, nch[r]);
            for (int a = 0; a < nch[r]; ++a) c_read[q + a] = r;
            q += nch[r];
        }
        int bad = 0;
#ifdef _OPENMP
        if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
        for (int64_t c = 0; c < total_chunks; ++c) {
            int n = c_len[c];
            sdo_rec* rr = (sdo_rec*)malloc(sizeof(sdo_rec) * (size_t)n);
            int k;
            if (ed_thr > -1) { /* main.cpp:91-93 */
                int* keep = (int*)malloc(sizeof(int) * (size_t)T);
                int nk = sdo_filter_templates(read_seqs[c_read[c]] + c_off[c], n, (const char* const*)ts,
                                              tl, T, ed_thr, keep);
                const char** fs = (const char**)malloc(sizeof(char*) * (size_t)nk);
                int* fl = (int*)malloc(sizeof(int) * (size_t)nk);
                for (int a = 0; a < nk; ++a) { fs[a] = ts[keep[a]]; fl[a] = tl[keep[a]]; }
                k = sdo_align_chunk(read_seqs[c_read[c]] + c_off[c], n, fs, fl, nk, sc, rr, n, NULL,
                                    NULL, NULL);
                for (int a = 0; a < k; ++a) rr[a].tmpl = keep[rr[a].tmpl];
                free(keep); free(fs); free(fl);
            } else {
                k = sdo_align_chunk(read_seqs[c_read[c]] + c_off[c], n, (const char* const*)ts, tl,
                                    T, sc, rr, n, NULL, NULL, NULL);
            }
            if (k < 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                bad = 1;
                k = 0;
            }
            c_recs[c] = rr;
            c_cnt[c] = k;
        }
        if (bad) { set_err(err, errlen, "internal: chunk alignment failed"); rc = 7; }
    }
    (void)threads;
    sbuf sb = {0, 0, 0};
    if (!rc) {
        /* per-read flush, main.cpp:104-120 + SaveBatch 272-285 */
        int64_t q = 0;
        for (int r = 0; r < n_reads; ++r) {
            int tot = 0;
            for (int a = 0; a < nch[r]; ++a) tot += c_cnt[q + a];
            sdo_rec* batch = (sdo_rec*)malloc(sizeof(sdo_rec) * (size_t)(tot > 0 ? tot : 1));
            int m = 0;
            for (int a = 0; a < nch[r]; ++a) {
                for (int x = 0; x < c_cnt[q + a]; ++x) {
                    sdo_rec t = c_recs[q + a][x];
                    t.start += (int32_t)c_off[q + a]; /* main.cpp:109-111 */
                    t.end += (int32_t)c_off[q + a];
                    batch[m++] = t;
                }
            }
            m = sdo_postprocess(batch, m);
            int prev_end = 0;
            char num[128];
            for (int x = 0; x < m; ++x) {
                sb_put(&sb, read_names[r], strlen(read_names[r]));
                sb_put(&sb, "\t", 1);
                sb_put(&sb, tn[batch[x].tmpl], strlen(tn[batch[x].tmpl]));
                /* std::to_string(int) == "%d", std::to_string(float) == "%f" */
                int w = snprintf(num, sizeof num, "\t%d\t%d\t%f\t%d\t%d\n", batch[x].start,
                                 batch[x].end, (double)batch[x].score, batch[x].start - prev_end,
                                 batch[x].end - batch[x].start);
                sb_put(&sb, num, (size_t)w);
                prev_end = batch[x].end;
            }
            free(batch);
            q += nch[r];
        }
    }
    if (c_recs) for (int64_t c = 0; c < total_chunks; ++c) free(c_recs[c]);
    free(c_recs); free(c_cnt); free(c_off); free(c_len); free(c_read); free(nch);
    for (int j = 0; j < T; ++j) { free(tn[j]); free(ts[j]); }
    free(tn); free(ts); free(tl);
    if (rc) { free(sb.p); return rc; }
    if (!sb.p) { sb.p = (char*)malloc(1); sb.p[0] = 0; }
    *tsv = sb.p;
    *tsv_len = sb.len;
    return 0;
}

int sdo_decompose(const char* const* read_names, const char* const* read_seqs, int n_reads,
                  const char* const* mono_names, const char* const* mono_seqs, int n_mono,
                  int threads, int part_size, int overlap, sdo_scoring sc, char** tsv,
                  size_t* tsv_len, char* err, size_t errlen) {
    return sdo_decompose_ex(read_names, read_seqs, n_reads, mono_names, mono_seqs, n_mono, threads,
                            part_size, overlap, sc, -1, tsv, tsv_len, err, errlen);
}

int sdo_decompose_files(const char* reads_fa, const char* monomers_fa, int threads, int part_size,
                        int overlap, sdo_scoring sc, char** tsv, size_t* tsv_len, char* err,
                        size_t errlen) {
    return sdo_decompose_files_ex(reads_fa, monomers_fa, threads, part_size, overlap, sc, -1, tsv,
                                  tsv_len, err, errlen);
}

int sdo_decompose_files_ex(const char* reads_fa, const char* monomers_fa, int threads, int part_size,
                           int overlap, sdo_scoring sc, int ed_thr, char** tsv, size_t* tsv_len,
                           char* err, size_t errlen) {
    sdo_fasta R, M;
    int rc = sdo_load_fasta(reads_fa, &R, err, errlen);
    if (rc) return rc;
    rc = sdo_load_fasta(monomers_fa, &M, err, errlen);
    if (rc) { sdo_free_fasta(&R); return rc; }
    rc = sdo_decompose_ex((const char* const*)R.names, (const char* const*)R.seqs, R.n,
                          (const char* const*)M.names, (const char* const*)M.seqs, M.n, threads,
                          part_size, overlap, sc, ed_thr, tsv, tsv_len, err, errlen);
    sdo_free_fasta(&R);
    sdo_free_fasta(&M);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Unit-cost NW identity (restates what main.py:29-60 obtains from edlib NW + extended CIGAR)   */
/* ------------------------------------------------------------------------------------------ */
/* '=' and all columns of the path edlib reports for one (sub)problem whose distance is known to be the optimum:
 * edlib.cpp:1164-1213 (obtainAlignment).  Small problems: block traceback, priority up ('I') > left ('D') > diagonal
 * (edlib.cpp:945-1130, restated on the full matrix).  Once (2*8+4) * ceil(qlen/64) * tlen + 8 * tlen reaches 1 MB
 * edlib splits the TARGET in halves (Hirschberg, edlib.cpp:1234-1400): with L[i] = distance of query[0..i) to the
 * left half and R[i] = distance of query[i..) to the right half it takes the SMALLEST 0-based row index x in
 * 0..qlen-2 with L[x+1] + R[x+1] == best, else x = -1 if lw + R[0] == best, else x = qlen-1 if L[qlen] + rw == best
 * (edlib.cpp:1315-1349; its banded columns hold every cell of an optimal path exactly), and concatenates the paths
 * of (query[0..x], left half) and (query[x+1..], right half). */
static void nw_column(const char* q, int qlen, const char* t, int tlen, int rev, int32_t* out) {
    /* out[i] = distance of the first i symbols of q to the first tlen symbols of t (rev: both read from the end) */
    int32_t* prev = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    for (int i = 0; i <= qlen; ++i) out[i] = i;
    for (int c = 1; c <= tlen; ++c) {
        memcpy(prev, out, sizeof(int32_t) * (size_t)(qlen + 1));
        out[0] = c;
        const char tc = rev ? t[-(c - 1)] : t[c - 1];
        for (int i = 1; i <= qlen; ++i) {
            const char qc = rev ? q[-(i - 1)] : q[i - 1];
            int d = prev[i - 1] + (qc == tc ? 0 : 1);
            int u = out[i - 1] + 1;
            int l = prev[i] + 1;
            int m = d < u ? d : u;
            out[i] = m < l ? m : l;
        }
    }
    free(prev);
}

static int nw_path(const char* q, int qlen, const char* t, int tlen, int best, int* matches, int* columns) {
    if (qlen == 0 || tlen == 0) { *columns += qlen + tlen; return 0; }
    const long long blocks = (qlen + 63) / 64;
    if (20ll * blocks * tlen + 8ll * tlen < 1024 * 1024) {
        const int W = tlen + 1;
        int32_t* D = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1) * (size_t)W);
        for (int c = 0; c <= tlen; ++c) D[c] = c;
        for (int r = 1; r <= qlen; ++r) {
            D[(size_t)r * W] = r;
            for (int c = 1; c <= tlen; ++c) {
                int d = D[(size_t)(r - 1) * W + c - 1] + (q[r - 1] == t[c - 1] ? 0 : 1);
                int u = D[(size_t)(r - 1) * W + c] + 1;
                int l = D[(size_t)r * W + c - 1] + 1;
                int m = d < u ? d : u;
                D[(size_t)r * W + c] = m < l ? m : l;
            }
        }
        int r = qlen, c = tlen;
        while (r > 0 || c > 0) {
            int cur = D[(size_t)r * W + c];
            if (r > 0 && D[(size_t)(r - 1) * W + c] + 1 == cur) { --r; }               /* up: 'I' */
            else if (c > 0 && D[(size_t)r * W + c - 1] + 1 == cur) { --c; }            /* left: 'D' */
            else { if (D[(size_t)(r - 1) * W + c - 1] == cur) ++*matches; --r; --c; } /* '=' or 'X' */
            ++*columns;
        }
        int ed = D[(size_t)qlen * W + tlen];
        free(D);
        return ed == best ? 0 : -1;
    }
    const int lw = tlen / 2, rw = tlen - lw;
    int32_t* L = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    int32_t* Rr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    nw_column(q, qlen, t, lw, 0, L);
    nw_column(q + qlen - 1, qlen, t + tlen - 1, rw, 1, Rr);   /* Rr[k] = last k query symbols vs the right half */
    int x = -2, ls = 0, rs = 0;
    for (int i = 0; i <= qlen - 2 && x == -2; ++i)
        if (L[i + 1] + Rr[qlen - (i + 1)] == best) { x = i; ls = L[i + 1]; rs = Rr[qlen - (i + 1)]; }
    if (x == -2 && lw + Rr[qlen] == best) { x = -1; ls = lw; rs = Rr[qlen]; }
    if (x == -2 && L[qlen] + rw == best) { x = qlen - 1; ls = L[qlen]; rs = rw; }
    free(L);
    free(Rr);
    if (x == -2) return -1;
    const int ul = x + 1;
    if (nw_path(q, ul, t, lw, ls, matches, columns)) return -1;
    return nw_path(q + ul, qlen - ul, t + lw, rw, rs, matches, columns);
}

int sdo_nw_identity(const char* q, int qlen, const char* t, int tlen, int* matches, int* columns) {
    if (matches) *matches = 0;
    if (columns) *columns = 0;
    if (qlen == 0 || tlen == 0) return -1;
    int32_t* col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    nw_column(q, qlen, t, tlen, 0, col);
    const int ed = col[qlen];
    free(col);
    int m = 0, cols = 0;
    if (nw_path(q, qlen, t, tlen, ed, &m, &cols)) return -2;
    if (matches) *matches = m;
    if (columns) *columns = cols;
    return ed;
}
