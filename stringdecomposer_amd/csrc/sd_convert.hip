// sd_convert.hip -- native post-processing of the drop-in CLI: raw monomer alignments -> the rows of
// final_decomposition.tsv and final_decomposition_alt.tsv (stringdecomposer/main.py:107-184:
// convert_read, classify, print_read, convert_tsv), streaming per batch of reads.  Host code; the
// identities come from the device kernel (sd_nw.hip) when a device is given, from the host
// implementation (sd_post.hip) otherwise or for input the kernel does not take.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sd_hip.h"
#include "sd_convert.hpp"
#include "sd_host.hpp"
#include "sd_nw.hpp"

namespace sd {

int PostProcessor::init(const std::vector<Seq>& monos, int min_identity_, bool second_best_, const double coef_[3],
                        int device_, int threads_, std::string& err) {
    min_identity = min_identity_;
    second_best = second_best_;
    coef[0] = coef_[0]; coef[1] = coef_[1]; coef[2] = coef_[2];
    device = device_;
    threads = std::max(1, threads_);
    const size_t M = monos.size();
    if (M == 0) { err = "no monomers"; return SD_ERR_PARAM; }
    il_name.clear(); il_seq.clear(); tname.clear();
    for (const Seq& m : monos) {  // add_rc_monomers, main.py:79-84: m0, m0', m1, m1', ...
        std::string rc;
        if (!reverse_complement(m.seq, rc)) { err = "undefined symbol in monomer " + m.name; return SD_ERR_SYMBOL; }
        il_name.push_back(m.name);
        il_seq.push_back(m.seq);
        il_name.push_back(m.name + "'");
        il_seq.push_back(rc);
    }
    for (const Seq& m : monos) tname.push_back(m.name);          // the DP's order, main.cpp:364-371
    for (const Seq& m : monos) tname.push_back(m.name + "'");
    // `scores` of main.py:118-121 is a dict keyed by monomer name: a repeated name keeps its first position
    // and its last value
    keys.clear(); kcol.clear();
    std::unordered_map<std::string, int> kidx;
    for (size_t x = 0; x < il_name.size(); ++x) {
        auto it = kidx.find(il_name[x]);
        if (it == kidx.end()) {
            kidx.emplace(il_name[x], (int)keys.size());
            keys.push_back(il_name[x]);
            kcol.push_back((int)x);
        } else {
            kcol[(size_t)it->second] = (int)x;
        }
    }
    key_of_t.assign(tname.size(), -1);
    own_il_of_t.assign(tname.size(), -1);
    for (size_t t = 0; t < tname.size(); ++t) {
        key_of_t[t] = kidx[tname[t]];
        own_il_of_t[t] = kcol[(size_t)key_of_t[t]];   // light mode keeps the last monomer of that name (main.py:112-116)
    }
    own_il32.assign(own_il_of_t.begin(), own_il_of_t.end());
    return SD_OK;
}

int PostProcessor::tmpl_of_name(const std::string& nm) const {
    for (size_t t = 0; t < tname.size(); ++t)
        if (tname[t] == nm) return (int)t;
    return -1;
}

// identities (percent, main.py:38-60) of n_seg segments against `tmpl` (all-vs-all, out[s * T + t]) or against
// tmpl[pair[s]] only (out[s])
int PostProcessor::identities(const std::vector<std::pair<const char*, int64_t>>& spans, const std::vector<int64_t>& seg_start,
                              const std::vector<int32_t>& seg_len, const int32_t* pair, bool homo, RawVec<double>& out,
                              std::string& err, bool host_only) {
    const int64_t n_seg = (int64_t)seg_start.size();
    const int T = (int)il_seq.size();
    const int64_t n_pairs = pair ? n_seg : n_seg * T;
    // (14 M pairs per 50 Mbp with --second-best: vectors that zero-fill cost 40 ms of one thread per call)
    RawVec<int32_t> d, m;
    d.resize((size_t)n_pairs);
    m.resize((size_t)n_pairs);
    int rc = SD_ERR_UNSUPPORTED;
    if (device >= 0 && !host_only) {
        rc = nw_identity_device(spans, seg_start.data(), seg_len.data(), n_seg, il_seq, pair, homo, device, threads, d.data(), m.data());
        if (rc != SD_OK && rc != SD_ERR_UNSUPPORTED) { err = "identity kernel failed (rc " + std::to_string(rc) + ")"; return rc; }
    }
    RawVec<int32_t> c;
    if (rc == SD_ERR_UNSUPPORTED) {  // host implementation on the concatenated text
        std::string text;
        for (const auto& sp : spans) text.append(sp.first, (size_t)sp.second);
        std::vector<int64_t> en((size_t)n_seg);
        for (int64_t s = 0; s < n_seg; ++s) en[(size_t)s] = seg_start[(size_t)s] + seg_len[(size_t)s] - 1;
        std::vector<const char*> tp;
        std::vector<int32_t> tl;
        for (const std::string& t : il_seq) { tp.push_back(t.data()); tl.push_back((int32_t)t.size()); }
        c.resize((size_t)n_pairs);
        rc = sd_identity_segments(text.data(), (int64_t)text.size(), seg_start.data(), en.data(), n_seg, tp.data(), tl.data(), T,
                                  pair, homo ? 1 : 0, threads, d.data(), m.data(), c.data());
        if (rc != SD_OK) { err = "identity computation failed (rc " + std::to_string(rc) + ")"; return rc; }
    }
    out.resize((size_t)n_pairs);
    parallel_for((n_pairs + 65535) / 65536, threads, 1, [&](int64_t blk) {
        const int64_t e = std::min<int64_t>(n_pairs, (blk + 1) * 65536);
        for (int64_t p = blk * 65536; p < e; ++p) {
            if (d[(size_t)p] < 0) { out[(size_t)p] = 0.0; continue; }
            double a = 0.0;
            a += (double)m[(size_t)p];                    // sum of the '=' run lengths
            a /= (double)(d[(size_t)p] + m[(size_t)p]);   // all CIGAR columns
            out[(size_t)p] = a * 100;
        }
    });
    return SD_OK;
}

namespace {
inline double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void put_f2(std::string& o, double v) { put_fixed2(o, v); }   // == Python "{:.2f}".format(v)
}  // namespace

namespace {
// identity in percent of a device word (dist << 16) | matches: the arithmetic of main.py:47-60 on the same integers
inline double ident_percent(uint32_t w) {
    const uint32_t d = w >> 16, m = w & 0xffffu;
    double a = 0.0;
    a += (double)m;
    a /= (double)(d + m);
    return a * 100;
}
}  // namespace

int PostProcessor::process(const PostRead* reads, size_t n_reads, const sd_rec* rows, const int64_t* row_off, TextBuf& fin,
                           TextBuf& alt, std::string& err, const IdentRef* ident) {
    fin.clear();
    alt.clear();
    std::vector<std::string> pf;
    std::vector<TextBuf> pa;
    const int rc = process_parts(reads, n_reads, rows, row_off, pf, pa, err, ident);
    if (rc) return rc;
    const double t_c = now_seconds();
    gather_text(pf, threads, fin);
    gather_text(pa, threads, alt);
    t_concat += now_seconds() - t_c;
    return SD_OK;
}

int PostProcessor::process_parts(const PostRead* reads, size_t n_reads, const sd_rec* rows, const int64_t* row_off,
                                 std::vector<std::string>& pf, std::vector<TextBuf>& pa, std::string& err,
                                 const IdentRef* ident) {
    // (pf / pa may come back from an earlier call: their elements keep their memory -- a fresh 0.5-MB buffer per slice of
    // blocks is an mmap and a few hundred page faults, 280 MB of them per C4 --second-best job)
    const int64_t nB = row_off[n_reads];
    if (nB == 0) { pf.clear(); pa.clear(); return SD_OK; }
    const double t_0 = now_seconds();
    const int T = (int)il_seq.size();
    const int nK = (int)keys.size();
    const int per = second_best ? T : 1;
    bool have = ident && ident->id && (!second_best || ident->idh);
    bool host_only = false;   // the batch holds a pair edlib aligns by Hirschberg's split
    // words of row b (plain / compressed)
    auto words = [&](int64_t b, bool homo) -> const uint32_t* {
        if (!ident->src) return (homo ? ident->idh : ident->id) + (size_t)b * (size_t)per;
        const int64_t sx = ident->src[b];
        return sx >= 0 ? (homo ? ident->idh : ident->id) + (size_t)sx * (size_t)per
                       : (homo ? ident->xidh : ident->xid) + (size_t)(-1 - sx) * (size_t)per;
    };
    if (have) {
        // every word computed?  (0xffffffff: a pair the kernel left out; dist + matches == 0 cannot be an alignment)
        std::vector<uint8_t> bad((size_t)((nB + 4095) / 4096), 0);
        parallel_for((int64_t)bad.size(), threads, 1, [&](int64_t blk) {
            const int64_t e = std::min<int64_t>(nB, (blk + 1) * 4096);
            uint8_t bb = 0;
            for (int64_t b = blk * 4096; b < e; ++b) {
                const uint32_t* v = words(b, false);
                for (int x = 0; x < per; ++x) bb |= (uint8_t)(v[x] == 0xffffffffu || v[x] == 0u);
                if (second_best) {
                    const uint32_t* h = words(b, true);
                    for (int x = 0; x < per; ++x) bb |= (uint8_t)(h[x] == 0xffffffffu || h[x] == 0u);
                }
            }
            bad[(size_t)blk] = bb;
        });
        for (uint8_t bb : bad) if (bb) { have = false; break; }
    }
    // text = the reads that have blocks, concatenated; blocks never cross a read
    std::vector<std::pair<const char*, int64_t>> spans;
    std::vector<int64_t> seg_start((size_t)nB);
    std::vector<int32_t> seg_len((size_t)nB), read_of((size_t)nB);
    {
        int64_t pos = 0;
        for (size_t r = 0; r < n_reads; ++r) {
            if (row_off[r + 1] == row_off[r]) continue;
            for (int64_t b = row_off[r]; b < row_off[r + 1]; ++b) {
                const sd_rec& x = rows[b];
                if (x.tmpl < 0 || x.tmpl >= (int)tname.size()) { err = "row with a template outside the monomer set"; return SD_ERR_PARAM; }
                // Python slicing read.seq[start:end + 1] clamps to the sequence
                const int64_t s0 = std::min<int64_t>(std::max<int64_t>(x.start, 0), reads[r].len);
                const int64_t e1 = std::min<int64_t>(std::max<int64_t>((int64_t)x.end + 1, s0), reads[r].len);
                seg_start[(size_t)b] = pos + s0;
                seg_len[(size_t)b] = (int32_t)std::min<int64_t>(e1 - s0, 0x7fffffff);
                read_of[(size_t)b] = (int32_t)r;
            }
            spans.emplace_back(reads[r].seq, reads[r].len);
            pos += reads[r].len;
        }
    }
    {
        // edlib computes the path of a long alignment with Hirschberg's split instead of its block traceback
        // (edlib.cpp:1186-1190; ~19.6 kb against a 171-bp monomer: -b >= 19000 or kb-long monomers).  The device
        // kernels implement the traceback only, so a batch that holds such a pair takes the host identities, which
        // follow edlib in both cases (sd_post.hip)
        size_t tmax = 1;
        for (const std::string& t : il_seq) tmax = std::max(tmax, t.size());
        int64_t worst = 0;
        for (int64_t b = 0; b < nB; ++b) worst = std::max<int64_t>(worst, seg_len[(size_t)b]);
        // (conservative: longest block x longest template.  Monomers beyond 512 bp: the device path deals the pairs one by
        // one -- block traceback on the device, the split on host threads, sd_nw.hip -- so the batch is not held back)
        host_only = edlib_splits(worst, (int64_t)tmax) && tmax <= 512;
        if (edlib_splits(worst, (int64_t)tmax)) have = false;
    }
    const bool id = have;   // identities came with the rows
    const double t_a = now_seconds();
    RawVec<double> vals, hvals;
    int rc;
    if (id) {
        // identities came with the rows
    } else if (!second_best) {
        std::vector<int32_t> pair((size_t)nB);
        for (int64_t b = 0; b < nB; ++b) pair[(size_t)b] = own_il_of_t[(size_t)rows[b].tmpl];
        rc = identities(spans, seg_start, seg_len, pair.data(), false, vals, err, host_only);
        if (rc) return rc;
    } else {
        rc = identities(spans, seg_start, seg_len, nullptr, false, vals, err, host_only);
        if (rc) return rc;
        rc = identities(spans, seg_start, seg_len, nullptr, true, hvals, err, host_only);
        if (rc) return rc;
    }
    const double t_b = now_seconds();
    // rows -> text, in slices of blocks formatted by all threads
    const int64_t grain = second_best ? 64 : 2048;
    const int64_t n_sl = (nB + grain - 1) / grain;
    pf.resize((size_t)n_sl);
    pa.resize((size_t)n_sl);
    for (std::string& q : pf) q.clear();
    for (TextBuf& q : pa) q.clear();
    size_t key_bytes = 0, name_max = 0;
    for (const std::string& k : keys) key_bytes += k.size();
    for (size_t r = 0; r < n_reads; ++r) name_max = std::max(name_max, reads[r].name_len);
    parallel_for(n_sl, threads, 1, [&](int64_t sl) {
        std::string& of = pf[(size_t)sl];
        TextBuf& oa = pa[(size_t)sl];
        const int64_t b1 = std::min(nB, (sl + 1) * grain);
        // room for the slice up front: a string that doubles its way up copies (and page-faults) the text twice over
        of.reserve((size_t)(b1 - sl * grain) * (name_max + 160));
        // _alt rows (nK per block, 300 MB per C4 batch) are written through a pointer into an uninitialised buffer
        // sized for the worst case: name + key + two 11-digit ints + a 48-byte number + 6 separators per row
        const size_t alt_row_max = name_max + 11 + 11 + 48 + 8;
        if (second_best) oa.resize((size_t)(b1 - sl * grain) * (key_bytes + (size_t)nK * alt_row_max));
        char* wa = oa.data();
        std::vector<double> kbuf((size_t)nK), hbuf((size_t)T);
        for (int64_t b = sl * grain; b < b1; ++b) {
            const sd_rec& x = rows[b];
            const PostRead& rd = reads[(size_t)read_of[(size_t)b]];
            const int ko = key_of_t[(size_t)x.tmpl];
            double score, sbs = -1, h0s = -1, h1s = -1;
            const std::string* sbn = nullptr;
            const std::string* h0n = nullptr;
            const std::string* h1n = nullptr;
            const double* kv = nullptr;
            if (!second_best) {
                score = id ? ident_percent(words(b, false)[0]) : vals[(size_t)b];
            } else {
                if (id) {
                    const uint32_t* v = words(b, false);
                    for (int k = 0; k < nK; ++k) kbuf[(size_t)k] = ident_percent(v[kcol[(size_t)k]]);
                    const uint32_t* hw = words(b, true);
                    for (int j = 0; j < T; ++j) hbuf[(size_t)j] = ident_percent(hw[j]);
                } else {
                    const double* v = &vals[(size_t)b * T];
                    for (int k = 0; k < nK; ++k) kbuf[(size_t)k] = v[kcol[(size_t)k]];
                }
                kv = kbuf.data();
                score = kv[ko];
                int sb = -1;   // main.py:124-128: first maximum among the other names
                for (int k = 0; k < nK; ++k) {
                    if (k == ko) continue;
                    if (sb < 0 || sbs < kv[k]) { sb = k; sbs = kv[k]; }
                }
                if (sb >= 0) sbn = &keys[(size_t)sb];
                else sbs = -1;
                // main.py:130-135: all monomers (the own one included), stable sort by -score: ranks 0 and 1
                const double* h = id ? hbuf.data() : &hvals[(size_t)b * T];
                int i0 = 0;
                for (int j = 1; j < T; ++j) if (h[j] > h[i0]) i0 = j;
                int i1 = -1;
                for (int j = 0; j < T; ++j) {
                    if (j == i0) continue;
                    if (i1 < 0 || h[j] > h[i1]) i1 = j;
                }
                h0n = &il_name[(size_t)i0]; h0s = h[i0];
                if (i1 >= 0) { h1n = &il_name[(size_t)i1]; h1s = h[i1]; }
            }
            if (!(score >= (double)min_identity)) continue;     // main.py:156
            // classify (main.py:95-104): intercept + c1 * identity + c2 * (identity - second best) > 0
            const double logit = (1.0 * coef[0] + score * coef[1]) + (score - sbs) * coef[2];
            static const std::string none = "None";
            of.append(rd.name, rd.name_len); of.push_back('\t');
            of.append(tname[(size_t)x.tmpl]); of.push_back('\t');
            put_int(of, x.start); of.push_back('\t');
            put_int(of, x.end); of.push_back('\t');
            put_f2(of, score); of.push_back('\t');
            of.append(sbn ? *sbn : none); of.push_back('\t');
            put_f2(of, sbs); of.push_back('\t');
            of.append(h0n ? *h0n : none); of.push_back('\t');
            put_f2(of, h0s); of.push_back('\t');
            of.append(h1n ? *h1n : none); of.push_back('\t');
            put_f2(of, h1s); of.push_back('\t');
            of.push_back(logit > 0 ? '+' : '?');
            of.push_back('\n');
            if (second_best) {   // main.py:161-165: one row per name of the dict
                char mid[32];    // "\t<start>\t<end>\t": the same for the nK rows of the block
                size_t ml = 0;
                {
                    std::string t;
                    t.push_back('\t'); put_int(t, x.start); t.push_back('\t'); put_int(t, x.end); t.push_back('\t');
                    ml = t.size();
                    std::memcpy(mid, t.data(), ml);
                }
                for (int k = 0; k < nK; ++k) {
                    std::memcpy(wa, rd.name, rd.name_len); wa += rd.name_len;
                    *wa++ = '\t';
                    const std::string& key = keys[(size_t)k];
                    std::memcpy(wa, key.data(), key.size()); wa += key.size();
                    std::memcpy(wa, mid, ml); wa += ml;
                    wa = put_fixed2_at(wa, kv[k]);
                    std::memcpy(wa, k == ko ? "\t*\n" : "\t-\n", 3); wa += 3;
                }
            }
        }
        if (second_best) oa.resize((size_t)(wa - oa.data()));
    });
    t_prepare += t_a - t_0;
    t_identity += t_b - t_a;
    t_format += now_seconds() - t_b;
    return SD_OK;
}

}  // namespace sd

// ---------------------------------------------------------------------------------------------
// convert_tsv (main.py:168-184) as one native call: raw TSV file + the two FASTA files -> final TSV
// and _alt TSV files, streamed in batches of reads.
// ---------------------------------------------------------------------------------------------
// convert_tsv on the rows of the raw TSV that begin in the byte range of `rank` (of `world` ranges cut at line starts:
// range g begins behind the first line end at or after byte size*g/world - 1).  Rows are independent of each other
// (convert_read, main.py:107-150, and classify, :95-104, work row by row), so the final / _alt texts of the ranges,
// concatenated in rank order, are the texts of the whole file.
static int convert_impl(const char* raw_tsv, const char* reads_fa, const char* monomers_fa,
                        const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity,
                        int32_t second_best, const double* lr_coef, int32_t device, int32_t threads,
                        int32_t rank, int32_t world, char* errbuf, size_t errlen) {
    auto fail = [&](int rc, const std::string& m) {
        if (errbuf && errlen) std::snprintf(errbuf, errlen, "%s", m.c_str());
        return rc;
    };
    if (!raw_tsv || !reads_fa || !monomers_fa || !final_tsv_out || !alt_tsv_out || !lr_coef) return fail(SD_ERR_PARAM, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(SD_ERR_PARAM, "bad rank / world");
    std::string err;
    sd::FastaFile rf, mf;
    int rc = rf.open(reads_fa, threads, err);
    if (rc) return fail(rc, err);
    rc = mf.open(monomers_fa, threads, err);
    if (rc) return fail(rc, err);
    std::vector<sd::Seq> monos;
    for (const auto& r : mf.recs) {
        sd::Seq m{std::string(r.name, r.name_len), std::string(r.seq, (size_t)r.len)};
        for (char& c : m.seq) c = (char)std::toupper((unsigned char)c);      // load_fasta(...).upper(), main.py:63-74
        monos.push_back(std::move(m));
    }
    sd::PostProcessor pp;
    rc = pp.init(monos, min_identity, second_best != 0, lr_coef, device, threads, err);
    if (rc) return fail(rc, err);
    // reads by name (SeqIO.to_dict, main.py:65: a repeated id is an error)
    std::unordered_map<std::string, size_t> by_name;
    by_name.reserve(rf.recs.size() * 2);
    for (size_t r = 0; r < rf.recs.size(); ++r)
        if (!by_name.emplace(std::string(rf.recs[r].name, rf.recs[r].name_len), r).second)
            return fail(SD_ERR_FORMAT, "Duplicate key '" + std::string(rf.recs[r].name, rf.recs[r].name_len) + "'");
    std::unordered_map<std::string, int> tmpl_by_name;
    for (size_t t = pp.tname.size(); t-- > 0;) tmpl_by_name[pp.tname[t]] = (int)t;
    // lower-case read sequences are upper-cased by the reference before slicing (main.py:66-67); the DP
    // itself only accepts upper case, so a copy is needed only for files that never went through it
    std::vector<std::string> upper(rf.recs.size());
    FILE* fr = std::fopen(raw_tsv, "rb");
    if (!fr) return fail(SD_ERR_IO, std::string("cannot open ") + raw_tsv);
    FILE* ff = std::fopen(final_tsv_out, "wb");
    FILE* fa = std::fopen(alt_tsv_out, "wb");
    if (!ff || !fa) {
        std::fclose(fr);
        if (ff) std::fclose(ff);
        if (fa) std::fclose(fa);
        return fail(SD_ERR_IO, std::string("cannot write ") + final_tsv_out);
    }
    std::vector<sd::PostRead> reads;      // reads of the current batch, in file order of the raw TSV
    std::vector<sd_rec> rows;
    std::vector<int64_t> row_off(1, 0);
    sd::TextBuf fin, alt;
    auto flush = [&]() -> int {
        if (reads.empty()) return SD_OK;
        const int r2 = pp.process(reads.data(), reads.size(), rows.data(), row_off.data(), fin, alt, err);
        if (r2) return r2;
        if (std::fwrite(fin.data(), 1, fin.size(), ff) != fin.size() || std::fwrite(alt.data(), 1, alt.size(), fa) != alt.size()) {
            err = std::string("short write to ") + final_tsv_out;
            return SD_ERR_IO;
        }
        reads.clear();
        rows.clear();
        row_off.assign(1, 0);
        return SD_OK;
    };
    // this rank's byte range [lo, hi) of the raw file, cut at line starts
    int64_t lo = 0, hi = 0;
    {
        std::fseek(fr, 0, SEEK_END);
        const int64_t size = (int64_t)std::ftell(fr);
        auto bound = [&](int g) -> int64_t {
            if (g <= 0) return 0;
            if (g >= world) return size;
            int64_t at = size / world * g + size % world * g / world;
            if (at <= 0) return 0;
            std::fseek(fr, (long)(at - 1), SEEK_SET);
            int ch;
            int64_t q = at - 1;
            while ((ch = std::fgetc(fr)) != EOF) {
                ++q;
                if (ch == '\n') return q;
            }
            return size;
        };
        lo = bound(rank);
        hi = bound(rank + 1);
        std::fseek(fr, (long)lo, SEEK_SET);
    }
    int64_t pos = lo;
    char* line = nullptr;
    size_t cap = 0;
    ssize_t got;
    std::string prev;
    bool have_prev = false;
    const int64_t batch_blocks = second_best ? 65536 : 1 << 20;
    while (rc == SD_OK && pos < hi && (got = getline(&line, &cap, fr)) > 0) {
        pos += got;
        if (line[got - 1] != '\n') break;   // decomposition.split("\n")[:-1] drops an unterminated last line
        // read \t monomer \t start \t end ...
        const char* f[5];
        int nf = 0;
        f[nf++] = line;
        for (ssize_t i = 0; i < got && nf < 5; ++i)
            if (line[i] == '\t') f[nf++] = line + i + 1;
        if (nf < 4) { err = "malformed raw TSV line"; rc = SD_ERR_FORMAT; break; }
        auto token = [&](int k) {   // field k up to its tab, then .split()[0]
            const char* b = f[k];
            const char* e = (k + 1 < nf ? f[k + 1] - 1 : line + got - 1);
            while (b < e && sd::is_ws(*b)) ++b;
            const char* z = b;
            while (z < e && !sd::is_ws(*z)) ++z;
            return std::string(b, (size_t)(z - b));
        };
        const std::string rname = token(0), mname = token(1);
        if (!have_prev || rname != prev) {
            if (!reads.empty()) {
                row_off.push_back((int64_t)rows.size());
                if ((int64_t)rows.size() >= batch_blocks) rc = flush();
                if (rc) break;
            }
            auto it = by_name.find(rname);
            if (it == by_name.end()) { err = "read '" + rname + "' of the raw TSV is not in " + reads_fa; rc = SD_ERR_FORMAT; break; }
            const sd::FastaFile::Rec& rr = rf.recs[it->second];
            const char* sq = rr.seq;
            bool lower = false;
            for (int64_t i = 0; i < rr.len && !lower; ++i) lower = rr.seq[i] >= 'a' && rr.seq[i] <= 'z';
            if (lower) {
                std::string& u = upper[it->second];
                if (u.empty()) {
                    u.assign(rr.seq, (size_t)rr.len);
                    for (char& c : u) c = (char)std::toupper((unsigned char)c);
                }
                sq = u.data();
            }
            reads.push_back(sd::PostRead{rr.name, rr.name_len, sq, rr.len});
            prev = rname;
            have_prev = true;
        }
        auto ti = tmpl_by_name.find(mname);
        if (ti == tmpl_by_name.end()) { err = "monomer '" + mname + "' of the raw TSV is not in " + monomers_fa; rc = SD_ERR_FORMAT; break; }
        sd_rec x;
        x.tmpl = ti->second;
        x.start = (int32_t)std::strtol(f[2], nullptr, 10);
        x.end = (int32_t)std::strtol(f[3], nullptr, 10);
        x.score = 0;
        rows.push_back(x);
    }
    std::free(line);
    if (rc == SD_OK && !reads.empty()) {
        row_off.push_back((int64_t)rows.size());
        rc = flush();
    }
    std::fclose(fr);
    const bool wf = std::fclose(ff) == 0, wa = std::fclose(fa) == 0;
    if (rc == SD_OK && !(wf && wa)) { rc = SD_ERR_IO; err = std::string("short write to ") + final_tsv_out; }
    if (rc) return fail(rc, err);
    return SD_OK;
}

extern "C" int sd_convert_raw_tsv(const char* raw_tsv, const char* reads_fa, const char* monomers_fa,
                                  const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity,
                                  int32_t second_best, const double* lr_coef, int32_t device, int32_t threads,
                                  char* errbuf, size_t errlen) {
    return convert_impl(raw_tsv, reads_fa, monomers_fa, final_tsv_out, alt_tsv_out, min_identity, second_best, lr_coef, device,
                        threads, 0, 1, errbuf, errlen);
}

extern "C" int sd_convert_raw_tsv_range(const char* raw_tsv, const char* reads_fa, const char* monomers_fa,
                                        const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity,
                                        int32_t second_best, const double* lr_coef, int32_t device, int32_t threads,
                                        int32_t rank, int32_t world, char* errbuf, size_t errlen) {
    return convert_impl(raw_tsv, reads_fa, monomers_fa, final_tsv_out, alt_tsv_out, min_identity, second_best, lr_coef, device,
                        threads, rank, world, errbuf, errlen);
}
