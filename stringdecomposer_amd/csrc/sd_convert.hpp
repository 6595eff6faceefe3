// sd_convert.hpp -- native post-processing (sd_convert.hip): raw monomer alignments of a batch of reads ->
// rows of final_decomposition.tsv / _alt.tsv (stringdecomposer/main.py:107-165).
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sd_hip.h"
#include "sd_host.hpp"

namespace sd {

struct PostRead {
    const char* name; size_t name_len;
    const char* seq; int64_t len;       // upper-case sequence (main.py:66-67)
};

// Identities that came with the rows from the device (sd_ident.hip): one word (dist << 16) | matches per row (light
// mode: the row's own monomer) or per (row, interleaved monomer) pair, plain (id) / homopolymer-compressed (idh).
// src == nullptr: row b's words start at id + b * per.  Otherwise src[b] >= 0: record src[b] of id / idh; src[b] < 0:
// entry -1 - src[b] of xid / xidh (rows of a read that began in an earlier device batch).
struct IdentRef {
    const uint32_t* id = nullptr;
    const uint32_t* idh = nullptr;
    const int64_t* src = nullptr;
    const uint32_t* xid = nullptr;
    const uint32_t* xidh = nullptr;
};

class PostProcessor {
  public:
    // monomers in file order (names = first header token, sequences upper-case); device < 0: host identities
    int init(const std::vector<Seq>& monos, int min_identity, bool second_best, const double coef[3], int device,
             int threads, std::string& err);
    // rows[row_off[r] .. row_off[r+1]) = blocks of reads[r] (sd_rec.tmpl in the DP's template order: monomers,
    // then their reverse complements; read-global inclusive coordinates).  Appends the text of the final TSV
    // rows (main.py:157-160) and, with second_best, of the _alt rows (:161-165).
    // id / idh (optional): identities that came with the rows from the device (sd_ident.hip), one word
    // (dist << 16) | matches per row (light mode: the row's own monomer) or per (row, interleaved monomer) pair,
    // plain / homopolymer-compressed; rows without a computed word send the batch through the text-based path.
    int process(const PostRead* reads, size_t n_reads, const sd_rec* rows, const int64_t* row_off, TextBuf& fin,
                TextBuf& alt, std::string& err, const IdentRef* ident = nullptr);   // fin / alt are REPLACED
    // the same, as the slices the threads formatted (in order; the caller writes them without a gather copy)
    int process_parts(const PostRead* reads, size_t n_reads, const sd_rec* rows, const int64_t* row_off,
                      std::vector<std::string>& fin_parts, std::vector<TextBuf>& alt_parts, std::string& err,
                      const IdentRef* ident = nullptr);
    const std::vector<std::string>& interleaved_seqs() const { return il_seq; }   // m0, m0', m1, m1', ... (main.py:79-84)
    const std::vector<int32_t>& own_interleaved() const { return own_il32; }      // DP template -> interleaved index
    int tmpl_of_name(const std::string& nm) const;   // first template of that name in the DP's order, -1 if none
    std::vector<std::string> tname;                  // the DP's template names: m, ..., m', ...
    double t_prepare = 0, t_identity = 0, t_format = 0, t_concat = 0;   // seconds spent in process(), by stage

  private:
    int identities(const std::vector<std::pair<const char*, int64_t>>& spans, const std::vector<int64_t>& seg_start,
                   const std::vector<int32_t>& seg_len, const int32_t* pair, bool homo, RawVec<double>& out,
                   std::string& err, bool host_only = false);
    std::vector<std::string> il_name, il_seq;        // interleaved m0, m0', m1, m1', ... (main.py:79-84)
    std::vector<std::string> keys;                   // distinct names in first-occurrence order
    std::vector<int> kcol;                           // key -> last interleaved index of that name
    std::vector<int> key_of_t, own_il_of_t;
    std::vector<int32_t> own_il32;
    int min_identity = 0;
    bool second_best = false;
    double coef[3] = {0, 0, 0};
    int device = -1;
    int threads = 1;
};

}  // namespace sd
