/*
 * sd_hip.h -- C-ABI of libsd_hip.so: the MI355X-native StringDecomposer DP hot path.
 *
 * Drop-in boundary.  The reference crosses from Python into native code exactly once, by running
 * its `dp` binary as a subprocess with stdout redirected to <out>_raw.tsv:
 *     stringdecomposer/main.py:194   subprocess.run([SD_BIN, sequences, monomers, num_threads,
 *                                     batch_size, overlap, ins, dels, mm, match, str(ed_thr)], stdout=f)
 *     stringdecomposer/src/main.cpp:374-402   (argv contract of that binary)
 * This header is what an FFI for that call binds instead (INTEGRATION.md shows the ctypes stub).
 * Plain C types only; no torch/HIP types in any signature (a HIP stream is passed as void*).
 *
 * There is NO CPU fallback behind these entry points: every compute call needs a gfx950 device
 * and fails with SD_ERR_NO_DEVICE otherwise.
 */
#ifndef SD_HIP_H
#define SD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define SD_OK 0
#define SD_ERR_IO 2           /* cannot open / write a file                                      */
#define SD_ERR_FORMAT 3       /* malformed FASTA (no header, header without name)                */
#define SD_ERR_PARAM 4        /* bad parameter                                                   */
#define SD_ERR_EMPTY 6        /* empty read or monomer (the reference segfaults, main.cpp:115)   */
#define SD_ERR_INTERNAL 7
#define SD_ERR_NO_DEVICE 8    /* no usable HIP device: the product path never falls back to CPU  */
#define SD_ERR_UNSUPPORTED 9  /* input outside what the device kernels support (documented)      */
#define SD_ERR_HIP 10         /* a HIP runtime call failed (message in errbuf)                   */
#define SD_ERR_SYMBOL 255     /* undefined symbol in a sequence: reference exit(-1), main.cpp:335 */

/* ---- parameters: the argv of the reference binary (main.cpp:374-402) ----------------------- */
typedef struct sd_params {
    int32_t ins, del, mismatch, match; /* argv[6..9]; defaults -1,-1,-1,1 (main.cpp:380)          */
    int32_t part_size;                 /* argv[4] (-b/--batch-size, main.py:212): chunk step, bp   */
    int32_t overlap;                   /* argv[5] (-v/--overlap,   main.py:216)                    */
    int32_t ed_thr;                    /* argv[10] (--ed_thr): -1 = off; >=0 = per-chunk prefilter */
    int32_t threads;                   /* argv[3] (-t): host threads for parse / format            */
    int32_t device;                    /* HIP device ordinal                                      */
    int32_t kernel;                    /* 0 auto, 1 generic family (int32, workgroup per chunk), 2 fast family (packed 16-bit cells, wave(s) per chunk) */
    int32_t max_batch_rows;            /* 0 = size device batches from free HBM; >0 = cap on chunk rows per batch */
    /* Switches that have no counterpart in the reference's argv (all 0 = the defaults):
     *   reserved[0]  kernel streams of the batch pipeline: 0 default (SD_PIPE_* below), else mode + 1
     *                (1: every kernel in order on one stream, 2: traceback on a second stream, 3: fills alternating too)
     *   reserved[1]  SD_FLAG_* bits
     *   reserved[2]  fp16 range guard of the fills: magnitude limit (0 = 2040); a smaller value makes the guard trip
     *                on ordinary input -- the test hook of the guard and of the re-run with integer cells
     *   reserved[3..4] must be 0 */
    int32_t reserved[5];
} sd_params;
#define SD_FLAG_NO_F16 1           /* no fp16 cell format: integer cells (or the generic family)                   */
#define SD_FLAG_NO_U16 128         /* narrow layout: no biased-u16 cells (fp16 where the range allows, as in rounds 1-5; A/B) */
#define SD_FLAG_FULL_FLOOR 2       /* fills that take the start-term maximum in every slot (A/B of the FL variants) */
#define SD_FLAG_NO_EDTHR_COMPACT 4 /* --ed_thr with > 128 templates: every chunk on the W-wave ranked kernel (a set
                                    * beyond eight waves, whose only fast form is the compacted one: generic family)   */
#define SD_FLAG_FILTER_GENERAL 8   /* --ed_thr: the general prefilter kernel instead of the uniform one             */
#define SD_FLAG_NO_STREAM_IDENT 16 /* sd_run_files: identities from the read text in the post-processing (round 2)  */
#define SD_FLAG_TRACE_V1 64        /* the one-block int32 traceback (sd_fast_trace) where the packed two-block form would run */
#define SD_FLAG_NO_IDENT_PRUNE 256 /* sd_run_files --second-best: every homopolymer-compressed pair aligned in full (rounds 3-5; A/B) */
#define SD_FLAG_PROGRESS 32        /* sd_run_files: the reference binary's progress lines on stderr ("Scores: ...",
                                      "Prepared reads", "<p>%: Aligned <read>", main.cpp:82,115,393); the command line sets it */

void sd_params_default(sd_params* p); /* -1,-1,-1,1 / 5000 / 500 / -1 / 1 / 0 / auto */

/* One monomer alignment = one raw TSV row (MonomerAlignment, main.cpp:37-49), chunk-local or
 * read-global coordinates depending on the call.  score = dp at the monomer end minus the
 * between-monomers score at its start (main.cpp:255), always integral. */
typedef struct sd_rec {
    int32_t tmpl;  /* template index: 0..M-1 monomers in file order, M..2M-1 their reverse complements */
    int32_t start;
    int32_t end;
    int32_t score;
} sd_rec;

const char* sd_version(void);
int sd_device_count(void);           /* number of visible HIP devices (0 if none / no runtime)   */
void sd_free(void* p);               /* frees anything this library returned                     */
/* Engines return their large device buffers to a process-wide cache instead of the driver (hipMalloc /
 * hipFree of the multi-GB workspaces can take longer than the kernels); this hands the cached buffers
 * back.  Environment SD_DEVICE_POOL=0 disables the cache.
 * The file and chunk-range entry points (sd_run_files*, sd_decompose_files*, sd_decompose_chunk_range) also keep the
 * device pipeline of a finished job -- engines, streams, pinned staging and their device buffers, i.e. GIGABYTES of
 * HBM that other users of the GPU in this process or on this device do not see as free -- for the next job with the
 * same parameters and monomer set (at most two pipelines, none above SD_PIPE_CACHE_GB [96] GB, none whose engines had
 * to leave their layout; SD_PIPE_CACHE_OFF=1 disables it).  sd_release_cache() destroys them too; nothing of either
 * cache is torn down at process exit. */
void sd_release_cache(void);

/* ---- one-shot entry points (replace main.py:194) ------------------------------------------- */

/* reads.fa + monomers.fa -> raw TSV file, byte-identical to `dp ... > raw_tsv_out`
 * (SaveBatch, main.cpp:272-285; reads in input order).  Error text for bad symbols equals the
 * reference's stderr line (main.cpp:335). */
int sd_decompose_files(const char* reads_fa, const char* monomers_fa, const sd_params* p,
                       const char* raw_tsv_out, char* errbuf, size_t errlen);

/* In-memory variant: monomers WITHOUT reverse complements (appended here as main.cpp:364-371).
 * *tsv is malloc'ed (sd_free). */
int sd_decompose(const char* const* read_names, const char* const* read_seqs,
                 const int64_t* read_lens, int32_t n_reads, const char* const* mono_names,
                 const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono,
                 const sd_params* p, char** tsv, size_t* tsv_len, char* errbuf, size_t errlen);

/* The whole CLI job in one call (main.py:186-197 `run` + :168-184 `convert_tsv`): both FASTA files are
 * mapped and indexed by all host threads, the chunks stream through the device in batches, and every
 * batch's rows are written three ways -- raw TSV (as sd_decompose_files), final TSV (main.py:157-160) and
 * _alt TSV (:161-165, empty without second_best) -- with the identities of main.py:29-60 computed by the
 * device kernel of sd_identity_segments_dev.  lr_coef = the three logistic-regression coefficients of
 * main.py:25-26.  Nothing is re-read from the raw file. */
int sd_run_files(const char* reads_fa, const char* monomers_fa, const sd_params* p, const char* raw_tsv_out,
                 const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity, int32_t second_best,
                 const double* lr_coef, char* errbuf, size_t errlen);

/* The same for one rank of a multi-process launch: the read set is split into `world` contiguous groups of
 * reads with about equal chunk counts and this call runs group `rank` completely (DP, identities, the three
 * TSV texts of its reads) into its own files, which the launcher concatenates in rank order.  Only the
 * reads of the group are alphabet-checked.  info (may be NULL): [0] first read, [1] one past the last,
 * [2] reads in the file, [3] chunks of this rank.  SD_ERR_UNSUPPORTED (nothing written) when the read set
 * cannot be split by reads -- one read holds more than half a rank's share, e.g. a single chromosome --
 * then shard by chunk range (sd_decompose_files_range). */
int sd_run_files_range(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                       int32_t world, const char* raw_tsv_out, const char* final_tsv_out,
                       const char* alt_tsv_out, int32_t min_identity, int32_t second_best,
                       const double* lr_coef, int64_t* info, char* errbuf, size_t errlen);
/* Stage times (ms) of the last sd_run_files / sd_run_files_range call of this process: [0] fill, [1] traceback,
 * [2] compaction, [3] in-stream identity kernels (HIP events, summed over the batches), [4] identity pairs computed
 * in-stream, [5] device batches, [6] DP rows, [7] pack + enqueue, [8] waits for the device, [9] raw text,
 * [10] post-processing, [11] file writes, [12] text-based identities (0 when they all came in-stream), [13] final /
 * _alt text, [14] whole call, [15] device / pinned allocations, [16] engine / pipeline set-up, [17] per-read assembly,
 * [18] homopolymer-compressed pairs of the job (--second-best), [19] those of them that were aligned in full (the pruned pass
 * aligns only the pairs whose identity bounds reach a record's two best), [20..23] 0.  Measurement only (bench.py, tools/). */
void sd_last_run_stats(double out[24]);
/* Batches of this process that were repeated with integer cells because the fp16 range guard of a fill tripped
 * (0 unless sd_params.reserved[2] lowers the limit, or the layout plan's range bound is wrong). */
int64_t sd_guard_trips(void);

/* convert_tsv (main.py:168-184) alone: an existing raw TSV + the two FASTA files -> final TSV and _alt TSV,
 * streamed in batches of reads.  device < 0: host identities (sd_identity_segments). */
int sd_convert_raw_tsv(const char* raw_tsv, const char* reads_fa, const char* monomers_fa,
                       const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity,
                       int32_t second_best, const double* lr_coef, int32_t device, int32_t threads,
                       char* errbuf, size_t errlen);
/* The same for one rank of a multi-process launch: the rows of the raw TSV that begin in this rank's byte range (world
 * ranges cut at line starts) into this rank's own part files; rows are independent (main.py:95-150), so the parts
 * concatenated in rank order are the files sd_convert_raw_tsv writes (shard.convert_sharded does that). */
int sd_convert_raw_tsv_range(const char* raw_tsv, const char* reads_fa, const char* monomers_fa,
                             const char* final_tsv_out, const char* alt_tsv_out, int32_t min_identity,
                             int32_t second_best, const double* lr_coef, int32_t device, int32_t threads,
                             int32_t rank, int32_t world, char* errbuf, size_t errlen);

/* ---- on-disk binary record stream ---------------------------------------
 * The reference's only hand-over between its DP stage and everything downstream is text: <out>_raw.tsv, written by
 * SaveBatch (main.cpp:272-285) and re-read line by line by convert_tsv (main.py:168-184; column spec README.md:77-83).
 * The record stream holds the same rows -- per read, in read order -- as 16-byte sd_rec records, so a consumer
 * skips the text round trip; the raw TSV is a pure function of it (sd_records_to_raw_tsv gives SaveBatch's bytes).
 * Layout: csrc/sd_records.hpp and stringdecomposer_amd/formats.py (read_records / write_records, pure Python).
 * Template names are column 2 of the raw TSV: the monomers in file order, then their names + "'" (main.cpp:364-371). */
typedef struct sd_records {
    int32_t ins, del, mismatch, match, part_size, overlap, ed_thr;   /* the job's argv (main.cpp:374-402)           */
    int32_t n_templates;
    char** tmpl_names;
    int32_t n_reads;
    char** read_names;
    int64_t* read_lens;    /* -1 where the writer did not know it                                              */
    int64_t* row_off;      /* n_reads + 1: read r owns rows[row_off[r] .. row_off[r+1]), read-global coordinates */
    int64_t n_rows;
    sd_rec* rows;
} sd_records;
/* Host only.  rows / row_off as sd_stream_collect returns them (or any rows in SaveBatch order). */
int sd_write_records(const char* path, const sd_params* p, const char* const* tmpl_names, int32_t n_templates,
                     const char* const* read_names, const int64_t* read_lens, int32_t n_reads, const sd_rec* rows,
                     const int64_t* row_off, char* errbuf, size_t errlen);
/* Host only.  Checks the file (magic, bounds, template indices, the trailer's totals: a stream whose writer did not
 * finish is SD_ERR_FORMAT); *out is filled with malloc'ed arrays, released by sd_records_free. */
int sd_read_records(const char* path, sd_records* out, char* errbuf, size_t errlen);
void sd_records_free(sd_records* r);
/* Host only: the raw TSV of a record stream, byte for byte what `dp` prints for the same job. */
int sd_records_to_raw_tsv(const char* records_path, const char* raw_tsv_out, int32_t threads, char* errbuf,
                          size_t errlen);
/* sd_decompose_files with the record stream as its output (no text is made anywhere). */
int sd_decompose_files_records(const char* reads_fa, const char* monomers_fa, const sd_params* p,
                               const char* records_out, char* errbuf, size_t errlen);
/* sd_run_files that also writes the record stream, read by read as the batches complete (records_out == NULL:
 * exactly sd_run_files).  Single process only. */
int sd_run_files_records(const char* reads_fa, const char* monomers_fa, const sd_params* p, const char* raw_tsv_out,
                         const char* final_tsv_out, const char* alt_tsv_out, const char* records_out,
                         int32_t min_identity, int32_t second_best, const double* lr_coef, char* errbuf,
                         size_t errlen);

/* ---- chunk-range form: one job sharded over several GPUs, one process per GPU ---------------
 * The chunks of a read set (main.cpp:70-81, all reads, input order) form one global table; a chunk's
 * DP depends on nothing but its own bases and the template set (main.cpp:88-96), so rank g runs the
 * contiguous range [lo_g, hi_g) of the table -- a single 200-Mb sequence splits like a million reads --
 * and one rank turns the concatenated records into the raw TSV.  No collective on the data path. */
int64_t sd_chunk_table_size(const int64_t* read_lens, int32_t n_reads, int32_t part_size, int32_t overlap);

/* Records (chunk-local coordinates, as sd_engine_fetch) of chunks [chunk_lo, chunk_hi); batched and
 * pipelined on the device like sd_decompose.  *recs / *rec_off (chunk_hi - chunk_lo + 1 entries) are
 * malloc'ed (sd_free). */
int sd_decompose_chunk_range(const char* const* read_seqs, const int64_t* read_lens, int32_t n_reads,
                             const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono,
                             const sd_params* p, int64_t chunk_lo, int64_t chunk_hi, sd_rec** recs,
                             int64_t** rec_off, char* errbuf, size_t errlen);

/* File forms, for the multi-process command line: every rank maps + indexes the FASTA (sequences are not
 * copied), runs its share block_range(n_chunks, rank, world) of the chunk table and validates only the reads
 * that share touches; rank 0 then turns the gathered records into the raw TSV file. */
int sd_decompose_files_range(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                             int32_t world, sd_rec** recs, int64_t** rec_off, int64_t* chunk_lo, int64_t* chunk_hi,
                             int64_t* n_chunks_total, char* errbuf, size_t errlen);
int sd_assemble_files_tsv(const char* reads_fa, const char* monomers_fa, const sd_params* p, const sd_rec* recs,
                          const int64_t* rec_off, int64_t n_chunks, const char* raw_tsv_out, char* errbuf,
                          size_t errlen);

/* Host only: records of ALL chunks in table order -> raw TSV (chunk offsets main.cpp:109-111, seam
 * merge :287-302, SaveBatch :272-285).  The bytes equal sd_decompose's. */
int sd_assemble_tsv(const char* const* read_names, const int64_t* read_lens, int32_t n_reads,
                    const char* const* mono_names, int32_t n_mono, const sd_params* p,
                    const sd_rec* recs, const int64_t* rec_off, int64_t n_chunks, char** tsv,
                    size_t* tsv_len, char* errbuf, size_t errlen);

/* ---- one read over several ranks: every rank makes the text of its own chunk range --------------------
 * The gather above leaves the seam merge and the text of a whole chromosome to rank 0 (46 ms for 200 Mb, whatever
 * the number of GPUs).  The merge (main.cpp:287-302) is a scan whose state is one index, so a rank can run it on its
 * own records once it knows where the scan enters them -- which follows from a few records either side of every
 * range boundary (csrc/sd_seam.hpp).  Protocol, host only, no record leaves its rank:
 *   1. sd_range_assemble_begin[_files]: chunk offsets, the merge and text of the reads that lie completely inside
 *      [chunk_lo, chunk_hi), the scan of the crossing pieces from an assumed entry, the text of their middle part;
 *      fills `edge` (POD, 160 bytes) for the exchange;
 *   2. the caller all-gathers the edges (torch.distributed all_gather_object in shard.py);
 *   3. sd_range_assemble_text(h, edges, world, rank, &bytes): the real entry from the chain of edges, the first
 *      and last rows of the crossing pieces; SD_ERR_UNSUPPORTED when some rank's edge has ok == 0 (an empty share or
 *      a crossing piece of fewer than 32 rows): gather on rank 0 instead;
 *   4. the caller all-gathers `bytes` and every rank calls sd_range_assemble_write(h, path, offset of its text,
 *      total bytes).
 * The concatenation of the ranks' texts is byte for byte what sd_assemble_tsv makes of all records. */
typedef struct sd_seam_edge {
    int32_t ok;          /* 0: this share cannot take part */
    int32_t has_front;   /* the share begins inside a read (its first records continue the previous rank's last read) */
    int32_t has_back;    /* the share ends inside a read */
    int32_t through;     /* both, and it is the same read: the share lies inside one read */
    int32_t head[8][2];  /* start, end (read coordinates) of the first eight records of the front piece */
    int32_t tail[8][2];  /* ... of the last eight records of the back piece */
    int8_t exit_of[8];   /* position 0..7 at which the scan reaches the last eight records, by entry position 0..7 */
    int64_t reserved;
} sd_seam_edge;
typedef struct sd_range_asm sd_range_asm;
int sd_range_assemble_begin(const char* const* read_names, const int64_t* read_lens, int32_t n_reads,
                            const char* const* mono_names, int32_t n_mono, const sd_params* p, int64_t chunk_lo,
                            int64_t chunk_hi, const sd_rec* recs, const int64_t* rec_off, sd_seam_edge* edge,
                            sd_range_asm** h, char* errbuf, size_t errlen);
/* names and lengths from the FASTA index; the range is block_range(n_chunks, rank, world) as in
 * sd_decompose_files_range */
int sd_range_assemble_begin_files(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                                  int32_t world, const sd_rec* recs, const int64_t* rec_off, sd_seam_edge* edge,
                                  sd_range_asm** h, char* errbuf, size_t errlen);
int sd_range_assemble_text(sd_range_asm* h, const sd_seam_edge* edges, int32_t world, int32_t rank, int64_t* text_bytes,
                           char* errbuf, size_t errlen);
/* The text into the file `path` at byte `offset`.  file_bytes >= 0: the file is created if missing and set to that
 * size first (the sum of all ranks' text_bytes: every rank passes the same value, so no rank has to wait for another);
 * file_bytes < 0: an existing file, size untouched.  sd_range_assemble_copy: the text into `buf` (text_bytes of room). */
int sd_range_assemble_write(sd_range_asm* h, const char* path, int64_t offset, int64_t file_bytes, char* errbuf,
                            size_t errlen);
int sd_range_assemble_copy(sd_range_asm* h, char* buf, int64_t room);
/* DP + step 1 in one call (the FASTA is mapped and indexed once, the records never leave the library): what
 * sd_decompose_files_range followed by sd_range_assemble_begin_files does.  sd_range_assemble_records lends the
 * share's records (valid until sd_range_assemble_free) for the gather on rank 0 when step 3 refuses. */
int sd_decompose_files_range_begin(const char* reads_fa, const char* monomers_fa, const sd_params* p, int32_t rank,
                                   int32_t world, sd_seam_edge* edge, sd_range_asm** h, int64_t* chunk_lo,
                                   int64_t* chunk_hi, int64_t* n_chunks_total, char* errbuf, size_t errlen);
int sd_range_assemble_records(sd_range_asm* h, const sd_rec** recs, const int64_t** rec_off, int64_t* n_chunks);
/* stage times of this handle in ms: [0] begin, [1] of it merge + text of complete reads, [2] assumed scans + text
 * made ahead, [3] sd_range_assemble_text, [4] rows printed by it (repaired head + tail), [5] 1 if the real scan joined
 * the assumed one too late and the piece was formatted again, [6] write */
void sd_range_assemble_stats(sd_range_asm* h, double out[8]);
void sd_range_assemble_free(sd_range_asm* h);

/* ---- engine: device-resident batches (what bench.py and the parity tests drive) ------------ */
typedef struct sd_engine sd_engine;

/* Templates = monomers followed by their reverse complements (built here).  Uploads the template
 * tables to HBM and picks the kernel family (p->kernel). */
int sd_engine_create(sd_engine** out, const sd_params* p, const char* const* mono_seqs,
                     const int32_t* mono_lens, int32_t n_mono, char* errbuf, size_t errlen);
void sd_engine_destroy(sd_engine* e);

/* Chunk the reads (main.cpp:70-81), pack them 2-bit (+N mask) and copy them to HBM.
 * Replaces any previously loaded batch.  Returns the chunk count via *n_chunks. */
int sd_engine_load_reads(sd_engine* e, const char* const* read_seqs, const int64_t* read_lens,
                         int32_t n_reads, int64_t* n_chunks, char* errbuf, size_t errlen);

/* One pass of the hot path over the loaded batch: DP fill + traceback + record compaction, all
 * on `hip_stream` (a hipStream_t cast to void*, NULL = default stream).  Asynchronous. */
int sd_engine_run(sd_engine* e, void* hip_stream, char* errbuf, size_t errlen);

/* Wait for the last run and copy the compact records to the host.  rec_off has n_chunks+1
 * entries; chunk c owns recs[rec_off[c] .. rec_off[c+1]) in read order, chunk-local coordinates.
 * Both arrays are malloc'ed (sd_free). */
int sd_engine_fetch(sd_engine* e, sd_rec** recs, int64_t** rec_off, char* errbuf, size_t errlen);

/* Per-read assembly of fetched records: chunk offsets added (main.cpp:109-111) and the seam merge
 * PostProcessing (main.cpp:287-302) applied; read r owns rows[row_off[r] .. row_off[r+1]). */
int sd_engine_assemble(sd_engine* e, const sd_rec* recs, const int64_t* rec_off, sd_rec** rows,
                       int64_t** row_off, char* errbuf, size_t errlen);

/* HIP-event timings (ms) of the last completed sd_engine_run, measured on its stream:
 * [0] fill kernel(s)  [1] traceback kernel(s)  [2] compaction  [3] whole run.  */
int sd_engine_timings(sd_engine* e, float ms[4]);

/* Static facts about the engine / loaded batch (for roofline arithmetic in bench.py):
 * [0] n_templates [1] sum of template lengths [2] n_chunks [3] sum of chunk rows
 * [4] low byte: kernel family actually used (1 generic, 2 fast); bits 8..: cell arithmetic of the fill
 *     (0 int32, 1 packed int16, 2 packed fp16 [exact small integers], 3 wide: int16 cells / int8 table,
 *      4 wide: fp16 cells / bf8 table, 5 multi-wave wide: fp16 cells / template codes in LDS, > 128 templates,
 *      6 tiled multi-wave: as 5 with a template over several virtual lanes -- templates longer than 224 bp in sets
 *      beyond one wave of the narrow layout, 7 / 8: as 5 / 6 with int16 cells and int8 table bytes -- scorings beyond
 *      the fp16 range, or after a tripped fp16 guard)
 * [5] generic: cells-per-thread parameter Q; fast: slots per lane P in bits 0..15, bits 16..: the last slot
 *     of a lane whose diagonal input needs the maximum with the start term (0 = every slot takes it)
 * [6] bytes of HBM workspace allocated [7] bits 0..15: number of fill launches per run; bits 16..: traceback of the
 *     fast family: 1 = one block per step, int32 cells (sd_fast_trace), 2 = two blocks per step, packed 16-bit cells
 *     (sd_fast_trace_pk: layouts with one wave per chunk, templates <= 256 bp, unless SD_FLAG_TRACE_V1) */
int sd_engine_info(sd_engine* e, int64_t info[8]);

/* Host only (no device needed): the layout sd_engine_create would choose for this monomer set and scoring.
 * info: [0] kernel family of "auto" (2 fast, 1 generic; generic: the reason is in errbuf, rc is still SD_OK)
 * [1] slots per lane P [2] cell arithmetic code (as sd_engine_info [4] >> 8) [3] last slot of a lane that needs
 * the maximum with the start term [4] low byte: waves per chunk; bits 8..39: the proven bound on the magnitude of a
 * stored cell (fp16 cell formats are chosen when it is <= 2040; tests/test_host_cpu.py checks it against the
 * recurrence itself); bits 40..: rows between two rebases of the stored cells (128, or 64 where only that keeps
 * the set inside the fp16 range) [5] cells in the shortest first lane of a template
 * [6] cells in the fullest lane [7] bits 0..15: common factor divided out of the four scores; bits 16..23: registers per
 * lane of the packed two-block traceback at its widest level (0 = the one-block int32 traceback runs); bits 24..55: the
 * proven bound on |E' - base| of that traceback's 16-bit words (tests/test_host_cpu.py checks it against the
 * recurrence); bit 56: the narrow fill takes its carry scan through ds_bpermute. */
int sd_plan_info(const sd_params* p, const char* const* mono_seqs, const int32_t* mono_lens, int32_t n_mono,
                 int64_t info[8], char* errbuf, size_t errlen);

/* ---- streaming form: sequences in host memory -> rows in host memory ------------------------
 * AlignReadsSet (main.cpp:67-122) without the text: chunk table (:70-81), DP + traceback per chunk
 * (:84-102), per-read flush with chunk offsets and seam merge (:104-117).  A stream keeps up to three device
 * batches in flight: every submitted job (a read set) is cut into `sub_batches` device batches of
 * consecutive chunks; while the device works on one batch the host packs and uploads the next ones (pinned
 * staging, asynchronous copies) and assembles the previous one, across job boundaries.  A caller whose jobs
 * are one batch each keeps the device busiest with TWO jobs outstanding before it collects the oldest (the
 * traceback of a batch runs at low priority beside the next fill and ends with it: with one job outstanding
 * the job after that is enqueued late).  This is the
 * region SURVEY.md 8(d) defines the throughput metric on, and what bench.py times.
 * Sequences are NOT validated here (they went through sd_fasta_load / sd_decompose's check). */
typedef struct sd_stream sd_stream;
int sd_stream_create(sd_stream** out, const sd_params* p, const char* const* mono_seqs,
                     const int32_t* mono_lens, int32_t n_mono, int32_t sub_batches, char* errbuf,
                     size_t errlen);
void sd_stream_destroy(sd_stream* s);
/* Packs, uploads and enqueues the job's batches; returns while the last ones are still running.  The
 * read buffers are no longer needed when it returns. */
int sd_stream_submit(sd_stream* s, const char* const* read_seqs, const int64_t* read_lens,
                     int32_t n_reads, char* errbuf, size_t errlen);
/* Rows of the oldest submitted job (FIFO): read r owns rows[row_off[r] .. row_off[r+1]), read-global
 * coordinates, seam-merged.  Both arrays are malloc'ed (sd_free). */
int sd_stream_collect(sd_stream* s, sd_rec** rows, int64_t** row_off, int64_t* n_rows, char* errbuf,
                      size_t errlen);
/* Accumulated over all collected batches: [0] fill [1] traceback [2] compaction [3] whole-run HIP-event
 * ms (per-batch spans; batches on the two streams overlap, so these do not add up to wall time),
 * [4] fill launches [5] batches [6] chunk rows, host ms: [7] pack+enqueue [8] wait for the device
 * [9] assembly [10] inside submit [11] inside collect, [12] jobs [13] sub_batches [14] row budget. */
int sd_stream_stats(sd_stream* s, double out[16]);
int sd_stream_info(sd_stream* s, int64_t info[8]);   /* as sd_engine_info, of the stream's engine */

/* ---- host-side pieces of the path, exported for CPU-only tests ----------------------------- */

/* chunk plan of one read (main.cpp:70-81): up to cap (offset,len) pairs; returns the count */
int32_t sd_chunk_plan(int64_t read_len, int32_t part_size, int32_t overlap, int64_t* off,
                      int32_t* len, int32_t cap);
/* PostProcessing seam merge (main.cpp:287-302), in place; returns the new count */
int32_t sd_seam_merge(sd_rec* recs, int32_t n);
/* SaveBatch text of one read's rows (main.cpp:272-285); *txt malloc'ed (sd_free) */
int sd_format_rows(const char* read_name, const char* const* tmpl_names, const sd_rec* rows,
                   int32_t n_rows, char** txt, size_t* txt_len);
/* 2-bit packing of a chunk as the device reads it (16 bases per dword, base i at bits 2*(i&15), A,C,G,T
 * = 0..3, N = 0 + a set bit in the optional 1-bit mask); returns 1 if the chunk holds an N, -1 on bad
 * arguments.  words: (n+15)/16 dwords, nmask (may be NULL): (n+31)/32 dwords. */
int32_t sd_pack_bases(const char* seq, int64_t n, uint32_t* words, uint32_t* nmask);
/* Self-test of the file writer behind sd_run_files (no device): n_parts parts of part_bytes bytes appended to
 * `path` in two calls, read back and compared.  fail_reserve != 0 makes the page reservation of the mapped
 * (tmpfs) path fail, as on a full /dev/shm -- the text must then arrive through the pwritev loop, which reports
 * ENOSPC as SD_ERR_IO instead of dying of SIGBUS.  out (may be NULL): [0] bytes written, [1] 1 on tmpfs / ramfs. */
int sd_write_parts_selftest(const char* path, int32_t n_parts, int64_t part_bytes, int32_t threads,
                            int32_t fail_reserve, int64_t out[2]);
/* Host only (CPU test): the pipeline-cache key (which jobs share cached engines) and the batch planner (how a job is cut
 * into device batches) against their contracts, without a device; SD_OK, or SD_ERR_INTERNAL with the broken property. */
int sd_pipeline_logic_selftest(char* errbuf, size_t errlen);
/* Rates of the host stages alone (no device): out[0] = chunk table + 2-bit packing, bp/s; out[1] = per-read
 * assembly (chunk offsets, seam merge) + raw TSV text of one synthetic record per 171 bases, bp/s; out[2] =
 * TSV rows/s; out[3] = bytes of text per pass.  p->threads host threads, `iters` passes over the reads. */
int sd_host_stage_rates(const char* const* read_seqs, const int64_t* read_lens, int32_t n_reads,
                        const sd_params* p, int32_t iters, double out[4]);
/* FASTA validation + load with the reference's semantics (main.cpp:314-346).  Arrays malloc'ed,
 * free with sd_fasta_free. */
typedef struct sd_fasta {
    int32_t n;
    char** names;
    char** seqs;
    int64_t* lens;
    int32_t has_n;
} sd_fasta;
int sd_fasta_load(const char* path, sd_fasta* out, char* errbuf, size_t errlen);
void sd_fasta_free(sd_fasta* f);

/* ---- post-processing helper (host): identity of a read segment vs a template ---------------
 * What main.py:29-60 (edist + aai) obtains from python-edlib: unit-cost global alignment
 * (edlib mode "NW", task "path"), number of '=' columns and total CIGAR columns.  Traceback from
 * the bottom-right corner with priority up (consume query, 'I') > left (consume target, 'D') >
 * diagonal, i.e. edlib's obtainAlignmentTraceback (edlib.cpp:945-1150); pairs whose traceback data would
 * reach 1 MB (~19.6 kb against a 171-bp target) by Hirschberg's split of the target as edlib does
 * (edlib.cpp:1186-1400).  Any byte alphabet (a symbol matches itself only), sequences of up to 2^27.
 * matches[i] = columns[i] = 0 and dist[i] = -1 if either sequence is empty (main.py:30-33).
 * Multi-threaded over pairs (threads >= 1). */
int sd_nw_identity_batch(const char* const* queries, const int32_t* qlens,
                         const char* const* targets, const int32_t* tlens, int64_t n_pairs,
                         int32_t threads, int32_t* dist, int32_t* matches, int32_t* columns);

/* The forms convert_read needs (main.py:107-150): segment s = seq[starts[s] .. ends[s]] (inclusive,
 * as in the raw TSV) is the query, a template the target.
 *   pair_tmpl == NULL: all-vs-all (--second-best), result index s * T + t;
 *   pair_tmpl != NULL: segment s against template pair_tmpl[s] only (light mode), result index s.
 * homo != 0 compresses homopolymer runs on both sides first (convert_to_homo, main.py:87-92).
 * Multi-threaded over segments. */
int sd_identity_segments(const char* seq, int64_t seqlen, const int64_t* starts, const int64_t* ends,
                         int64_t n_seg, const char* const* tmpl, const int32_t* tlen, int32_t T,
                         const int32_t* pair_tmpl, int32_t homo, int32_t threads, int32_t* dist,
                         int32_t* matches, int32_t* columns);

/* The same on the device (csrc/sd_nw.hip: one lane per (segment, template) pair, Myers bit-vectors with
 * a per-lane delta history in HBM and edlib's traceback priority): identical results.  Returns
 * SD_ERR_NO_DEVICE without a GPU and SD_ERR_UNSUPPORTED for input the kernel does not take (a symbol
 * outside ACGTN, a template longer than 512 bp, a segment longer than 65000 bp, a pair long enough for
 * edlib's Hirschberg split) -- callers then use
 * sd_identity_segments.  Device buffers are kept between calls (sd_nw_release_cache frees them). */
int sd_identity_segments_dev(const char* seq, int64_t seqlen, const int64_t* starts, const int64_t* ends,
                             int64_t n_seg, const char* const* tmpl, const int32_t* tlen, int32_t T,
                             const int32_t* pair_tmpl, int32_t homo, int32_t device, int32_t threads,
                             int32_t* dist, int32_t* matches, int32_t* columns);
void sd_nw_release_cache(void);

/* Text of `_alt.tsv` rows (main.py:161-165): for each of n_rows kept blocks one line per monomer name
 * (key): read, name, start, end, "%.2f" of vals[row * n_keys + key], '*' if key == own_key[row] else
 * '-'.  The read of a row is read_names[row_read[row]] (row_read == NULL: read_names[0] for all rows).
 * *txt is malloc'ed (sd_free).  Host only, multi-threaded. */
int sd_format_alt_rows(const char* const* read_names, int32_t n_reads, const int32_t* row_read,
                       const char* const* key_names, int32_t n_keys, const int64_t* starts,
                       const int64_t* ends, const int32_t* own_key, const double* vals, int64_t n_rows,
                       int32_t threads, char** txt, size_t* txt_len);

#ifdef __cplusplus
}
#endif
#endif /* SD_HIP_H */
