"""ctypes binding of libsd_hip.so (include/sd_hip.h).

This is the call that replaces the reference's subprocess boundary
(stringdecomposer/main.py:194: subprocess.run([SD_BIN, ...], stdout=raw_file)).  The library has
no CPU fallback; a missing library or a missing GPU is a loud error.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SD_HIP_LIB lets a developer A/B an alternative build of the same library (never a CPU path)
LIB_PATH = os.environ.get("SD_HIP_LIB") or os.path.join(HERE, "csrc", "libsd_hip.so")

SD_OK = 0
SD_ERR_IO = 2
SD_ERR_FORMAT = 3
SD_ERR_PARAM = 4
SD_ERR_EMPTY = 6
SD_ERR_INTERNAL = 7
SD_ERR_NO_DEVICE = 8
SD_ERR_UNSUPPORTED = 9
SD_ERR_HIP = 10
SD_ERR_SYMBOL = 255

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST = 0, 1, 2

EXPORTS = [
    "sd_pipeline_logic_selftest",
    "sd_params_default", "sd_version", "sd_device_count", "sd_free", "sd_decompose_files",
    "sd_decompose", "sd_engine_create", "sd_engine_destroy", "sd_engine_load_reads",
    "sd_engine_run", "sd_engine_fetch", "sd_engine_assemble", "sd_engine_timings",
    "sd_engine_info", "sd_chunk_plan", "sd_seam_merge", "sd_format_rows", "sd_fasta_load",
    "sd_fasta_free", "sd_nw_identity_batch", "sd_identity_segments", "sd_chunk_table_size",
    "sd_decompose_chunk_range", "sd_assemble_tsv", "sd_release_cache", "sd_format_alt_rows",
    "sd_stream_create", "sd_stream_destroy", "sd_stream_submit", "sd_stream_collect", "sd_stream_stats",
    "sd_stream_info", "sd_pack_bases", "sd_identity_segments_dev", "sd_nw_release_cache", "sd_last_run_stats", "sd_guard_trips",
    "sd_run_files", "sd_convert_raw_tsv", "sd_decompose_files_range", "sd_assemble_files_tsv",
    "sd_host_stage_rates", "sd_run_files_range", "sd_plan_info", "sd_write_parts_selftest", "sd_convert_raw_tsv_range",
    "sd_write_records", "sd_read_records", "sd_records_free", "sd_records_to_raw_tsv", "sd_decompose_files_records",
    "sd_run_files_records",
    "sd_range_assemble_begin", "sd_range_assemble_begin_files", "sd_range_assemble_text", "sd_range_assemble_write",
    "sd_range_assemble_copy", "sd_range_assemble_stats", "sd_range_assemble_free", "sd_decompose_files_range_begin",
    "sd_range_assemble_records",
]


class SdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsd_hip rc=%d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class Params(C.Structure):
    _fields_ = [("ins", C.c_int32), ("del_", C.c_int32), ("mismatch", C.c_int32),
                ("match", C.c_int32), ("part_size", C.c_int32), ("overlap", C.c_int32),
                ("ed_thr", C.c_int32), ("threads", C.c_int32), ("device", C.c_int32),
                ("kernel", C.c_int32), ("max_batch_rows", C.c_int32), ("reserved", C.c_int32 * 5)]


class Rec(C.Structure):
    _fields_ = [("tmpl", C.c_int32), ("start", C.c_int32), ("end", C.c_int32),
                ("score", C.c_int32)]


class Records(C.Structure):
    """sd_records (include/sd_hip.h): a parsed binary record stream."""
    _fields_ = [("ins", C.c_int32), ("del_", C.c_int32), ("mismatch", C.c_int32), ("match", C.c_int32),
                ("part_size", C.c_int32), ("overlap", C.c_int32), ("ed_thr", C.c_int32),
                ("n_templates", C.c_int32), ("tmpl_names", C.POINTER(C.c_char_p)),
                ("n_reads", C.c_int32), ("read_names", C.POINTER(C.c_char_p)),
                ("read_lens", C.POINTER(C.c_int64)), ("row_off", C.POINTER(C.c_int64)),
                ("n_rows", C.c_int64), ("rows", C.POINTER(Rec))]


class Fasta(C.Structure):
    _fields_ = [("n", C.c_int32), ("names", C.POINTER(C.c_char_p)),
                ("seqs", C.POINTER(C.c_void_p)), ("lens", C.POINTER(C.c_int64)),
                ("has_n", C.c_int32)]


_lib = None


def load():
    """Load libsd_hip.so.  Raises (never silently falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise SdError(SD_ERR_INTERNAL,
                      "%s is missing: build it with `make -C stringdecomposer_amd/csrc` "
                      "(or python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.sd_params_default.argtypes = [P(Params)]
    L.sd_version.restype = C.c_char_p
    L.sd_device_count.restype = C.c_int
    L.sd_free.argtypes = [C.c_void_p]
    L.sd_decompose_files.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_char_p, C.c_char_p, C.c_size_t]
    L.sd_decompose.argtypes = [P(C.c_char_p), P(C.c_char_p), P(C.c_int64), C.c_int32, P(C.c_char_p),
                               P(C.c_char_p), P(C.c_int32), C.c_int32, P(Params), P(C.c_void_p),
                               P(C.c_size_t), C.c_char_p, C.c_size_t]
    L.sd_engine_create.argtypes = [P(C.c_void_p), P(Params), P(C.c_char_p), P(C.c_int32), C.c_int32,
                                   C.c_char_p, C.c_size_t]
    L.sd_engine_destroy.argtypes = [C.c_void_p]
    L.sd_engine_load_reads.argtypes = [C.c_void_p, P(C.c_char_p), P(C.c_int64), C.c_int32,
                                       P(C.c_int64), C.c_char_p, C.c_size_t]
    L.sd_engine_run.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.sd_engine_fetch.argtypes = [C.c_void_p, P(P(Rec)), P(P(C.c_int64)), C.c_char_p, C.c_size_t]
    L.sd_engine_assemble.argtypes = [C.c_void_p, P(Rec), P(C.c_int64), P(P(Rec)), P(P(C.c_int64)),
                                     C.c_char_p, C.c_size_t]
    L.sd_engine_timings.argtypes = [C.c_void_p, P(C.c_float)]
    L.sd_engine_info.argtypes = [C.c_void_p, P(C.c_int64)]
    L.sd_plan_info.argtypes = [P(Params), P(C.c_char_p), P(C.c_int32), C.c_int32, P(C.c_int64), C.c_char_p, C.c_size_t]
    L.sd_chunk_plan.restype = C.c_int32
    L.sd_chunk_plan.argtypes = [C.c_int64, C.c_int32, C.c_int32, P(C.c_int64), P(C.c_int32), C.c_int32]
    L.sd_seam_merge.restype = C.c_int32
    L.sd_seam_merge.argtypes = [P(Rec), C.c_int32]
    L.sd_format_rows.argtypes = [C.c_char_p, P(C.c_char_p), P(Rec), C.c_int32, P(C.c_void_p), P(C.c_size_t)]
    L.sd_fasta_load.argtypes = [C.c_char_p, P(Fasta), C.c_char_p, C.c_size_t]
    L.sd_fasta_free.argtypes = [P(Fasta)]
    L.sd_nw_identity_batch.argtypes = [P(C.c_char_p), P(C.c_int32), P(C.c_char_p), P(C.c_int32),
                                       C.c_int64, C.c_int32, P(C.c_int32), P(C.c_int32), P(C.c_int32)]
    L.sd_identity_segments.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                       P(C.c_char_p), P(C.c_int32), C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sd_identity_segments_dev.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                           P(C.c_char_p), P(C.c_int32), C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sd_run_files.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32,
                               C.c_int32, P(C.c_double), C.c_char_p, C.c_size_t]
    L.sd_run_files_range.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_int32, C.c_int32, C.c_char_p, C.c_char_p,
                                     C.c_char_p, C.c_int32, C.c_int32, P(C.c_double), P(C.c_int64), C.c_char_p, C.c_size_t]
    L.sd_convert_raw_tsv.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32,
                                     P(C.c_double), C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]
    L.sd_decompose_files_range.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_int32, C.c_int32, P(P(Rec)),
                                           P(P(C.c_int64)), P(C.c_int64), P(C.c_int64), P(C.c_int64), C.c_char_p,
                                           C.c_size_t]
    L.sd_assemble_files_tsv.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_char_p, C.c_char_p, C.c_size_t]
    L.sd_host_stage_rates.argtypes = [P(C.c_char_p), P(C.c_int64), C.c_int32, P(Params), C.c_int32, P(C.c_double)]
    L.sd_format_alt_rows.argtypes = [P(C.c_char_p), C.c_int32, C.c_void_p, P(C.c_char_p), C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, P(C.c_void_p),
                                     P(C.c_size_t)]
    L.sd_chunk_table_size.restype = C.c_int64
    L.sd_chunk_table_size.argtypes = [P(C.c_int64), C.c_int32, C.c_int32, C.c_int32]
    L.sd_decompose_chunk_range.argtypes = [P(C.c_char_p), P(C.c_int64), C.c_int32, P(C.c_char_p), P(C.c_int32),
                                           C.c_int32, P(Params), C.c_int64, C.c_int64, P(P(Rec)),
                                           P(P(C.c_int64)), C.c_char_p, C.c_size_t]
    L.sd_assemble_tsv.argtypes = [P(C.c_char_p), P(C.c_int64), C.c_int32, P(C.c_char_p), C.c_int32, P(Params),
                                  C.c_void_p, C.c_void_p, C.c_int64, P(C.c_void_p), P(C.c_size_t),
                                  C.c_char_p, C.c_size_t]
    L.sd_stream_create.argtypes = [P(C.c_void_p), P(Params), P(C.c_char_p), P(C.c_int32), C.c_int32, C.c_int32,
                                   C.c_char_p, C.c_size_t]
    L.sd_stream_destroy.argtypes = [C.c_void_p]
    L.sd_stream_submit.argtypes = [C.c_void_p, P(C.c_char_p), P(C.c_int64), C.c_int32, C.c_char_p, C.c_size_t]
    L.sd_stream_collect.argtypes = [C.c_void_p, P(P(Rec)), P(P(C.c_int64)), P(C.c_int64), C.c_char_p, C.c_size_t]
    L.sd_stream_stats.argtypes = [C.c_void_p, P(C.c_double)]
    L.sd_stream_info.argtypes = [C.c_void_p, P(C.c_int64)]
    L.sd_pack_bases.restype = C.c_int32
    L.sd_pack_bases.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.sd_write_records.argtypes = [C.c_char_p, P(Params), P(C.c_char_p), C.c_int32, P(C.c_char_p), C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.sd_read_records.argtypes = [C.c_char_p, P(Records), C.c_char_p, C.c_size_t]
    L.sd_records_free.argtypes = [P(Records)]
    L.sd_records_free.restype = None
    L.sd_records_to_raw_tsv.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_size_t]
    L.sd_decompose_files_records.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_char_p, C.c_char_p, C.c_size_t]
    L.sd_run_files_records.argtypes = [C.c_char_p, C.c_char_p, P(Params), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                       C.c_int32, C.c_int32, P(C.c_double), C.c_char_p, C.c_size_t]
    _lib = L
    return L


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


def _strs(seq):
    arr = (C.c_char_p * max(len(seq), 1))()
    for i, s in enumerate(seq):
        arr[i] = _b(s)
    return arr


FLAG_NO_F16, FLAG_FULL_FLOOR, FLAG_NO_EDTHR_COMPACT, FLAG_FILTER_GENERAL, FLAG_NO_STREAM_IDENT, FLAG_PROGRESS = 1, 2, 4, 8, 16, 32
FLAG_TRACE_V1 = 64
FLAG_NO_IDENT_PRUNE = 256   # --second-best: every homopolymer-compressed pair aligned in full (no distance-only pruning)
FLAG_NO_U16 = 128     # narrow layout: fp16 / int16 cells as in rounds 1-5 instead of the biased-u16 format


def make_params(scoring=(-1, -1, -1, 1), part_size=5000, overlap=500, ed_thr=-1, threads=1,
                device=0, kernel=KERNEL_AUTO, max_batch_rows=0, pipe_mode=None, flags=0, f16_guard=0):
    """sd_params; pipe_mode (None = the library default, 0 / 1 / 2), flags (FLAG_* bits) and f16_guard (magnitude
    limit of the fills' fp16 range guard, 0 = 2040) are the reserved[] switches of include/sd_hip.h."""
    L = load()
    p = Params()
    L.sd_params_default(C.byref(p))
    p.ins, p.del_, p.mismatch, p.match = [int(x) for x in scoring]
    p.part_size, p.overlap, p.ed_thr = int(part_size), int(overlap), int(ed_thr)
    p.threads, p.device, p.kernel = int(threads), int(device), int(kernel)
    p.max_batch_rows = int(max_batch_rows)
    p.reserved[0] = 0 if pipe_mode is None else int(pipe_mode) + 1
    p.reserved[1] = int(flags)
    p.reserved[2] = int(f16_guard)
    return p


def guard_trips():
    """Batches of this process repeated with integer cells because the fp16 range guard tripped (sd_guard_trips)."""
    L = load()
    L.sd_guard_trips.restype = C.c_int64
    return int(L.sd_guard_trips())


def plan_info(mono_seqs, **kw):
    """The layout the fast kernel family would use for this monomer set and scoring (host only, no GPU needed):
    {"family", "cells_per_lane", "cells", "floor_slots", "waves", "range_bound", "min_first_lane_cells", "max_lane_cells",
    "score_factor", "trace_regs" (0 = one-block int32 traceback), "trace_bound", "bperm_scan", "why"} -- family "generic"
    carries the reason in "why"."""
    L = load()
    p = make_params(**kw)
    ms = [_b(s) for s in mono_seqs]
    ml = (C.c_int32 * max(len(ms), 1))(*[len(s) for s in ms])
    v = (C.c_int64 * 8)()
    err = C.create_string_buffer(4096)
    rc = L.sd_plan_info(C.byref(p), _strs(ms), ml, len(ms), v, err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    cells = {0: "int32", 1: "int16", 2: "f16", 9: "u16", 3: "int16/int8-table", 4: "f16/bf8-table", 5: "f16/bf8-codes x waves", 6: "f16/bf8-codes tiled x waves",
             7: "int16/int8-codes x waves", 8: "int16/int8-codes tiled x waves"}
    return {"family": {1: "generic", 2: "fast"}[v[0]], "cells_per_lane": v[1], "cells": cells.get(v[2], "?") if v[0] == 2 else "int32",
            "floor_slots": v[3], "waves": v[4] & 0xff, "range_bound": (v[4] >> 8) & 0xffffffff, "rebase": v[4] >> 40, "min_first_lane_cells": v[5], "max_lane_cells": v[6],
            "score_factor": v[7] & 0xffff, "trace_regs": (v[7] >> 16) & 0xff, "trace_bound": (v[7] >> 24) & 0xffffffff,
            "bperm_scan": bool((v[7] >> 56) & 1), "why": err.value.decode(errors="replace") if v[0] == 1 else ""}


def release_cache():
    """Return the library's cached device buffers to the driver."""
    load().sd_release_cache()


def device_count():
    return load().sd_device_count()


def decompose_files(reads_fa, monomers_fa, raw_tsv_out, **kw):
    """reads.fa + monomers.fa -> raw TSV file (the bytes `dp` would print on stdout)."""
    L = load()
    p = make_params(**kw)
    err = C.create_string_buffer(4096)
    rc = L.sd_decompose_files(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p),
                              os.fsencode(raw_tsv_out), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def run_files(reads_fa, monomers_fa, raw_tsv_out, final_tsv_out, alt_tsv_out, min_identity=0, second_best=False,
              lr_coef=(-31.48494996, 0.41784018, 0.69186882), records_out=None, **kw):
    """The whole CLI job natively (sd_run_files): raw, final and _alt TSV files from the two FASTA files; with
    records_out also the binary record stream of the raw rows (sd_run_files_records)."""
    L = load()
    p = make_params(**kw)
    err = C.create_string_buffer(4096)
    coef = (C.c_double * 3)(*[float(x) for x in lr_coef])
    if records_out is None:
        rc = L.sd_run_files(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), os.fsencode(raw_tsv_out),
                            os.fsencode(final_tsv_out), os.fsencode(alt_tsv_out), int(min_identity), 1 if second_best else 0,
                            coef, err, 4096)
    else:
        rc = L.sd_run_files_records(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), os.fsencode(raw_tsv_out),
                                    os.fsencode(final_tsv_out), os.fsencode(alt_tsv_out), os.fsencode(records_out),
                                    int(min_identity), 1 if second_best else 0, coef, err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def decompose_files_records(reads_fa, monomers_fa, records_out, **kw):
    """reads.fa + monomers.fa -> binary record stream (sd_decompose_files_records): the rows of the raw TSV, no text."""
    L = load()
    p = make_params(**kw)
    err = C.create_string_buffer(4096)
    rc = L.sd_decompose_files_records(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p),
                                      os.fsencode(records_out), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def write_records(path, tmpl_names, read_names, read_lens, rows, row_off, **kw):
    """sd_write_records (host only): rows = structured array / sequence of (tmpl, start, end, score), row_off[n_reads + 1]."""
    import numpy as np
    L = load()
    p = make_params(**kw)
    r = np.ascontiguousarray(np.asarray(rows, dtype=_rec_dtype()) if not isinstance(rows, np.ndarray) else rows, dtype=_rec_dtype())
    o = np.ascontiguousarray(row_off, dtype=np.int64)
    rl = None if read_lens is None else np.ascontiguousarray(read_lens, dtype=np.int64)
    err = C.create_string_buffer(4096)
    rc = L.sd_write_records(os.fsencode(path), C.byref(p), _strs([_b(t) for t in tmpl_names]), len(tmpl_names),
                            _strs([_b(x) for x in read_names]), None if rl is None else rl.ctypes.data, len(read_names),
                            r.ctypes.data, o.ctypes.data, err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def read_records(path):
    """sd_read_records (host only) -> dict(params, templates, reads, read_lens, row_off, rows[structured array])."""
    import numpy as np
    L = load()
    out = Records()
    err = C.create_string_buffer(4096)
    rc = L.sd_read_records(os.fsencode(path), C.byref(out), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    try:
        n, nr = out.n_reads, out.n_rows
        res = {"params": {"scoring": (out.ins, out.del_, out.mismatch, out.match), "part_size": out.part_size,
                          "overlap": out.overlap, "ed_thr": out.ed_thr},
               "templates": [out.tmpl_names[t].decode() for t in range(out.n_templates)],
               "reads": [out.read_names[r].decode() for r in range(n)],
               "read_lens": [int(out.read_lens[r]) for r in range(n)],
               "row_off": np.ctypeslib.as_array(out.row_off, shape=(n + 1,)).copy(),
               "rows": (np.frombuffer(C.string_at(out.rows, nr * C.sizeof(Rec)), dtype=_rec_dtype()).copy() if nr
                        else np.zeros(0, dtype=_rec_dtype()))}
    finally:
        L.sd_records_free(C.byref(out))
    return res


def records_to_raw_tsv(records_path, raw_tsv_out, threads=1):
    """sd_records_to_raw_tsv (host only): the raw TSV `dp` prints for the rows of a record stream."""
    L = load()
    err = C.create_string_buffer(4096)
    rc = L.sd_records_to_raw_tsv(os.fsencode(records_path), os.fsencode(raw_tsv_out), int(threads), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def last_run_stats():
    """Stage times of the last run_files / run_files_range call of this process (sd_last_run_stats)."""
    L = load()
    v = (C.c_double * 24)()
    L.sd_last_run_stats(v)
    keys = ("fill_ms", "trace_ms", "compact_ms", "ident_ms", "ident_pairs", "batches", "rows", "pack_ms", "wait_ms",
            "raw_text_ms", "post_ms", "io_ms", "text_identity_ms", "final_text_ms", "total_ms", "alloc_ms", "setup_ms",
            "assemble_ms", "homo_pairs", "homo_full_pairs")
    return dict(zip(keys, [float(x) for x in v]))


def run_files_range(reads_fa, monomers_fa, rank, world, raw_tsv_out, final_tsv_out, alt_tsv_out, min_identity=0,
                    second_best=False, lr_coef=(-31.48494996, 0.41784018, 0.69186882), **kw):
    """One rank's group of reads of a multi-process launch, completely (sd_run_files_range) -> (first read, one
    past the last read, reads in the file, chunks of this rank).  SdError(SD_ERR_UNSUPPORTED) when the read set
    cannot be split by reads."""
    L = load()
    p = make_params(**kw)
    err = C.create_string_buffer(4096)
    coef = (C.c_double * 3)(*[float(x) for x in lr_coef])
    info = (C.c_int64 * 4)()
    rc = L.sd_run_files_range(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), int(rank), int(world),
                              os.fsencode(raw_tsv_out), os.fsencode(final_tsv_out), os.fsencode(alt_tsv_out),
                              int(min_identity), 1 if second_best else 0, coef, info, err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    return tuple(info)


def convert_raw_tsv(raw_tsv, reads_fa, monomers_fa, final_tsv_out, alt_tsv_out, min_identity=0, second_best=False,
                    lr_coef=(-31.48494996, 0.41784018, 0.69186882), device=-1, threads=1):
    """convert_tsv (main.py:168-184) natively; device=-1: host identities, else the HIP kernel."""
    L = load()
    err = C.create_string_buffer(4096)
    coef = (C.c_double * 3)(*[float(x) for x in lr_coef])
    rc = L.sd_convert_raw_tsv(os.fsencode(raw_tsv), os.fsencode(reads_fa), os.fsencode(monomers_fa),
                              os.fsencode(final_tsv_out), os.fsencode(alt_tsv_out), int(min_identity),
                              1 if second_best else 0, coef, int(device), int(threads), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def convert_raw_tsv_range(raw_tsv, reads_fa, monomers_fa, final_tsv_out, alt_tsv_out, rank, world, min_identity=0,
                          second_best=False, lr_coef=(-31.48494996, 0.41784018, 0.69186882), device=-1, threads=1):
    """convert_tsv on this rank's byte range of the raw TSV (cut at line starts) into this rank's part files."""
    L = load()
    err = C.create_string_buffer(4096)
    coef = (C.c_double * 3)(*[float(x) for x in lr_coef])
    rc = L.sd_convert_raw_tsv_range(os.fsencode(raw_tsv), os.fsencode(reads_fa), os.fsencode(monomers_fa),
                                    os.fsencode(final_tsv_out), os.fsencode(alt_tsv_out), int(min_identity),
                                    1 if second_best else 0, coef, int(device), int(threads), int(rank), int(world), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


def decompose(read_names, read_seqs, mono_names, mono_seqs, **kw):
    """In-memory variant -> raw TSV bytes."""
    L = load()
    p = make_params(**kw)
    err = C.create_string_buffer(4096)
    rs = [_b(s) for s in read_seqs]
    ms = [_b(s) for s in mono_seqs]
    rl = (C.c_int64 * max(len(rs), 1))(*[len(s) for s in rs])
    ml = (C.c_int32 * max(len(ms), 1))(*[len(s) for s in ms])
    out = C.c_void_p()
    ln = C.c_size_t()
    rc = L.sd_decompose(_strs(read_names), _strs(rs), rl, len(rs), _strs(mono_names), _strs(ms), ml,
                        len(ms), C.byref(p), C.byref(out), C.byref(ln), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    data = C.string_at(out, ln.value)
    L.sd_free(out)
    return data


class Engine:
    """Device-resident batches: create (templates) -> load_reads -> run -> fetch -> assemble."""

    def __init__(self, mono_seqs, **kw):
        self.L = load()
        self.params = make_params(**kw)
        self._err = C.create_string_buffer(4096)
        ms = [_b(s) for s in mono_seqs]
        ml = (C.c_int32 * max(len(ms), 1))(*[len(s) for s in ms])
        self.h = C.c_void_p()
        rc = self.L.sd_engine_create(C.byref(self.h), C.byref(self.params), _strs(ms), ml, len(ms),
                                     self._err, 4096)
        self._check(rc)
        self.n_chunks = 0
        self.n_reads = 0

    def _check(self, rc):
        if rc != SD_OK:
            raise SdError(rc, self._err.value.decode(errors="replace"))

    def close(self):
        if self.h:
            self.L.sd_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_reads(self, read_seqs):
        rs = [_b(s) for s in read_seqs]
        self._keep = rs
        rl = (C.c_int64 * max(len(rs), 1))(*[len(s) for s in rs])
        n = C.c_int64()
        self._check(self.L.sd_engine_load_reads(self.h, _strs(rs), rl, len(rs), C.byref(n), self._err, 4096))
        self.n_chunks = n.value
        self.n_reads = len(rs)
        return n.value

    def run(self, stream=None):
        self._check(self.L.sd_engine_run(self.h, C.c_void_p(stream or 0), self._err, 4096))

    def fetch_raw(self):
        recs = C.POINTER(Rec)()
        off = C.POINTER(C.c_int64)()
        self._check(self.L.sd_engine_fetch(self.h, C.byref(recs), C.byref(off), self._err, 4096))
        return recs, off

    def fetch(self):
        """-> list over chunks of [(tmpl, start, end, score), ...] (chunk-local coordinates)."""
        recs, off = self.fetch_raw()
        out = []
        for c in range(self.n_chunks):
            out.append([(recs[x].tmpl, recs[x].start, recs[x].end, recs[x].score)
                        for x in range(off[c], off[c + 1])])
        self.L.sd_free(recs)
        self.L.sd_free(off)
        return out

    def rows(self):
        """run results assembled per read -> list over reads of [(tmpl, start, end, score), ...]."""
        recs, off = self.fetch_raw()
        rows = C.POINTER(Rec)()
        roff = C.POINTER(C.c_int64)()
        self._check(self.L.sd_engine_assemble(self.h, recs, off, C.byref(rows), C.byref(roff), self._err, 4096))
        out = []
        for r in range(self.n_reads):
            out.append([(rows[x].tmpl, rows[x].start, rows[x].end, rows[x].score)
                        for x in range(roff[r], roff[r + 1])])
        for p in (recs, off, rows, roff):
            self.L.sd_free(p)
        return out

    def total_rows(self):
        """Number of assembled rows (cheap check used by bench.py)."""
        recs, off = self.fetch_raw()
        n = off[self.n_chunks]
        self.L.sd_free(recs)
        self.L.sd_free(off)
        return n

    def timings(self):
        ms = (C.c_float * 4)()
        rc = self.L.sd_engine_timings(self.h, ms)
        if rc != SD_OK:
            raise SdError(rc, "sd_engine_timings")
        return {"fill_ms": ms[0], "trace_ms": ms[1], "compact_ms": ms[2], "run_ms": ms[3]}

    def info(self):
        v = (C.c_int64 * 8)()
        self.L.sd_engine_info(self.h, v)
        return _info_dict(v)


def _info_dict(v):
    return {"n_templates": v[0], "sum_template_len": v[1], "n_chunks": v[2], "rows": v[3],
            "family": {1: "generic", 2: "fast"}.get(v[4] & 0xff, "?"),
            "cells": {0: "int32", 1: "int16", 2: "f16", 9: "u16", 3: "int16/int8-table", 4: "f16/bf8-table",
                      5: "f16/bf8-codes x waves", 6: "f16/bf8-codes tiled x waves", 7: "int16/int8-codes x waves",
                      8: "int16/int8-codes tiled x waves"}.get(v[4] >> 8, "?"),
            "cells_per_lane": v[5] if (v[4] & 0xff) == 1 else v[5] & 0xffff,
            "floor_slots": 0 if (v[4] & 0xff) == 1 else v[5] >> 16,
            "workspace_bytes": v[6], "fill_launches": v[7] & 0xffff,
            "trace": {0: "generic", 1: "one-block int32", 2: "two-block packed16"}.get(v[7] >> 16, "?")}


class ReadSet:
    """Sequences in host memory, as C arrays (built once, reusable across submits)."""

    def __init__(self, read_seqs):
        self.seqs = [_b(s) for s in read_seqs]
        self.n = len(self.seqs)
        self.ptrs = _strs(self.seqs)
        self.lens = (C.c_int64 * max(self.n, 1))(*[len(s) for s in self.seqs])
        self.bp = sum(len(s) for s in self.seqs)


class Stream:
    """Pipelined sequences-in-host-memory -> rows-in-host-memory path (sd_stream_*): submit() read sets,
    collect() their rows in FIFO order; two device batches are in flight across job boundaries."""

    def __init__(self, mono_seqs, sub_batches=1, **kw):
        self.L = load()
        self.params = make_params(**kw)
        self._err = C.create_string_buffer(4096)
        ms = [_b(s) for s in mono_seqs]
        ml = (C.c_int32 * max(len(ms), 1))(*[len(s) for s in ms])
        self.h = C.c_void_p()
        self._check(self.L.sd_stream_create(C.byref(self.h), C.byref(self.params), _strs(ms), ml, len(ms),
                                            int(sub_batches), self._err, 4096))
        self._n_reads = []

    def _check(self, rc):
        if rc != SD_OK:
            raise SdError(rc, self._err.value.decode(errors="replace"))

    def close(self):
        if self.h:
            self.L.sd_stream_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, reads):
        rs = reads if isinstance(reads, ReadSet) else ReadSet(reads)
        self._check(self.L.sd_stream_submit(self.h, rs.ptrs, rs.lens, rs.n, self._err, 4096))
        self._n_reads.append(rs.n)

    def collect(self, as_lists=False):
        """Rows of the oldest job: (n_rows,) by default -- the arrays are freed at once -- or, with
        as_lists, a list over reads of [(tmpl, start, end, score), ...]."""
        rows = C.POINTER(Rec)()
        off = C.POINTER(C.c_int64)()
        n = C.c_int64()
        self._check(self.L.sd_stream_collect(self.h, C.byref(rows), C.byref(off), C.byref(n), self._err, 4096))
        nr = self._n_reads.pop(0)
        out = n.value
        if as_lists:
            import numpy as np
            o = np.ctypeslib.as_array(off, shape=(nr + 1,)).copy()
            if n.value:
                r = np.frombuffer(C.string_at(rows, n.value * C.sizeof(Rec)), dtype=_rec_dtype())
            else:
                r = np.zeros(0, dtype=_rec_dtype())
            out = [[(int(x["tmpl"]), int(x["start"]), int(x["end"]), int(x["score"])) for x in r[o[i]:o[i + 1]]]
                   for i in range(nr)]
        self.L.sd_free(rows)
        self.L.sd_free(off)
        return out

    DEPTH = 2   # jobs outstanding before the oldest is collected: all three engines of the pipeline have a batch then

    def imap(self, jobs, as_lists=False, depth=None):
        """Rows of every job of the iterable `jobs` (read lists / ReadSets), in order, with `depth` later jobs submitted
        before a job is collected -- the order of calls that keeps the device busy (sd_hip.h at sd_stream_create: the
        traceback of a batch ends with the fill of the next one, so with only ONE job outstanding the job after that is
        enqueued late; bench.py's timed loop is this generator)."""
        depth = self.DEPTH if depth is None else max(0, int(depth))
        out = 0
        for reads in jobs:
            self.submit(reads)
            out += 1
            if out > depth:
                out -= 1
                yield self.collect(as_lists=as_lists)
        while out > 0:
            out -= 1
            yield self.collect(as_lists=as_lists)

    def stats(self):
        v = (C.c_double * 16)()
        self.L.sd_stream_stats(self.h, v)
        keys = ["fill_ms", "trace_ms", "compact_ms", "run_ms", "fill_launches", "batches", "rows", "host_pack_ms",
                "host_wait_ms", "host_assemble_ms", "submit_ms", "collect_ms", "jobs", "sub_batches", "row_budget"]
        return dict(zip(keys, list(v)[:15]))

    def info(self):
        v = (C.c_int64 * 8)()
        self.L.sd_stream_info(self.h, v)
        return _info_dict(v)


def host_stage_rates(reads, iters=3, **kw):
    """{"pack_bp_per_s", "assemble_format_bp_per_s", "rows_per_s", "text_bytes"} of the host stages alone."""
    L = load()
    rs = reads if isinstance(reads, ReadSet) else ReadSet(reads)
    p = make_params(**kw)
    out = (C.c_double * 4)()
    rc = L.sd_host_stage_rates(rs.ptrs, rs.lens, rs.n, C.byref(p), int(iters), out)
    if rc != SD_OK:
        raise SdError(rc, "sd_host_stage_rates")
    return {"pack_bp_per_s": out[0], "assemble_format_bp_per_s": out[1], "rows_per_s": out[2], "text_bytes": out[3]}


def write_parts_selftest(path, n_parts, part_bytes, threads=4, fail_reserve=False):
    """sd::write_parts alone (the file writer of sd_run_files); returns (bytes written, on tmpfs)."""
    L = load()
    L.sd_write_parts_selftest.restype = C.c_int
    L.sd_write_parts_selftest.argtypes = [C.c_char_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    out = (C.c_int64 * 2)()
    rc = L.sd_write_parts_selftest(_b(path), n_parts, part_bytes, threads, 1 if fail_reserve else 0, out)
    if rc != SD_OK:
        raise SdError(rc, "sd_write_parts_selftest")
    return int(out[0]), bool(out[1])


def pack_bases(seq):
    """(words uint32[(n+15)//16], nmask uint32[(n+31)//32], has_n) exactly as the device reads a chunk."""
    import numpy as np
    L = load()
    b = _b(seq)
    w = np.zeros((len(b) + 15) // 16, dtype=np.uint32)
    m = np.zeros((len(b) + 31) // 32, dtype=np.uint32)
    rc = L.sd_pack_bases(b, len(b), w.ctypes.data, m.ctypes.data)
    if rc < 0:
        raise SdError(SD_ERR_PARAM, "sd_pack_bases")
    return w, m, bool(rc)


def format_rows(read_name, tmpl_names, rows):
    L = load()
    arr = (Rec * max(len(rows), 1))()
    for i, r in enumerate(rows):
        arr[i] = Rec(*[int(x) for x in r])
    out = C.c_void_p()
    ln = C.c_size_t()
    rc = L.sd_format_rows(_b(read_name), _strs(tmpl_names), arr, len(rows), C.byref(out), C.byref(ln))
    if rc != SD_OK:
        raise SdError(rc, "sd_format_rows")
    data = C.string_at(out, ln.value)
    L.sd_free(out)
    return data


def chunk_plan(length, part=5000, overlap=500):
    L = load()
    n = L.sd_chunk_plan(length, part, overlap, None, None, 0)
    off = (C.c_int64 * max(n, 1))()
    ln = (C.c_int32 * max(n, 1))()
    L.sd_chunk_plan(length, part, overlap, off, ln, n)
    return [(off[i], ln[i]) for i in range(n)]


def seam_merge(recs):
    L = load()
    arr = (Rec * max(len(recs), 1))()
    for i, r in enumerate(recs):
        arr[i] = Rec(*[int(x) for x in r])
    m = L.sd_seam_merge(arr, len(recs))
    return [(arr[i].tmpl, arr[i].start, arr[i].end, arr[i].score) for i in range(m)]


def fasta_load(path):
    """-> (names, seqs, has_n) with the reference binary's FASTA semantics (main.cpp:314-346)."""
    L = load()
    f = Fasta()
    err = C.create_string_buffer(4096)
    rc = L.sd_fasta_load(os.fsencode(path), C.byref(f), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    names = [f.names[i].decode() for i in range(f.n)]
    seqs = [C.string_at(f.seqs[i], f.lens[i]) for i in range(f.n)]
    has_n = bool(f.has_n)
    L.sd_fasta_free(C.byref(f))
    return names, seqs, has_n


def nw_identity_batch(queries, targets, threads=1):
    """[(dist, matches, columns)] of unit-cost NW alignments (main.py:29-60 semantics)."""
    L = load()
    n = len(queries)
    q = [_b(s) for s in queries]
    t = [_b(s) for s in targets]
    ql = (C.c_int32 * max(n, 1))(*[len(s) for s in q])
    tl = (C.c_int32 * max(n, 1))(*[len(s) for s in t])
    d = (C.c_int32 * max(n, 1))()
    m = (C.c_int32 * max(n, 1))()
    c = (C.c_int32 * max(n, 1))()
    rc = L.sd_nw_identity_batch(_strs(q), ql, _strs(t), tl, n, threads, d, m, c)
    if rc != SD_OK:
        raise SdError(rc, "sd_nw_identity_batch")
    return [(d[i], m[i], c[i]) for i in range(n)]


def identity_segments(seq, starts, ends, templates, homo=False, threads=1, pair_tmpl=None, device=None):
    """(dist, matches, columns) int32 arrays of shape [n_segments, n_templates]: every segment
    seq[starts[s] .. ends[s]] (inclusive) against every template (main.py:107-150 all-vs-all);
    with pair_tmpl (template index per segment) arrays of shape [n_segments]: that pair only.
    device=None: the host implementation (sd_identity_segments); device=<ordinal>: the HIP kernel
    (sd_identity_segments_dev), with the host implementation for input the kernel does not take."""
    import numpy as np
    L = load()
    sb = _b(seq)
    st = np.ascontiguousarray(starts, dtype=np.int64)
    en = np.ascontiguousarray(ends, dtype=np.int64)
    n = int(st.shape[0])
    tb = [_b(t) for t in templates]
    T = len(tb)
    tl = (C.c_int32 * max(T, 1))(*[len(t) for t in tb])
    pt = None if pair_tmpl is None else np.ascontiguousarray(pair_tmpl, dtype=np.int32)
    shape = (n, T) if pt is None else (n,)
    d = np.zeros(shape, dtype=np.int32)
    m = np.zeros(shape, dtype=np.int32)
    c = np.zeros(shape, dtype=np.int32)
    tarr = _strs(tb)
    rc = SD_ERR_UNSUPPORTED
    if device is not None:
        rc = L.sd_identity_segments_dev(sb, len(sb), st.ctypes.data, en.ctypes.data, n, tarr, tl, T,
                                        None if pt is None else pt.ctypes.data, 1 if homo else 0, int(device),
                                        threads, d.ctypes.data, m.ctypes.data, c.ctypes.data)
        if rc not in (SD_OK, SD_ERR_UNSUPPORTED):
            raise SdError(rc, "sd_identity_segments_dev")
    if rc == SD_ERR_UNSUPPORTED:
        rc = L.sd_identity_segments(sb, len(sb), st.ctypes.data, en.ctypes.data, n, tarr, tl, T,
                                    None if pt is None else pt.ctypes.data, 1 if homo else 0, threads,
                                    d.ctypes.data, m.ctypes.data, c.ctypes.data)
    if rc != SD_OK:
        raise SdError(rc, "sd_identity_segments")
    return d, m, c


# ---- chunk-range form (one job over several GPUs, one process per GPU; see shard.py) ---------------
def _rec_dtype():
    import numpy as np
    return np.dtype([("tmpl", np.int32), ("start", np.int32), ("end", np.int32), ("score", np.int32)])


def chunk_table_size(read_lens, part_size=5000, overlap=500):
    L = load()
    arr = (C.c_int64 * max(len(read_lens), 1))(*[int(x) for x in read_lens])
    return L.sd_chunk_table_size(arr, len(read_lens), part_size, overlap)


def decompose_chunk_range(read_seqs, mono_seqs, chunk_lo, chunk_hi, **kw):
    """Records of the chunks [chunk_lo, chunk_hi) of the global chunk table of `read_seqs`:
    (recs structured array [tmpl, start, end, score], rec_off int64[chunk_hi - chunk_lo + 1])."""
    import numpy as np
    L = load()
    p = make_params(**kw)
    rs = [_b(s) for s in read_seqs]
    ms = [_b(s) for s in mono_seqs]
    rl = (C.c_int64 * max(len(rs), 1))(*[len(s) for s in rs])
    ml = (C.c_int32 * max(len(ms), 1))(*[len(s) for s in ms])
    recs = C.POINTER(Rec)()
    off = C.POINTER(C.c_int64)()
    err = C.create_string_buffer(4096)
    rc = L.sd_decompose_chunk_range(_strs(rs), rl, len(rs), _strs(ms), ml, len(ms), C.byref(p), chunk_lo,
                                    chunk_hi, C.byref(recs), C.byref(off), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    n = chunk_hi - chunk_lo
    o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
    nrec = int(o[n])
    if nrec:   # one copy out of the library's buffer
        r = np.ctypeslib.as_array(C.cast(recs, C.POINTER(C.c_int32)), shape=(nrec * 4,)).copy().view(_rec_dtype())
    else:
        r = np.zeros(0, dtype=_rec_dtype())
    L.sd_free(recs)
    L.sd_free(off)
    return r, o


def decompose_files_range(reads_fa, monomers_fa, rank, world, **kw):
    """A rank's share of a sharded job straight from the FASTA files (mapped, not copied):
    (recs, rec_off, chunk_lo, chunk_hi, n_chunks_total)."""
    import numpy as np
    L = load()
    p = make_params(**kw)
    recs = C.POINTER(Rec)()
    off = C.POINTER(C.c_int64)()
    lo, hi, tot = C.c_int64(), C.c_int64(), C.c_int64()
    err = C.create_string_buffer(4096)
    rc = L.sd_decompose_files_range(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), rank, world,
                                    C.byref(recs), C.byref(off), C.byref(lo), C.byref(hi), C.byref(tot), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    n = hi.value - lo.value
    o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
    nrec = int(o[n])
    r = (np.frombuffer(C.string_at(recs, nrec * C.sizeof(Rec)), dtype=_rec_dtype()).copy() if nrec
         else np.zeros(0, dtype=_rec_dtype()))
    L.sd_free(recs)
    L.sd_free(off)
    return r, o, lo.value, hi.value, tot.value


def assemble_files_tsv(reads_fa, monomers_fa, recs, rec_off, raw_tsv_out, **kw):
    """Rank 0: gathered records of all chunks -> raw TSV file (names / lengths from the FASTA index)."""
    import numpy as np
    L = load()
    p = make_params(**kw)
    r = np.ascontiguousarray(recs, dtype=_rec_dtype())
    o = np.ascontiguousarray(rec_off, dtype=np.int64)
    err = C.create_string_buffer(4096)
    rc = L.sd_assemble_files_tsv(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), r.ctypes.data,
                                 o.ctypes.data, len(o) - 1, os.fsencode(raw_tsv_out), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))


class SeamEdge(C.Structure):
    """sd_seam_edge (include/sd_hip.h): what a rank publishes about the records either side of its range boundaries."""
    _fields_ = [("ok", C.c_int32), ("has_front", C.c_int32), ("has_back", C.c_int32), ("through", C.c_int32),
                ("head", (C.c_int32 * 2) * 8), ("tail", (C.c_int32 * 2) * 8), ("exit_of", C.c_int8 * 8),
                ("reserved", C.c_int64)]


class RangeAssembler:
    """One rank's part of the raw TSV of a job sharded by chunk range (sd_range_assemble_*, include/sd_hip.h):
         a = RangeAssembler.from_files(reads_fa, monomers_fa, rank, world, recs, rec_off, **params)   # or .from_lists
         edges = <all-gather of a.edge (bytes)>
         nbytes = a.text(edges)            # SdError(SD_ERR_UNSUPPORTED): gather on rank 0 instead
         a.write(path, offset) / a.bytes()
    Host only; the concatenation of the ranks' texts equals assemble_tsv of all records."""

    def __init__(self, handle, edge, keep):
        self._h = handle
        self._keep = keep
        self.edge = bytes(edge)
        self.nbytes = None

    @staticmethod
    def _bind(L):
        L.sd_range_assemble_begin.restype = C.c_int
        L.sd_range_assemble_begin_files.restype = C.c_int
        L.sd_range_assemble_text.restype = C.c_int
        L.sd_range_assemble_write.restype = C.c_int
        L.sd_range_assemble_copy.restype = C.c_int
        L.sd_range_assemble_stats.restype = None
        L.sd_range_assemble_free.restype = None
        L.sd_range_assemble_free.argtypes = [C.c_void_p]
        L.sd_range_assemble_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.sd_range_assemble_text.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_char_p, C.c_size_t]
        L.sd_range_assemble_write.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_size_t]
        L.sd_decompose_files_range_begin.restype = C.c_int
        L.sd_range_assemble_records.restype = C.c_int
        L.sd_range_assemble_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]

    @classmethod
    def from_files(cls, reads_fa, monomers_fa, rank, world, recs, rec_off, **kw):
        import numpy as np
        L = load()
        cls._bind(L)
        p = make_params(**kw)
        r = np.ascontiguousarray(recs, dtype=_rec_dtype())
        o = np.ascontiguousarray(rec_off, dtype=np.int64)
        edge, h = SeamEdge(), C.c_void_p()
        err = C.create_string_buffer(4096)
        rc = L.sd_range_assemble_begin_files(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), C.c_int32(rank),
                                             C.c_int32(world), C.c_void_p(r.ctypes.data), C.c_void_p(o.ctypes.data),
                                             C.byref(edge), C.byref(h), err, C.c_size_t(4096))
        if rc != SD_OK:
            raise SdError(rc, err.value.decode(errors="replace"))
        return cls(h, edge, (r, o))

    @classmethod
    def run_files(cls, reads_fa, monomers_fa, rank, world, **kw):
        """DP of this rank's share + step 1 in one library call (sd_decompose_files_range_begin): the records stay in the
        library.  Sets .chunk_lo / .chunk_hi / .n_chunks; .records() copies the records out for the gather fall-back."""
        L = load()
        cls._bind(L)
        p = make_params(**kw)
        edge, h = SeamEdge(), C.c_void_p()
        lo, hi, tot = C.c_int64(), C.c_int64(), C.c_int64()
        err = C.create_string_buffer(4096)
        rc = L.sd_decompose_files_range_begin(os.fsencode(reads_fa), os.fsencode(monomers_fa), C.byref(p), C.c_int32(rank),
                                              C.c_int32(world), C.byref(edge), C.byref(h), C.byref(lo), C.byref(hi),
                                              C.byref(tot), err, C.c_size_t(4096))
        if rc != SD_OK:
            raise SdError(rc, err.value.decode(errors="replace"))
        a = cls(h, edge, None)
        a.chunk_lo, a.chunk_hi, a.n_chunks = lo.value, hi.value, tot.value
        return a

    def records(self):
        """(recs, rec_off) of the share held by the library (copies)."""
        import numpy as np
        L = load()
        recs, off, n = C.POINTER(Rec)(), C.POINTER(C.c_int64)(), C.c_int64()
        rc = L.sd_range_assemble_records(self._h, C.byref(recs), C.byref(off), C.byref(n))
        if rc != SD_OK:
            raise SdError(rc, "this assembler does not hold its records")
        o = np.ctypeslib.as_array(off, shape=(n.value + 1,)).copy()
        nrec = int(o[n.value])
        r = (np.frombuffer(C.string_at(recs, nrec * C.sizeof(Rec)), dtype=_rec_dtype()).copy() if nrec
             else np.zeros(0, dtype=_rec_dtype()))
        return r, o

    @classmethod
    def from_lists(cls, read_names, read_lens, mono_names, chunk_lo, chunk_hi, recs, rec_off, **kw):
        import numpy as np
        L = load()
        cls._bind(L)
        p = make_params(**kw)
        rn = [_b(s) for s in read_names]
        mn = [_b(s) for s in mono_names]
        rl = (C.c_int64 * max(len(rn), 1))(*[int(x) for x in read_lens])
        r = np.ascontiguousarray(recs, dtype=_rec_dtype())
        o = np.ascontiguousarray(rec_off, dtype=np.int64)
        edge, h = SeamEdge(), C.c_void_p()
        err = C.create_string_buffer(4096)
        rc = L.sd_range_assemble_begin(_strs(rn), rl, C.c_int32(len(rn)), _strs(mn), C.c_int32(len(mn)), C.byref(p),
                                       C.c_int64(chunk_lo), C.c_int64(chunk_hi), C.c_void_p(r.ctypes.data),
                                       C.c_void_p(o.ctypes.data), C.byref(edge), C.byref(h), err, C.c_size_t(4096))
        if rc != SD_OK:
            raise SdError(rc, err.value.decode(errors="replace"))
        return cls(h, edge, (r, o))

    def text(self, edges, rank):
        """edges: the edges of all ranks in rank order (bytes each).  Returns the bytes of this rank's text."""
        L = load()
        arr = (SeamEdge * len(edges))()
        for k, e in enumerate(edges):
            if e is None or len(e) != C.sizeof(SeamEdge):
                raise SdError(SD_ERR_UNSUPPORTED, "a rank published no edge")
            C.memmove(C.byref(arr[k]), e, C.sizeof(SeamEdge))
        n = C.c_int64()
        err = C.create_string_buffer(4096)
        rc = L.sd_range_assemble_text(self._h, arr, len(edges), int(rank), C.byref(n), err, 4096)
        if rc != SD_OK:
            raise SdError(rc, err.value.decode(errors="replace"))
        self.nbytes = n.value
        return n.value

    def write(self, path, offset, file_bytes=-1):
        """file_bytes >= 0: create the file if missing and set its size first (every rank passes the same total)."""
        L = load()
        err = C.create_string_buffer(4096)
        rc = L.sd_range_assemble_write(self._h, os.fsencode(path), int(offset), int(file_bytes), err, 4096)
        if rc != SD_OK:
            raise SdError(rc, err.value.decode(errors="replace"))

    def bytes(self):
        L = load()
        buf = C.create_string_buffer(max(self.nbytes, 1))
        rc = L.sd_range_assemble_copy(self._h, buf, self.nbytes)
        if rc != SD_OK:
            raise SdError(rc, "sd_range_assemble_copy")
        return buf.raw[:self.nbytes]

    def stats(self):
        L = load()
        out = (C.c_double * 8)()
        L.sd_range_assemble_stats(self._h, out)
        return {"begin_ms": out[0], "complete_reads_ms": out[1], "assumed_scan_and_text_ahead_ms": out[2],
                "text_ms": out[3], "rows_printed_after_exchange": int(out[4]), "formatted_again": bool(out[5]),
                "write_ms": out[6]}

    def close(self):
        if self._h:
            load().sd_range_assemble_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def assemble_tsv(read_names, read_lens, mono_names, recs, rec_off, **kw):
    """Raw TSV bytes from the records of all chunks in table order (host only)."""
    import numpy as np
    L = load()
    p = make_params(**kw)
    rn = [_b(s) for s in read_names]
    mn = [_b(s) for s in mono_names]
    rl = (C.c_int64 * max(len(rn), 1))(*[int(x) for x in read_lens])
    r = np.ascontiguousarray(recs, dtype=_rec_dtype())
    o = np.ascontiguousarray(rec_off, dtype=np.int64)
    out = C.c_void_p()
    ln = C.c_size_t()
    err = C.create_string_buffer(4096)
    rc = L.sd_assemble_tsv(_strs(rn), rl, len(rn), _strs(mn), len(mn), C.byref(p), r.ctypes.data,
                           o.ctypes.data, len(o) - 1, C.byref(out), C.byref(ln), err, 4096)
    if rc != SD_OK:
        raise SdError(rc, err.value.decode(errors="replace"))
    data = C.string_at(out, ln.value)
    L.sd_free(out)
    return data


def format_alt_rows(read_name, key_names, starts, ends, own_key, vals, threads=1, row_read=None):
    """Text (str) of _alt.tsv rows; vals is a [n_rows, n_keys] float64 array.  read_name: one name, or a
    list of names with row_read giving each row's index into it."""
    import numpy as np
    L = load()
    names = [read_name] if isinstance(read_name, (str, bytes)) else list(read_name)
    rr = None if row_read is None else np.ascontiguousarray(row_read, dtype=np.int32)
    v = np.ascontiguousarray(vals, dtype=np.float64)
    n, nk = (int(v.shape[0]), int(v.shape[1])) if v.ndim == 2 else (0, len(key_names))
    st = np.ascontiguousarray(starts, dtype=np.int64)
    en = np.ascontiguousarray(ends, dtype=np.int64)
    ow = np.ascontiguousarray(own_key, dtype=np.int32)
    out = C.c_void_p()
    ln = C.c_size_t()
    rc = L.sd_format_alt_rows(_strs([_b(x) for x in names]), len(names), None if rr is None else rr.ctypes.data,
                              _strs([_b(k) for k in key_names]), nk, st.ctypes.data, en.ctypes.data,
                              ow.ctypes.data, v.ctypes.data, n, threads, C.byref(out), C.byref(ln))
    if rc != SD_OK:
        raise SdError(rc, "sd_format_alt_rows")
    data = C.string_at(out, ln.value).decode()
    L.sd_free(out)
    return data
