"""ctypes binding of the CPU oracle (oracle/sd_oracle.c) and helpers around the reference binary.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under stringdecomposer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libsd_oracle.so")
REF_DP = os.path.join(HERE, "_ref", "dp")          # real reference binary (built by `make -C oracle ref`)
REF_EDLIB = os.path.join(HERE, "_ref", "libedlib.so")


class Scoring(C.Structure):
    _fields_ = [("ins", C.c_int), ("del_", C.c_int), ("mismatch", C.c_int), ("match", C.c_int)]


class Rec(C.Structure):
    _fields_ = [("tmpl", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("score", C.c_float)]


def build(force=False):
    """Compile the C restatement (and the reference binaries when /root/reference is present)."""
    if force or not os.path.isfile(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "sd_oracle.c")):
        subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/stringdecomposer/src") and \
            not (os.path.isfile(REF_DP) and os.path.isfile(REF_EDLIB)):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.sdo_align_chunk.restype = C.c_int
        L.sdo_align_chunk.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                      C.c_int, Scoring, C.POINTER(Rec), C.c_int,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int64)]
        L.sdo_postprocess.restype = C.c_int
        L.sdo_postprocess.argtypes = [C.POINTER(Rec), C.c_int]
        L.sdo_chunk_plan.restype = C.c_int
        L.sdo_chunk_plan.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int32), C.c_int]
        L.sdo_decompose_files.restype = C.c_int
        L.sdo_decompose_files.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, Scoring,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p,
                                          C.c_size_t]
        L.sdo_decompose.restype = C.c_int
        L.sdo_decompose.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int,
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_int,
                                    C.c_int, C.c_int, Scoring, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.sdo_decompose_files_ex.restype = C.c_int
        L.sdo_decompose_files_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, Scoring,
                                             C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                             C.c_char_p, C.c_size_t]
        L.sdo_decompose_ex.restype = C.c_int
        L.sdo_decompose_ex.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int,
                                       C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_int,
                                       C.c_int, C.c_int, Scoring, C.c_int, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.sdo_hw_edit_distance.restype = C.c_int
        L.sdo_hw_edit_distance.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.sdo_nw_identity.restype = C.c_int
        L.sdo_nw_identity.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
        L.sdo_reverse_complement.restype = C.c_int
        L.sdo_reverse_complement.argtypes = [C.c_char_p, C.c_int64, C.c_char_p]
        L.sdo_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oracle rc=%d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def _strs(seq):
    arr = (C.c_char_p * len(seq))()
    arr[:] = [s.encode() if isinstance(s, str) else s for s in seq]
    return arr


def scoring(s=(-1, -1, -1, 1)):
    return Scoring(int(s[0]), int(s[1]), int(s[2]), int(s[3]))


def align_chunk(read, templates, sc=(-1, -1, -1, 1), internals=False):
    """AlignPartClassicDP on one chunk -> list of (tmpl, start, end, score) [+ (B, argB, hist)]."""
    L = lib()
    n = len(read)
    T = len(templates)
    t_arr = _strs(templates)
    tl = (C.c_int * T)(*[len(t) for t in templates])
    out = (Rec * n)()
    B = (C.c_int32 * (n + 1))() if internals else None
    A = (C.c_int32 * (n + 1))() if internals else None
    H = (C.c_int64 * 4)() if internals else None
    k = L.sdo_align_chunk(read.encode() if isinstance(read, str) else read, n, t_arr, tl, T,
                          scoring(sc), out, n, B, A, H)
    if k < 0:
        raise OracleError(k, "sdo_align_chunk failed")
    recs = [(out[i].tmpl, out[i].start, out[i].end, out[i].score) for i in range(k)]
    if internals:
        return recs, list(B), list(A), list(H)
    return recs


def chunk_plan(length, part=5000, overlap=500):
    L = lib()
    cnt = L.sdo_chunk_plan(length, part, overlap, None, None, 0)
    off = (C.c_int64 * max(cnt, 1))()
    ln = (C.c_int32 * max(cnt, 1))()
    L.sdo_chunk_plan(length, part, overlap, off, ln, cnt)
    return [(off[i], ln[i]) for i in range(cnt)]


def postprocess(recs):
    L = lib()
    arr = (Rec * max(len(recs), 1))()
    for i, r in enumerate(recs):
        arr[i] = Rec(r[0], r[1], r[2], r[3])
    m = L.sdo_postprocess(arr, len(recs))
    return [(arr[i].tmpl, arr[i].start, arr[i].end, arr[i].score) for i in range(m)]


def decompose_files(reads_fa, monomers_fa, threads=1, part=5000, overlap=500, sc=(-1, -1, -1, 1),
                    ed_thr=-1):
    """Raw TSV bytes, as the reference `dp` prints on stdout."""
    L = lib()
    out = C.c_void_p()
    ln = C.c_size_t()
    err = C.create_string_buffer(2048)
    rc = L.sdo_decompose_files_ex(os.fsencode(reads_fa), os.fsencode(monomers_fa), threads, part,
                                  overlap, scoring(sc), ed_thr, C.byref(out), C.byref(ln), err, 2048)
    if rc != 0:
        raise OracleError(rc, err.value.decode(errors="replace"))
    data = C.string_at(out, ln.value)
    L.sdo_free(out)
    return data


def decompose(read_names, read_seqs, mono_names, mono_seqs, threads=1, part=5000, overlap=500,
              sc=(-1, -1, -1, 1), ed_thr=-1):
    L = lib()
    out = C.c_void_p()
    ln = C.c_size_t()
    err = C.create_string_buffer(2048)
    rc = L.sdo_decompose_ex(_strs(read_names), _strs(read_seqs), len(read_names), _strs(mono_names),
                            _strs(mono_seqs), len(mono_names), threads, part, overlap, scoring(sc),
                            ed_thr, C.byref(out), C.byref(ln), err, 2048)
    if rc != 0:
        raise OracleError(rc, err.value.decode(errors="replace"))
    data = C.string_at(out, ln.value)
    L.sdo_free(out)
    return data


def nw_identity(query, target):
    """(edit_distance, matches, columns) of the unit-cost NW alignment edlib would report."""
    L = lib()
    m = C.c_int()
    c = C.c_int()
    q = query.encode() if isinstance(query, str) else query
    t = target.encode() if isinstance(target, str) else target
    ed = L.sdo_nw_identity(q, len(q), t, len(t), C.byref(m), C.byref(c))
    return ed, m.value, c.value


def hw_edit_distance(tmpl, text):
    L = lib()
    a = tmpl.encode() if isinstance(tmpl, str) else tmpl
    b = text.encode() if isinstance(text, str) else text
    return L.sdo_hw_edit_distance(a, len(a), b, len(b))


def reverse_complement(s):
    L = lib()
    b = s.encode()
    dst = C.create_string_buffer(len(b) + 1)
    if L.sdo_reverse_complement(b, len(b), dst) != 0:
        raise OracleError(-1, "unknown symbol in reverse_complement")
    return dst.value.decode()


# ---------------------------------------------------------------------------------------------
# the real reference binary (container only, or wherever oracle/_ref/dp travelled to)
# ---------------------------------------------------------------------------------------------
def have_ref_dp():
    return os.path.isfile(REF_DP) and os.access(REF_DP, os.X_OK)


def run_ref_dp(reads_fa, monomers_fa, threads=1, part=5000, overlap=500, sc=None, ed_thr=None):
    """Run oracle/_ref/dp.  sc=None -> 5-arg form (default scores); sc given -> 9-arg form
    (scores honoured, main.cpp:381); ed_thr given -> 10-arg form (scores IGNORED, main.cpp:389)."""
    cmd = [REF_DP, reads_fa, monomers_fa, str(threads), str(part), str(overlap)]
    if ed_thr is not None:
        s = sc or (-1, -1, -1, 1)
        cmd += [str(x) for x in s] + [str(ed_thr)]
    elif sc is not None:
        cmd += [str(x) for x in sc]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr
