"""Test infrastructure: identity of a (query, target) pair as stringdecomposer/main.py:29-60 gets it from
python-edlib -- through the reference's vendored edlib (oracle/_ref/libedlib.so, compiled from
/root/reference by `make -C oracle ref`; it travels to the GPU box) when present, else through the
oracle's own full-matrix NW (oracle/sd_oracle.c: sdo_nw_identity, pinned against edlib on CPU)."""
import ctypes
import os

from oracle import binding as oracle


class _Cfg(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int), ("mode", ctypes.c_int), ("task", ctypes.c_int),
                ("eq", ctypes.c_void_p), ("neq", ctypes.c_int)]


class _Res(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("editDistance", ctypes.c_int),
                ("endLocations", ctypes.POINTER(ctypes.c_int)),
                ("startLocations", ctypes.POINTER(ctypes.c_int)),
                ("numLocations", ctypes.c_int),
                ("alignment", ctypes.POINTER(ctypes.c_ubyte)),
                ("alignmentLength", ctypes.c_int), ("alphabetLength", ctypes.c_int)]


_ed = None


def have_edlib():
    return os.path.isfile(oracle.REF_EDLIB)


def _edlib():
    global _ed
    if _ed is None:
        ed = ctypes.CDLL(oracle.REF_EDLIB)
        ed.edlibAlign.restype = _Res
        ed.edlibAlign.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, _Cfg]
        ed.edlibFreeAlignResult.argtypes = [_Res]
        _ed = ed
    return _ed


def nw(q, t):
    """(edit distance, '=' columns, alignment columns) of edlib.align(q, t, mode='NW', task='path');
    (-1, 0, 0) for an empty side (main.py:30-33)."""
    q = q.encode() if isinstance(q, str) else q
    t = t.encode() if isinstance(t, str) else t
    if not q or not t:
        return (-1, 0, 0)
    if not have_edlib():
        return oracle.nw_identity(q, t)
    ed = _edlib()
    r = ed.edlibAlign(q, len(q), t, len(t), _Cfg(-1, 0, 2, None, 0))
    m = bytes(r.alignment[:r.alignmentLength]).count(b"\x00")
    out = (r.editDistance, m, r.alignmentLength)
    ed.edlibFreeAlignResult(r)
    return out


def identity(q, t):
    """main.py:38-60 (aai): percent identity, 0 for an empty side."""
    ed, m, c = nw(q, t)
    if ed == -1:
        return 0
    a = 0.0
    a += m
    a /= c
    return a * 100


def homo(s):
    """main.py:87-92 convert_to_homo."""
    out = []
    prev = None
    for ch in s:
        if ch != prev:
            out.append(ch)
            prev = ch
    return "".join(out)
