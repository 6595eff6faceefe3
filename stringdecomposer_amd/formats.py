"""On-disk formats of StringDecomposer as a small library API.

Three tab-separated text files make up the contract with downstream tools (centroFlye, HORmon):

  <out>_raw.tsv   7 columns, written by the DP stage (reference main.cpp:272-285, SaveBatch):
                  read, monomer, start, end, score ("%f" of a float), gap to the previous row of the
                  same read (start - previous end; the first row of a read: its start), end - start
  <out>.tsv       12 columns, README.md:77-83 of the reference / main.py:153-160:
                  read, best monomer, start, end, identity, second best monomer, its identity,
                  homopolymer-compressed best monomer, its identity, homo second best, its identity,
                  reliability ('+' or '?')
  <out>_alt.tsv   6 columns (main.py:161-165): read, monomer, start, end, identity, '*' for the
                  monomer reported in <out>.tsv else '-'

A fourth, binary, file is this package's own (the reference has no counterpart; opt-in, `--records`):

  <out>_raw.sdr   the rows of <out>_raw.tsv as 16-byte records per read (read_records / write_records below;
                  layout in csrc/sd_records.hpp, C-ABI sd_write_records / sd_read_records in include/sd_hip.h).
                  records_to_raw_tsv(read_records(path)) is the raw TSV byte for byte.

Readers return lists of named tuples with numeric fields converted; writers reproduce the reference's
text exactly (round trip = identity on files the reference or this package wrote), so a tool can
filter / merge decompositions without re-deriving the formatting rules.
"""
from collections import namedtuple

RawRow = namedtuple("RawRow", "read monomer start end score gap length")
FinalRow = namedtuple("FinalRow", "read monomer start end identity second_best second_best_identity "
                                   "homo_best homo_best_identity homo_second_best "
                                   "homo_second_best_identity reliability")
AltRow = namedtuple("AltRow", "read monomer start end identity best")


class FormatError(ValueError):
    def __init__(self, path, lineno, msg):
        super().__init__("%s:%d: %s" % (path, lineno, msg))
        self.path, self.lineno = path, lineno


def _lines(path_or_text, is_text):
    if is_text:
        data = path_or_text
    else:
        with open(path_or_text, "r", newline="") as f:
            data = f.read()
    if not data:
        return []
    body = data[:-1] if data.endswith("\n") else data
    return body.split("\n")


def _parse(path_or_text, is_text, ncol, conv, cls):
    name = "<text>" if is_text else str(path_or_text)
    out = []
    for no, ln in enumerate(_lines(path_or_text, is_text), 1):
        f = ln.split("\t")
        if len(f) != ncol:
            raise FormatError(name, no, "expected %d tab-separated columns, found %d" % (ncol, len(f)))
        try:
            out.append(cls(*[c(x) for c, x in zip(conv, f)]))
        except ValueError as e:
            raise FormatError(name, no, str(e))
    return out


def _ident(x):
    return x


# ---- raw ------------------------------------------------------------------------------------------
def read_raw(path):
    return _parse(path, False, 7, (_ident, _ident, int, int, float, int, int), RawRow)


def parse_raw(text):
    return _parse(text, True, 7, (_ident, _ident, int, int, float, int, int), RawRow)


def format_raw(rows):
    """SaveBatch (main.cpp:272-285): '%f' of the score; gap / length columns are stored, not recomputed."""
    return "".join("%s\t%s\t%d\t%d\t%f\t%d\t%d\n" % (r.read, r.monomer, r.start, r.end, r.score, r.gap,
                                                    r.length) for r in rows)


def raw_rows(read, triples):
    """Rows of one read from (monomer, start, end, score) in position order, with the derived
    columns filled in as SaveBatch does (prev_end starts at 0 for every read)."""
    out, prev_end = [], 0
    for monomer, start, end, score in triples:
        out.append(RawRow(read, monomer, int(start), int(end), float(score), int(start) - prev_end,
                          int(end) - int(start)))
        prev_end = int(end)
    return out


# ---- final ----------------------------------------------------------------------------------------
_FINAL_CONV = (_ident, _ident, int, int, float, _ident, float, _ident, float, _ident, float, _ident)


def read_final(path):
    return _parse(path, False, 12, _FINAL_CONV, FinalRow)


def parse_final(text):
    return _parse(text, True, 12, _FINAL_CONV, FinalRow)


def format_final(rows):
    """main.py:153-160: identities with '{:.2f}'."""
    f2 = "{:.2f}".format
    return "".join("\t".join([r.read, r.monomer, str(r.start), str(r.end), f2(r.identity), r.second_best,
                              f2(r.second_best_identity), r.homo_best, f2(r.homo_best_identity),
                              r.homo_second_best, f2(r.homo_second_best_identity), r.reliability]) + "\n"
                   for r in rows)


# ---- alt ------------------------------------------------------------------------------------------
def _star(x):
    if x not in ("*", "-"):
        raise ValueError("last column must be '*' or '-', found %r" % x)
    return x == "*"


def read_alt(path):
    return _parse(path, False, 6, (_ident, _ident, int, int, float, _star), AltRow)


def parse_alt(text):
    return _parse(text, True, 6, (_ident, _ident, int, int, float, _star), AltRow)


def format_alt(rows):
    f2 = "{:.2f}".format
    return "".join("\t".join([r.read, r.monomer, str(r.start), str(r.end), f2(r.identity),
                              "*" if r.best else "-"]) + "\n" for r in rows)


def by_read(rows):
    """Group consecutive rows by read name, preserving file order: [(read, [rows])]."""
    out = []
    for r in rows:
        if not out or out[-1][0] != r.read:
            out.append((r.read, []))
        out[-1][1].append(r)
    return out


# ---- binary record stream (<out>_raw.sdr) -------------------------------------------------------------
# Pure Python (struct only): a downstream tool needs neither the HIP library nor numpy to consume it.  The native
# reader / writer of the same bytes is sd_read_records / sd_write_records (lib.read_records / lib.write_records).
import struct

RECORDS_MAGIC = b"SDRECS1\n"
Records = namedtuple("Records", "scoring part_size overlap ed_thr templates reads")
"""scoring = (ins, del, mismatch, match); templates = names as in column 2 of the raw TSV (monomers, then monomers
+ "'"); reads = [(name, read_len or -1, [(template index, start, end, score), ...])] in file order."""


def _pad8(n):
    return (-n) & 7


def write_records(path, rec):
    """Records -> file.  Layout: csrc/sd_records.hpp."""
    head = bytearray(RECORDS_MAGIC)
    head += struct.pack("<II", 0, 0)
    head += struct.pack("<7i", *(list(rec.scoring) + [rec.part_size, rec.overlap, rec.ed_thr]))
    head += struct.pack("<I", len(rec.templates))
    for t in rec.templates:
        b = t.encode()
        head += struct.pack("<I", len(b)) + b
    head += b"\0" * _pad8(len(head))
    struct.pack_into("<I", head, 8, len(head))
    nt = len(rec.templates)
    total = 0
    with open(path, "wb") as f:
        f.write(head)
        for name, rlen, rows in rec.reads:
            b = name.encode()
            f.write(struct.pack("<IIqq", len(b), 0, -1 if rlen is None else int(rlen), len(rows)))
            f.write(b + b"\0" * _pad8(len(b)))
            for t, s, e, sc in rows:
                if not 0 <= t < nt:
                    raise ValueError("template index %d outside the template table" % t)
                f.write(struct.pack("<4i", t, s, e, int(sc)))
            total += len(rows)
        f.write(struct.pack("<IIqq", 0xFFFFFFFF, 0, len(rec.reads), total))


def read_records(path):
    """file -> Records; FormatError for anything that is not a complete, consistent record stream."""
    with open(path, "rb") as f:
        d = f.read()

    def bad(msg):
        return FormatError(str(path), 0, msg)

    if len(d) < 48 or d[:8] != RECORDS_MAGIC:
        raise bad("not a record stream (bad magic)")
    hb, _ = struct.unpack_from("<II", d, 8)
    v = struct.unpack_from("<7i", d, 16)
    (nt,) = struct.unpack_from("<I", d, 44)
    at = 48
    templates = []
    for _ in range(nt):
        if at + 4 > len(d):
            raise bad("truncated template table")
        (l,) = struct.unpack_from("<I", d, at)
        at += 4
        if at + l > len(d):
            raise bad("truncated template table")
        templates.append(d[at:at + l].decode())
        at += l
    at += _pad8(at)
    if at != hb or hb > len(d):
        raise bad("header size does not match the template table")
    reads, total = [], 0
    while True:
        if at + 8 > len(d):
            raise bad("truncated: no trailer (the writer did not finish)")
        nl, _ = struct.unpack_from("<II", d, at)
        at += 8
        if nl == 0xFFFFFFFF:
            if at + 16 > len(d):
                raise bad("truncated trailer")
            nr, nrow = struct.unpack_from("<qq", d, at)
            at += 16
            if nr != len(reads) or nrow != total:
                raise bad("trailer totals do not match the read blocks")
            if at != len(d):
                raise bad("bytes after the trailer")
            return Records(tuple(v[:4]), v[4], v[5], v[6], templates, reads)
        if at + 16 > len(d):
            raise bad("truncated read block")
        rlen, n = struct.unpack_from("<qq", d, at)
        at += 16
        if n < 0 or at + nl > len(d):
            raise bad("truncated read block")
        name = d[at:at + nl].decode()
        at += nl + _pad8(nl)
        if at + 16 * n > len(d):
            raise bad("truncated read block")
        flat = struct.unpack_from("<%di" % (4 * n), d, at)
        at += 16 * n
        rows = [tuple(flat[4 * i:4 * i + 4]) for i in range(n)]
        for r in rows:
            if not 0 <= r[0] < nt:
                raise bad("record with a template index outside the template table")
        reads.append((name, rlen, rows))
        total += n


def records_to_raw_rows(rec):
    """Records -> [RawRow] exactly as SaveBatch derives them (gap from the previous end of the same read, length)."""
    out = []
    for name, _, rows in rec.reads:
        out.extend(raw_rows(name, [(rec.templates[t], s, e, sc) for t, s, e, sc in rows]))
    return out


def records_to_raw_tsv(rec):
    """Records -> the text of <out>_raw.tsv (main.cpp:272-285)."""
    return format_raw(records_to_raw_rows(rec))


def raw_to_records(rows, templates, scoring=(-1, -1, -1, 1), part_size=5000, overlap=500, ed_thr=-1, read_lens=None):
    """[RawRow] (+ the template names in the DP's order: monomers, then monomers + "'") -> Records.  A repeated template
    name maps to its first index, which is what the raw TSV can tell."""
    idx = {}
    for i, t in enumerate(templates):
        idx.setdefault(t, i)
    reads = []
    for name, rr in by_read(rows):
        reads.append((name, -1 if read_lens is None else read_lens.get(name, -1),
                      [(idx[r.monomer], r.start, r.end, int(r.score)) for r in rr]))
    return Records(tuple(scoring), part_size, overlap, ed_thr, list(templates), reads)
