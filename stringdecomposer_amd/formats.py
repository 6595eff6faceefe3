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
