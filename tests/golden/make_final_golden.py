#!/usr/bin/env python3
"""Golden final / _alt TSVs from the UNMODIFIED reference command line (container only).

SURVEY.md App. C recipe 3: /root/reference is copied to a scratch directory (never into this repository),
the reference binary built by `make -C oracle ref` is placed where stringdecomposer/main.py:21 expects it,
and the two third-party imports the image lacks are provided by shim packages written HERE (not reference
code): `edlib` = ctypes over oracle/_ref/libedlib.so (the reference's vendored edlib, compiled where it
lies), `Bio` = the handful of SeqIO / SeqRecord / Seq calls main.py makes.  The reference's own
bin/stringdecomposer then runs unchanged; its outputs are committed as data:

  tests/golden/final/<case>/{params.json, final.tsv[, alt.tsv.gz]}   (+ sha256 of every output)

usage: python tests/golden/make_final_golden.py
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from stringdecomposer_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "final")
TD = os.path.join(HERE, "test_data")

EDLIB_SHIM = '''
import ctypes as C, os
_L = C.CDLL(os.environ["SD_EDLIB_SO"])
class _Cfg(C.Structure):
    _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int), ("eq", C.c_void_p), ("neq", C.c_int)]
class _Res(C.Structure):
    _fields_ = [("status", C.c_int), ("editDistance", C.c_int), ("endLocations", C.POINTER(C.c_int)),
                ("startLocations", C.POINTER(C.c_int)), ("numLocations", C.c_int),
                ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int), ("alphabetLength", C.c_int)]
_L.edlibAlign.restype = _Res
_L.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _Cfg]
_L.edlibAlignmentToCigar.restype = C.c_void_p
_L.edlibAlignmentToCigar.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int]
_L.edlibFreeAlignResult.argtypes = [_Res]
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_MODE = {"NW": 0, "SHW": 1, "HW": 2}
_TASK = {"distance": 0, "locations": 1, "path": 2}
def align(query, target, mode="NW", task="distance", k=-1):
    q, t = query.encode(), target.encode()
    r = _L.edlibAlign(q, len(q), t, len(t), _Cfg(k, _MODE[mode], _TASK[task], None, 0))
    out = {"editDistance": r.editDistance, "cigar": None}
    if task == "path" and r.alignment:
        p = _L.edlibAlignmentToCigar(r.alignment, r.alignmentLength, 1)   # EDLIB_CIGAR_EXTENDED
        out["cigar"] = C.string_at(p).decode()
        _libc.free(p)
    _L.edlibFreeAlignResult(r)
    return out
'''

BIO_INIT = ""
BIO_SEQ = '''
_C = str.maketrans("ACGTNacgtn", "TGCANtgcan")
class Seq(str):
    def reverse_complement(self):
        return Seq(str(self).translate(_C)[::-1])
    def upper(self):
        return Seq(str.upper(self))
    def __getitem__(self, k):
        r = str.__getitem__(self, k)
        return Seq(r) if isinstance(k, slice) else r
'''
BIO_SEQRECORD = '''
from Bio.Seq import Seq
class SeqRecord:
    def __init__(self, seq, id="<unknown id>", name="<unknown name>", description="<unknown description>"):
        self.seq = seq if isinstance(seq, Seq) else Seq(seq)
        self.id, self.name, self.description = id, name, description
    def upper(self):
        return SeqRecord(self.seq.upper(), self.id, self.name, self.description)
'''
BIO_SEQIO = '''
from Bio.Seq import Seq
from Bio.SeqRecord import SeqRecord
def parse(filename, fmt):
    assert fmt == "fasta"
    title, lines = None, []
    with open(filename) as f:
        for line in f:
            if line.startswith(">"):
                if title is not None:
                    yield _rec(title, lines)
                title, lines = line[1:].rstrip(), []
            elif title is not None:
                lines.append("".join(line.split()))
    if title is not None:
        yield _rec(title, lines)
def _rec(title, lines):
    first = title.split(None, 1)[0] if title.split() else ""
    return SeqRecord(Seq("".join(lines)), id=first, name=first, description=title)
def to_dict(records):
    d = {}
    for r in records:
        if r.id in d:
            raise ValueError("Duplicate key '%s'" % r.id)
        d[r.id] = r
    return d
'''


def sha(b):
    return hashlib.sha256(b).hexdigest()


def run_reference(work, reads_fa, mono_fa, extra):
    out = os.path.join(work, "out")
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, PYTHONPATH=os.path.join(work, "shims"), SD_EDLIB_SO=ob.REF_EDLIB)
    p = subprocess.run([sys.executable, os.path.join(work, "ref", "bin", "stringdecomposer"), reads_fa, mono_fa,
                        "-o", out, "-t", "4"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        raise SystemExit(p.stdout.decode()[-3000:])
    rd = lambda fn: open(os.path.join(out, fn), "rb").read()  # noqa: E731
    return rd("final_decomposition_raw.tsv"), rd("final_decomposition.tsv"), rd("final_decomposition_alt.tsv")


def emit(work, name, reads_fa, mono_fa, extra, inputs_rel, keep_alt):
    raw, final, alt = run_reference(work, reads_fa, mono_fa, extra)
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "final.tsv"), "wb") as f:
        f.write(final)
    if keep_alt and alt:
        with gzip.GzipFile(os.path.join(d, "alt.tsv.gz"), "wb", mtime=0) as f:
            f.write(alt)
    meta = {"args": extra, "inputs": inputs_rel, "raw_sha256": sha(raw), "final_sha256": sha(final),
            "alt_sha256": sha(alt), "alt_bytes": len(alt), "final_rows": final.count(b"\n")}
    with open(os.path.join(d, "params.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
        f.write("\n")
    print("%-28s final %s (%d rows)  alt %s (%d B)" % (name, meta["final_sha256"][:16], meta["final_rows"],
                                                      meta["alt_sha256"][:16], len(alt)))


def prepare():
    """Scratch copy of the reference with its binary in place and the two shim packages -> the scratch directory."""
    ob.build()
    if not (ob.have_ref_dp() and os.path.isfile(ob.REF_EDLIB)):
        raise SystemExit("needs oracle/_ref (make -C oracle ref), i.e. /root/reference")
    work = tempfile.mkdtemp(prefix="sd_ref_cli_")
    shutil.copytree("/root/reference", os.path.join(work, "ref"))
    os.makedirs(os.path.join(work, "ref", "stringdecomposer", "build", "bin"))
    shutil.copy(ob.REF_DP, os.path.join(work, "ref", "stringdecomposer", "build", "bin", "dp"))
    sh = os.path.join(work, "shims")
    os.makedirs(os.path.join(sh, "Bio"))
    os.makedirs(os.path.join(sh, "edlib"))
    for fn, txt in (("edlib/__init__.py", EDLIB_SHIM), ("Bio/__init__.py", BIO_INIT), ("Bio/Seq.py", BIO_SEQ),
                    ("Bio/SeqRecord.py", BIO_SEQRECORD), ("Bio/SeqIO.py", BIO_SEQIO)):
        with open(os.path.join(sh, fn), "w") as f:
            f.write(txt)
    return work


def main():
    work = prepare()
    rf, mf = os.path.join(TD, "read.fa"), os.path.join(TD, "DXZ1_star_monomers.fa")
    rel = ["test_data/read.fa", "test_data/DXZ1_star_monomers.fa"]
    emit(work, "td_second_best", rf, mf, ["--second-best"], rel, keep_alt=False)
    emit(work, "td_light", rf, mf, [], rel, keep_alt=False)
    emit(work, "td_second_best_i95", rf, mf, ["--second-best", "-i", "95"], rel, keep_alt=True)
    # 64-monomer set (BASELINE config 4 shape), one 12-kb read: all 128 templates per block
    d = os.path.join(OUT, "syn64_second_best")
    os.makedirs(d, exist_ok=True)
    mn, ms = synth.make_monomers(64, seed=11)
    rn, rs = synth.make_reads(ms, 2, read_len=12000, seed=15)
    synth.write_fasta(os.path.join(d, "reads.fa"), rn, rs, width=80)
    synth.write_fasta(os.path.join(d, "monomers.fa"), mn, ms)
    emit(work, "syn64_second_best", os.path.join(d, "reads.fa"), os.path.join(d, "monomers.fa"), ["--second-best"],
         ["final/syn64_second_best/reads.fa", "final/syn64_second_best/monomers.fa"], keep_alt=True)
    # blocks of 21 kb and 5 kb (N runs inside a monomer; -b 30000 keeps each read in one chunk): 20 * ceil(21171 / 64) *
    # 171 bytes of traceback data pass 1 MB, so edlib aligns the long block by Hirschberg's split (edlib.cpp:1186)
    d = os.path.join(OUT, "long_block")
    os.makedirs(d, exist_ok=True)
    st = synth.Stream(3, 14)
    ms = [synth._ACGT[st.below(171, 4)].tobytes(), synth._ACGT[st.below(168, 4)].tobytes()]
    rs = [ms[0] * 3 + ms[0][:100] + b"N" * 21000 + ms[0][100:] + ms[1] * 2,
          ms[1] * 4 + ms[0][:60] + b"N" * 5000 + ms[0][60:] + ms[0]]
    synth.write_fasta(os.path.join(d, "reads.fa"), ["long0", "mid1"], rs, width=80)
    synth.write_fasta(os.path.join(d, "monomers.fa"), ["L0", "L1"], ms)
    emit(work, "long_block", os.path.join(d, "reads.fa"), os.path.join(d, "monomers.fa"), ["--second-best", "-b", "30000"],
         ["final/long_block/reads.fa", "final/long_block/monomers.fa"], keep_alt=True)
    # thirty 342-bp monomers (pairs of synthetic ones): 60 templates of 20.5 k cells -- beyond one wave of the narrow
    # layout with templates longer than the widest lane: the tiled multi-wave layout of csrc/sd_fast_wt.hip
    d = os.path.join(OUT, "tiled_second_best")
    os.makedirs(d, exist_ok=True)
    mn, ms = synth.make_monomers(60, seed=21)
    mn, ms = ["T%d" % j for j in range(30)], [ms[2 * j] + ms[2 * j + 1] for j in range(30)]
    rn, rs = synth.make_reads(ms, 2, read_len=6000, seed=23)
    synth.write_fasta(os.path.join(d, "reads.fa"), rn, rs, width=80)
    synth.write_fasta(os.path.join(d, "monomers.fa"), mn, ms)
    emit(work, "tiled_second_best", os.path.join(d, "reads.fa"), os.path.join(d, "monomers.fa"), ["--second-best"],
         ["final/tiled_second_best/reads.fa", "final/tiled_second_best/monomers.fa"], keep_alt=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
