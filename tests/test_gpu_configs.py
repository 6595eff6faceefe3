"""BASELINE.json configs 3, 4 and 5 at (or at one GPU's share of) their real size, on a real MI355X.

  C4  64-monomer set (128 templates) x 50-kb reads with --second-best: the final and _alt TSVs of the
      CLI, every identity column of a sample of >= 200 blocks x all 128 templates against the
      reference's vendored edlib (second-best order m0, m0', m1, ... of main.py:79-84,122-146).
  C5  one 200-Mb sequence x 12 monomers, custom -s: one-shot == 3 chunk ranges + host assembly, and
      >= 64 chunks (both sides of every range seam included) record-for-record against the oracle.
  C3  one GPU's share of 100 000 x 50 kb (12 500 reads, 625 Mbp) through sd_decompose in many device
      batches: sampled reads byte-for-byte against the oracle, structural invariants on every row.
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

import edlib_ref
from stringdecomposer_amd import lib, shard, synth

pytestmark = pytest.mark.gpu

LR = [-31.48494996, 0.41784018, 0.69186882]   # main.py:25-26
F2 = "{:.2f}".format


def _cli(args, out):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "stringdecomposer")] + args + ["-o", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    rd = lambda fn: open(os.path.join(out, fn)).read()  # noqa: E731
    return rd("final_decomposition_raw.tsv"), rd("final_decomposition.tsv"), rd("final_decomposition_alt.tsv")


def test_c4_64_monomers_second_best_final_and_alt_tsv(tmp_path, oracle):
    mn, ms = synth.make_monomers(64, seed=11)
    rn, rs = synth.make_reads(ms, 4, read_len=50000, seed=14)
    rf, mf = str(tmp_path / "r.fa"), str(tmp_path / "m.fa")
    synth.write_fasta(rf, rn, rs, width=80)
    synth.write_fasta(mf, mn, ms)
    raw, final, alt = _cli([rf, mf, "-t", "32", "--second-best"], str(tmp_path / "sb"))
    raw_rows = [ln.split("\t") for ln in raw.splitlines()]
    fin_rows = [ln.split("\t") for ln in final.splitlines()]
    alt_rows = [ln.split("\t") for ln in alt.splitlines()]
    assert len(raw_rows) > 1000 and len(fin_rows) == len(raw_rows)       # -i 0: every block is printed
    # monomers in the post-processing order of main.py:79-84: m0, m0', m1, m1', ...
    names, seqs = [], []
    for n, s in zip(mn, ms):
        names += [n, n + "'"]
        seqs += [s.decode(), synth.revcomp_bytes(s).decode()]
    T = len(names)
    assert T == 128 and len(alt_rows) == len(fin_rows) * T
    # raw rows themselves: the DP of the first read against the oracle
    exp_raw = oracle.decompose(rn[:1], rs[:1], mn, ms, threads=min(32, os.cpu_count() or 1)).decode()
    assert raw.startswith(exp_raw)
    for a, b in zip(raw_rows, fin_rows):
        assert a[:4] == b[:4]
    reads = {n: s.decode() for n, s in zip(rn, rs)}
    hseqs = [edlib_ref.homo(s) for s in seqs]
    pick = sorted(set(int(x) for x in synth.Stream(4, 4).below(230, len(fin_rows))) | {0, len(fin_rows) - 1})
    assert len(pick) >= 200

    def check(bi):
        row = fin_rows[bi]
        seg = reads[row[0]][int(row[2]):int(row[3]) + 1]
        scores = {nm: edlib_ref.identity(seg, sq) for nm, sq in zip(names, seqs)}
        sb, sbs = None, -1
        for m in scores:                              # main.py:124-128: first maximum among the others
            if m != row[1] and (not sb or sbs < scores[m]):
                sb, sbs = m, scores[m]
        hseg = edlib_ref.homo(seg)
        hs = sorted([[nm, edlib_ref.identity(hseg, hq)] for nm, hq in zip(names, hseqs)], key=lambda x: -x[1])
        q = "+" if LR[0] + LR[1] * scores[row[1]] + LR[2] * (scores[row[1]] - sbs) > 0 else "?"
        exp = [row[0], row[1], row[2], row[3], F2(scores[row[1]]), str(sb), F2(sbs), hs[0][0], F2(hs[0][1]),
               hs[1][0], F2(hs[1][1]), q]
        assert row == exp, (bi, row, exp)
        for x, nm in enumerate(names):
            assert alt_rows[bi * T + x] == [row[0], nm, row[2], row[3], F2(scores[nm]), "*" if nm == row[1] else "-"]
        return 1

    with ThreadPoolExecutor(16) as ex:               # ctypes releases the GIL inside edlib
        assert sum(ex.map(check, pick)) == len(pick)
    # light mode (main.py:112-121,169): same blocks and scores, no second best, an empty _alt.tsv
    raw2, final2, alt2 = _cli([rf, mf, "-t", "32"], str(tmp_path / "light"))
    assert raw2 == raw and alt2 == ""
    for a, b in zip(final2.splitlines(), fin_rows):
        f = a.split("\t")
        assert f[:5] == b[:5] and f[5:11] == ["None", "-1.00", "None", "-1.00", "None", "-1.00"]


def test_c4_full_size_raw_tsv_equals_the_reference_binary():
    """BASELINE config 4 at its full size -- 64 monomers (128 templates) x 256 reads x 50 kb, 305 G cells -- every row of the
    raw TSV against the REAL reference: sha256 of what oracle/_ref/dp printed for this input in the build container
    (tests/golden/make_fullsize_hashes.py; ~25 minutes of the reference's CPU path, not repeated here)."""
    with open(os.path.join(GOLDEN, "fullsize_sha256.json")) as f:
        gold = json.load(f).get("c4")
    assert gold, "tests/golden/fullsize_sha256.json has no c4 entry: run tests/golden/make_fullsize_hashes.py <threads> c4"
    mn, ms = synth.make_monomers(64, seed=11)
    rn, rs = synth.make_reads(ms, 256, read_len=50000, seed=14)
    got = lib.decompose(rn, rs, mn, ms, threads=min(64, os.cpu_count() or 1))
    assert got.count(b"\n") == gold["rows"] and len(got) == gold["bytes"]
    assert hashlib.sha256(got).hexdigest() == gold["sha256"]


def test_c5_200mb_single_sequence_chunk_ranges_and_oracle(oracle):
    mb = 200
    mn, ms = synth.make_monomers(12, seed=1)
    # 100 independent 2-Mb reads end to end: 40 000 DISTINCT chunks (rounds 4-5 tiled one 2-Mb read; 2 000 000 is a
    # multiple of the part size, so chunk c and chunk c + 400 were the same bytes)
    rn, rs = synth.make_reads(ms, mb // 2, read_len=2_000_000, seed=7)
    seq = b"".join(rs)
    assert len(seq) == mb * 1_000_000
    assert len({seq[c * 5000: c * 5000 + 5500] for c in range(0, 40000, 37)}) == len(range(0, 40000, 37))
    sc = (-2, -3, -4, 2)
    th = min(64, os.cpu_count() or 1)
    one = lib.decompose(["chr"], [seq], mn, ms, scoring=sc, threads=th)
    n = lib.chunk_table_size([len(seq)])
    assert n == 40000
    ranges = [shard.block_range(n, g, 3) for g in range(3)]
    parts = [lib.decompose_chunk_range([seq], ms, lo, hi, scoring=sc, threads=th) for lo, hi in ranges]
    recs = np.concatenate([p[0] for p in parts])
    off = np.concatenate([[0]] + [p[1][1:] + sum(len(q[0]) for q in parts[:i]) for i, p in enumerate(parts)])
    got = lib.assemble_tsv(["chr"], [len(seq)], mn, recs, off, scoring=sc, threads=th)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(one).hexdigest()
    assert one.count(b"\n") > 1_100_000
    # every row of the 200-Mb job against the REAL reference: sha256 of what oracle/_ref/dp printed for this sequence in the
    # build container (tests/golden/make_fullsize_hashes.py; 4-8 minutes of the reference's CPU path, not repeated here)
    with open(os.path.join(GOLDEN, "fullsize_sha256.json")) as f:
        gold = json.load(f)["c5"]
    assert one.count(b"\n") == gold["rows"] and len(one) == gold["bytes"]
    assert hashlib.sha256(one).hexdigest() == gold["sha256"]
    # the multi-GPU form: eight ranks, each makes the text of its own 25-Mb share from its records and the exchanged
    # edges (csrc/sd_seam.hpp); their texts in rank order are the file
    asm = []
    for g in range(8):
        lo, hi = shard.block_range(n, g, 8)
        asm.append(lib.RangeAssembler.from_lists(["chr"], [len(seq)], mn, lo, hi, recs[off[lo]:off[hi]], off[lo:hi + 1] - off[lo],
                                                 scoring=sc, threads=2))
    edges = [a.edge for a in asm]
    texts = []
    for g, a in enumerate(asm):
        a.text(edges, g)
        texts.append(a.bytes())
        st = a.stats()
        assert not st["formatted_again"] and st["rows_printed_after_exchange"] < 200, st
        a.close()
    assert hashlib.sha256(b"".join(texts)).hexdigest() == hashlib.sha256(one).hexdigest()
    # sampled chunks against AlignPartClassicDP (oracle), both sides of every range seam included
    pick = {0, n - 1}
    for lo, hi in ranges:
        pick |= {lo, hi - 1}
    pick |= set(int(x) for x in synth.Stream(5, 5).below(64, n))
    pick = sorted(pick)
    assert len(pick) >= 64
    tm = [m.decode() for m in ms] + [synth.revcomp_bytes(m).decode() for m in ms]

    def check(c):
        sub = seq[c * 5000: c * 5000 + 5500]
        exp = oracle.align_chunk(sub, tm, sc)
        g = recs[off[c]:off[c + 1]]
        assert [(int(r["tmpl"]), int(r["start"]), int(r["end"]), int(r["score"])) for r in g] == \
            [(t, a, b, int(v)) for (t, a, b, v) in exp], c
        return 1

    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        assert sum(ex.map(check, pick)) == len(pick)


def test_c3_one_gpu_share_12500_reads_multi_batch(oracle):
    """12 500 reads x 50 kb = 625 Mbp = one GPU's eighth of BASELINE config 3.  2 500 distinct reads, each
    appearing 5 times under different names (so the same read lands in different device batches: its
    rows must not depend on the batch)."""
    mn, ms = synth.make_monomers(12, seed=1)
    base_n, rs0 = synth.make_reads(ms, 2500, read_len=50000, seed=33)
    rs = rs0 * 5
    rn = ["q%d" % i for i in range(len(rs))]
    th = min(64, os.cpu_count() or 1)
    out = lib.decompose(rn, rs, mn, ms, threads=th, max_batch_rows=40_000_000)   # >= 17 device batches
    per_read = {}
    cur, buf = None, []
    for ln in out.split(b"\n")[:-1]:
        nm, rest = ln.split(b"\t", 1)
        if nm != cur:
            if cur is not None:
                per_read[cur] = buf
            cur, buf = nm, []
        buf.append(rest)
    per_read[cur] = buf
    assert len(per_read) == 12500
    for i in range(2500):
        a = per_read[b"q%d" % i]
        assert len(a) > 250
        for k in range(1, 5):
            assert per_read[b"q%d" % (i + 2500 * k)] == a, i
    # structural invariants on every row: rows ordered, inside the read, template range
    names = set(n.encode() for n in mn) | set(n.encode() + b"'" for n in mn)
    for i in range(0, 2500, 7):
        prev_end = -1
        for rest in per_read[b"q%d" % i]:
            f = rest.split(b"\t")
            s, e = int(f[1]), int(f[2])
            assert f[0] in names and 0 <= s <= e < 50000
            assert int(f[4]) == s - max(prev_end, 0) and int(f[5]) == e - s
            prev_end = e
    # sampled reads byte-for-byte against the oracle
    pick = sorted(set(int(x) for x in synth.Stream(6, 6).below(40, 12500)) | {0, 12499})
    exp = oracle.decompose([rn[i] for i in pick], [rs[i] for i in pick], mn, ms, threads=min(32, os.cpu_count() or 1))
    got = b"".join(b"".join(rn[i].encode() + b"\t" + r + b"\n" for r in per_read[rn[i].encode()]) for i in pick)
    assert got == exp
