#!/usr/bin/env python3
"""Randomised campaign for the batch pipeline (developer tool, needs a GPU): random streams of jobs through lib.Stream.imap
(random sub-batch counts, row caps per batch, numbers of jobs outstanding, reads from 1 bp to several chunks), and the same
read sets through the file / chunk-range entry points (fresh pipelines and pipelines from the cache, one batch and many) --
every result against the CPU oracle.  usage: python tools/fuzz_stream.py [cases] [seed] [log file to append the summary to]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, shard, synth
from oracle import binding as oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st = synth.Stream(seed, 7)


def rnd(k):
    return int(st.below(1, k)[0])


bad = 0
t0 = time.time()
d = tempfile.mkdtemp()
for case in range(cases):
    nm = [3, 12, 12, 30, 64][rnd(5)]
    mn, ms = synth.make_monomers(nm, seed=seed * 1000 + case)
    if rnd(3) == 0:
        ms = [m[:40 + rnd(100)] for m in ms]
    tn = list(mn) + [n + "'" for n in mn]
    sc = [(-1, -1, -1, 1), (-2, -3, -4, 2), (-1, -2, -1, 1)][rnd(3)]
    part, ov = [(5000, 500), (700, 100), (333, 77)][rnd(3)]
    kw = dict(scoring=sc, part_size=part, overlap=ov, threads=1 + rnd(8))
    jobs = []
    for j in range(2 + rnd(6)):
        lens = [[1, 7, part, part + 1, part + ov, 2 * part + 3][rnd(6)] if rnd(4) == 0 else 200 + rnd(6 * part) for _ in range(1 + rnd(7))]
        if rnd(5) == 0:
            lens.append(20 * part + rnd(part))
        rn, rs = synth.make_reads(ms, len(lens), read_len=max(lens), seed=seed * 77 + case * 13 + j)
        rs = [s[:L] for s, L in zip(rs, lens)]
        jobs.append((["j%d_%d" % (j, i) for i in range(len(rs))], rs))
    want = [oracle.decompose(names, seqs, mn, ms, threads=8, sc=sc, part=part, overlap=ov) for names, seqs in jobs]
    sub, cap, depth = 1 + rnd(5), [0, 0, 3 * (part + ov), 9 * (part + ov), 40 * (part + ov)][rnd(5)], rnd(5)
    s = lib.Stream(ms, sub_batches=sub, max_batch_rows=cap, **kw)
    got = list(s.imap([seqs for _, seqs in jobs], as_lists=True, depth=depth))
    s.close()
    ok = True
    for (names, seqs), rows, w in zip(jobs, got, want):
        txt = b"".join(lib.format_rows(n, tn, r) for n, r in zip(names, rows))
        ok = ok and txt == w
    # the same read sets through the one-shot, file and chunk-range forms (pipelines from the cache after the first)
    for (names, seqs), w in zip(jobs, want):
        ok = ok and lib.decompose(names, seqs, mn, ms, max_batch_rows=cap, **kw) == w
        rfa, mfa, out = os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), os.path.join(d, "o.tsv")
        synth.write_fasta(rfa, names, seqs, width=[0, 60][rnd(2)])
        synth.write_fasta(mfa, mn, ms)
        lib.decompose_files(rfa, mfa, out, max_batch_rows=cap, **kw)
        with open(out, "rb") as f:
            ok = ok and f.read() == w
        n = lib.chunk_table_size([len(x) for x in seqs], part, ov)
        g = 1 + rnd(3)
        parts = [lib.decompose_chunk_range(seqs, ms, *shard.block_range(n, q, g), max_batch_rows=cap, **kw) for q in range(g)]
        recs = np.concatenate([p[0] for p in parts])
        off = np.concatenate([[0]] + [p[1][1:] + sum(len(q[0]) for q in parts[:i]) for i, p in enumerate(parts)])
        keep = {k: v for k, v in kw.items() if k in ("scoring", "part_size", "overlap", "threads")}
        ok = ok and lib.assemble_tsv(names, [len(x) for x in seqs], mn, recs, off, **keep) == w
    if rnd(4) == 0:
        lib.release_cache()
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(nm=nm, sc=sc, part=part, ov=ov, sub=sub, cap=cap, depth=depth), flush=True)
line = "fuzz_stream: seed %d, %d cases, %d mismatches, %.1fs" % (seed, cases, bad, time.time() - t0)
print(line)
if len(sys.argv) > 3:
    with open(sys.argv[3], "a") as f:
        f.write(time.strftime("%Y-%m-%d %H:%M:%S  ") + line + "\n")
sys.exit(1 if bad else 0)
