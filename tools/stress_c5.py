#!/usr/bin/env python3
"""BASELINE config 5 shape on one GPU: a single 200-Mb sequence x 12 monomers with custom scoring through
(a) the one-shot call and (b) three chunk ranges + host assembly (what three GPUs would do); outputs must
be byte-identical.  usage: stress_c5.py [Mb]   (developer tool)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, shard, synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mn, ms = synth.make_monomers(12, seed=1)
t0 = time.time()
rn, rs = synth.make_reads(ms, 1, read_len=2_000_000, seed=7)
seq = (rs[0] * (mb // 2 + 1))[: mb * 1_000_000]
print("gen %.1fs, %d bp" % (time.time() - t0, len(seq)), flush=True)
sc = (-2, -3, -4, 2)
t0 = time.time()
one = lib.decompose(["chr"], [seq], mn, ms, scoring=sc, threads=64)
dt = time.time() - t0
print("one-shot: %.2fs  %.1f Mbp/s  rows=%d sha=%s" % (dt, mb / dt, one.count(b"\n"), hashlib.sha256(one).hexdigest()[:16]), flush=True)
n = lib.chunk_table_size([len(seq)])
t0 = time.time()
parts = [lib.decompose_chunk_range([seq], ms, *shard.block_range(n, g, 3), scoring=sc, threads=64) for g in range(3)]
t1 = time.time()
recs = np.concatenate([p[0] for p in parts])
off = np.concatenate([[0]] + [p[1][1:] + sum(len(q[0]) for q in parts[:i]) for i, p in enumerate(parts)])
got = lib.assemble_tsv(["chr"], [len(seq)], mn, recs, off, scoring=sc, threads=64)
print("3 ranges: device %.2fs + assembly %.2fs, %d chunks, sha=%s" % (t1 - t0, time.time() - t1, n, hashlib.sha256(got).hexdigest()[:16]), flush=True)
assert got == one
print("identical")
