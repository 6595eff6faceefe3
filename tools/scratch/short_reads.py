#!/usr/bin/env python3
"""Developer probe (GPU box): many short reads (one tiny chunk each) -- rows against the oracle, time per read."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, synth
from oracle import binding as oracle
mn, ms = synth.make_monomers(12, seed=1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rn0, rs0 = synth.make_reads(ms, max(1, N * L // 50000 + 1), read_len=50000, seed=3)
big = b"".join(rs0)
reads = [big[i * L:(i + 1) * L] for i in range(N)]
rn = ["s%d" % i for i in range(N)]
t0 = time.perf_counter()
got = lib.decompose(rn, reads, mn, ms)
t1 = time.perf_counter()
got2 = lib.decompose(rn, reads, mn, ms)
t2 = time.perf_counter()
print("%d reads x %d bp: first call %.2f s, second %.2f s (%.1f Mbp/s), %d bytes of rows" % (N, L, t1 - t0, t2 - t1, N * L / 1e6 / (t2 - t1), len(got)), flush=True)
M = min(N, 20000)
exp = oracle.decompose(rn[:M], reads[:M], mn, ms, threads=32)
sub = lib.decompose(rn[:M], reads[:M], mn, ms)
print("rows of the first %d reads equal the oracle's: %s" % (M, sub == exp))
