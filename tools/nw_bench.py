#!/usr/bin/env python3
"""Thread scaling of sd_nw_identity_batch (host post-processing) on monomer-sized pairs (developer tool)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import lib, synth
mn, ms = synth.make_monomers(64, seed=1)
rn, rs = synth.make_reads(ms, 1, read_len=50000, seed=1)
segs = [rs[0][i * 171:(i + 1) * 171 + 3] for i in range(280)]
q = [s for s in segs for _ in range(64)] * 4
t = [m for _ in segs for m in ms] * 4
L = lib.load()
n = len(q)
qa, ta = lib._strs(q), lib._strs(t)
ql = (C.c_int32 * n)(*[len(x) for x in q])
tl = (C.c_int32 * n)(*[len(x) for x in t])
d, m, c = (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * n)()
for th in (1, 2, 4, 8, 16, 32, 64):
    t0 = time.perf_counter()
    L.sd_nw_identity_batch(qa, ql, ta, tl, n, th, d, m, c)
    dt = time.perf_counter() - t0
    print("%2d threads: %.2f us per alignment, %.2f M alignments/s" % (th, dt / n * 1e6, n / dt / 1e6))
