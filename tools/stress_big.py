#!/usr/bin/env python3
"""One-off stress: a few hundred Mbp through sd_decompose with automatic HBM-sized batching vs a
forced small batch size; outputs must be byte-identical.  usage: stress_big.py [reads]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
mn, ms = synth.make_monomers(12, seed=1)
t0 = time.time(); rn, rs = synth.make_reads(ms, n, read_len=50000, seed=3); print("gen %.1fs" % (time.time() - t0), flush=True)
for cap in (0, 40_000_000):
    t0 = time.time()
    out = lib.decompose(rn, rs, mn, ms, threads=64, max_batch_rows=cap)
    dt = time.time() - t0
    print("cap=%d: %.2fs  %.1f Mbp/s end-to-end  rows=%d sha=%s" % (cap, dt, n * 0.05 / dt, out.count(b"\n"),
          hashlib.sha256(out).hexdigest()[:16]), flush=True)
