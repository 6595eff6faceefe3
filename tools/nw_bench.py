#!/usr/bin/env python3
"""Throughput of the NW identity post-processing (developer tool): C2-like blocks x 24 templates
(--second-best shape), host threads vs the device kernel.  usage: nw_bench.py [blocks] [monomers]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mn, ms = synth.make_monomers(nm, seed=1)
tm = [m.decode() for m in ms] + [synth.revcomp_bytes(m).decode() for m in ms]
rn, rs = synth.make_reads(ms, max(1, nb * 171 // 50000 + 1), read_len=50000, seed=3)
seq = b"".join(rs)
starts = np.arange(nb, dtype=np.int64) * 171
ends = starts + 170
for homo in (False, True):
    for dev in (None, 0, 0):
        t0 = time.time()
        d, m, c = lib.identity_segments(seq, starts, ends, tm, homo, threads=64, device=dev)
        dt = time.time() - t0
        print("homo=%d %s: %d pairs in %.3f s = %.1f M pairs/s  checksum %d" % (
            homo, "host x64" if dev is None else "device", d.size, dt, d.size / dt / 1e6, int(d.sum() + 3 * m.sum())), flush=True)
