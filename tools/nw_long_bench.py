#!/usr/bin/env python3
"""Identities of blocks of kilobase monomers: the device kernel (sd_nw_long.hip, pairs across lanes) against the host
threads.  usage: nw_long_bench.py [monomers] [monomer_len] [segments] [threads]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stringdecomposer_amd import lib, synth
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
th = int(sys.argv[4]) if len(sys.argv) > 4 else 16
st = synth.Stream(11, nm)
ms = [synth._to_ascii(st.below(L - 50 + int(st.below(1, 100)[0]), 4)) for _ in range(nm)]
tm = [m.decode() for m in ms] + [synth.revcomp_bytes(m).decode() for m in ms]
parts = []
while len(parts) < nseg:
    j = int(st.below(1, nm)[0])
    codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(ms[j], dtype=np.uint8))
    parts.append(synth._to_ascii(synth.mutate(codes, st, 0.05, 0.03, 0.03)))
seq = b"".join(parts)
bounds = np.cumsum([0] + [len(x) for x in parts])
starts, ends = bounds[:nseg].astype(np.int64), (bounds[1:nseg + 1] - 1).astype(np.int64)
Lb = lib.load()
tb = [t.encode() for t in tm]
tl = (C.c_int32 * len(tb))(*[len(t) for t in tb])
tarr = lib._strs(tb)
for homo in (0, 1):
    shape = (nseg, len(tm))
    d, m, c = (np.zeros(shape, dtype=np.int32) for _ in range(3))
    for rep in range(2):
        t0 = time.perf_counter()
        rc = Lb.sd_identity_segments_dev(seq, len(seq), starts.ctypes.data, ends.ctypes.data, nseg, tarr, tl, len(tb), None, homo, 0, th,
                                         d.ctypes.data, m.ctypes.data, c.ctypes.data)
        dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    hd, hm, hc = lib.identity_segments(seq, starts, ends, tm, bool(homo), threads=th)
    ht = time.perf_counter() - t0
    print("monomers %d x %d bp, %d blocks x %d templates, homo=%d: device rc=%d %.1f ms (%.2f M pairs/s), host %d threads %.1f ms, equal=%s" % (
        nm, L, nseg, len(tm), homo, rc, dt * 1e3, nseg * len(tm) / dt / 1e6, th, ht * 1e3, bool((d == hd).all() and (m == hm).all())))
