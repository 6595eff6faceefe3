#!/usr/bin/env python3
"""Developer tool (needs a GPU): monomers of 0.6-2 kb (fast family since round 3: up to 32 virtual lanes per template,
traceback with 16 / 32 cells per lane) against the generic family forced.  usage: long_monomer_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, synth
st = synth.Stream(7, 7)
for nm, L in ((3, 700), (2, 1400), (2, 2000)):
    ms = [synth._to_ascii(st.below(L + 10 * j, 4)) for j in range(nm)]
    parts = []
    reads = []
    for r in range(64):
        p, tot = [], 0
        while tot < 50000:
            j = int(st.below(1, nm)[0])
            codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(ms[j], dtype=np.uint8))
            x = synth._to_ascii(synth.mutate(codes, st, 0.05, 0.02, 0.02))
            p.append(x); tot += len(x)
        reads.append(b"".join(p)[:50000])
    for kern, name in ((lib.KERNEL_AUTO, "auto"), (lib.KERNEL_GENERIC, "generic")):
        e = lib.Engine(ms, kernel=kern)
        e.load_reads(reads)
        e.run(); e.total_rows()
        t0 = time.perf_counter()
        for _ in range(3):
            e.run(); e.total_rows()
        dt = (time.perf_counter() - t0) / 3
        tm, info = e.timings(), e.info()
        e.close()
        print("%d monomers of ~%d bp (%d template cells): %-7s family %-7s %s  %.1f ms per 3.2 Mbp = %.0f Mbp/s (fill %.1f, traceback %.1f ms)" % (
            nm, L, info["sum_template_len"], name, info["family"], info["cells"], dt * 1e3, 3.2 / dt, tm["fill_ms"], tm["trace_ms"]), flush=True)
