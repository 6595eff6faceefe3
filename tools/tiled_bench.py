#!/usr/bin/env python3
"""Developer tool (needs a GPU): template sets of the tiled multi-wave layout (csrc/sd_fast_wt.hip: templates longer than
the widest lane in sets beyond one wave) against the generic family forced, same rows.
usage: tiled_bench.py [reads of 50 kb, default 64]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, synth
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = synth.Stream(11, 3)
for nm, lo, hi in ((30, 330, 350), (100, 400, 500), (12, 900, 1100), (5, 950, 1000)):
    anc = st.below(hi + 16, 4)
    ms = []
    for j in range(nm):
        L = lo + int(st.below(1, hi - lo + 1)[0])
        c = synth.mutate(anc, st, 0.15, 0.02, 0.02)
        while len(c) < L:
            c = np.concatenate([c, st.below(L, 4)])
        ms.append(synth._to_ascii(c[:L]))
    reads = []
    for r in range(NR):
        p, tot = [], 0
        while tot < 50000:
            j = int(st.below(1, nm)[0])
            codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(ms[j], dtype=np.uint8))
            x = synth._to_ascii(synth.mutate(codes, st, 0.05, 0.02, 0.02))
            p.append(x); tot += len(x)
        reads.append(b"".join(p)[:50000])
    shas = []
    for kern, name in ((lib.KERNEL_AUTO, "auto"), (lib.KERNEL_GENERIC, "generic")):
        nr = NR if kern == lib.KERNEL_AUTO else min(NR, 8)
        e = lib.Engine(ms, kernel=kern)
        e.load_reads(reads[:nr])
        e.run(); e.total_rows()
        t0 = time.perf_counter()
        for _ in range(3):
            e.run(); e.total_rows()
        dt = (time.perf_counter() - t0) / 3
        tm, info = e.timings(), e.info()
        shas.append(e.rows()[:min(NR, 8)])
        e.close()
        cells = info["rows"] * info["sum_template_len"]
        print("%3d monomers of %d-%d bp (%6d template cells): %-7s %-7s %-28s P=%-3d %7.1f ms per %.1f Mbp = %6.1f Mbp/s, %.2f Tcell/s (fill %.1f, traceback %.1f ms)" % (
            nm, lo, hi, info["sum_template_len"], name, info["family"], info["cells"], info["cells_per_lane"], dt * 1e3, nr * 0.05, nr * 0.05 / dt,
            cells / dt / 1e12, tm["fill_ms"], tm["trace_ms"]), flush=True)
    print("    same rows on the first %d reads: %s" % (min(NR, 8), shas[0] == shas[1]), flush=True)
