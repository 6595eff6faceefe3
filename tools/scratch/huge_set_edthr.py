#!/usr/bin/env python3
"""Developer probe (GPU box): a monomer set beyond eight waves (700 synthetic monomers = 1 400 templates) with --ed_thr:
the filter-only form of the tiled layout against the generic family (SD_FLAG_NO_EDTHR_COMPACT), same rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stringdecomposer_amd import lib, synth
NM = int(sys.argv[1]) if len(sys.argv) > 1 else 700
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mn, ms = synth.make_monomers(NM, seed=2)
rn, rs = synth.make_reads(ms, NR, read_len=50000, seed=3)
rows = {}
cases = ((20, 0, "--ed_thr 20, filter-only compacted"), (20, lib.FLAG_NO_EDTHR_COMPACT, "--ed_thr 20, generic family"), (-1, 0, "no filter, generic family"))
if os.environ.get("SD_HUGE_ONLY_FAST"):   # (profiling the fast form only)
    cases = cases[:1]
for ed, flags, name in cases:
    nr = NR if flags == 0 and ed >= 0 else min(NR, 4)
    e = lib.Engine(ms, ed_thr=ed, flags=flags)
    e.load_reads(rs[:nr])
    e.run(); e.total_rows()
    t0 = time.perf_counter()
    for _ in range(2):
        e.run(); e.total_rows()
    dt = (time.perf_counter() - t0) / 2
    tm, info = e.timings(), e.info()
    rows[name] = e.rows()[:4]
    e.close()
    print("%d monomers: %-38s %-8s %8.1f ms per %.2f Mbp = %7.1f Mbp/s (fill %.1f trace %.1f)" % (NM, name, info["family"], dt * 1e3, nr * 0.05, nr * 0.05 / dt, tm["fill_ms"], tm["trace_ms"]), flush=True)
if len(cases) > 1:
    print("same rows (first 4 reads), filter-only vs generic with the filter:", rows["--ed_thr 20, filter-only compacted"] == rows["--ed_thr 20, generic family"])
