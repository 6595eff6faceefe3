import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from stringdecomposer_amd import lib, synth
NR = 128
st = synth.Stream(11, 3)
for nm, lo, hi in ((30, 330, 350), (100, 400, 500), (12, 900, 1100)):
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
    rows = {}
    for ed, flags, name in ((-1, 0, "no filter"), (40, lib.FLAG_NO_EDTHR_COMPACT, "--ed_thr 40, ranked (every template)"), (40, 0, "--ed_thr 40, compacted"),
                            (80, 0, "--ed_thr 80, compacted")):
        e = lib.Engine(ms, ed_thr=ed, flags=flags)
        e.load_reads(reads)
        e.run(); e.total_rows()
        t0 = time.perf_counter()
        for _ in range(3):
            e.run(); e.total_rows()
        dt = (time.perf_counter() - t0) / 3
        tm, info = e.timings(), e.info()
        rows[name] = e.rows()[:8]
        e.close()
        print("%3d x %d-%d  %-40s %7.1f ms (fill %.1f trace %.1f) launches %d" % (nm, lo, hi, name, dt*1e3, tm["fill_ms"], tm["trace_ms"], info["fill_launches"]), flush=True)
    print("    compacted rows == ranked rows:", rows["--ed_thr 40, compacted"] == rows["--ed_thr 40, ranked (every template)"])
