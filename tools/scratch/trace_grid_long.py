import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from stringdecomposer_amd import lib, synth
NR = 128
st = synth.Stream(11, 3)
for nm, lo, hi in ((12, 900, 1100), (5, 950, 1000), (3, 1500, 1800)):
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
    for g in (0, 1536, 2048):
        if g: os.environ["SD_TRACE_GRID"] = str(g)
        else: os.environ.pop("SD_TRACE_GRID", None)
        e = lib.Engine(ms)
        e.load_reads(reads)
        e.run(); e.total_rows()
        for _ in range(3):
            e.run(); e.total_rows()
        tm, info = e.timings(), e.info()
        e.close()
        print("%3d x %d-%d  grid %4d  fill %.1f trace %.2f  %s" % (nm, lo, hi, g, tm["fill_ms"], tm["trace_ms"], info["cells"]), flush=True)
