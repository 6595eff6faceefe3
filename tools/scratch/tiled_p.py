import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from stringdecomposer_amd import lib, synth
NR = 128
st = synth.Stream(11, 3)
for nm, lo, hi in ((30, 330, 350), (100, 400, 500), (12, 900, 1100), (20, 700, 1100), (70, 300, 480)):
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
    for P in (0, 96, 128, 160, 192, 224):
        if P: os.environ["SD_TILED_P"] = str(P)
        else: os.environ.pop("SD_TILED_P", None)
        try:
            e = lib.Engine(ms)
        except Exception as ex:
            print(nm, P, "n/a", str(ex)[:60]); continue
        if e.info()["family"] != "fast":
            print(nm, P, "generic"); e.close(); continue
        e.load_reads(reads)
        e.run(); e.total_rows()
        t0 = time.perf_counter()
        for _ in range(3):
            e.run(); e.total_rows()
        dt = (time.perf_counter() - t0) / 3
        tm, info = e.timings(), e.info()
        pi = lib.plan_info(ms)
        e.close()
        cells = info["rows"] * info["sum_template_len"]
        print("%3d x %d-%d  P=%3d W=%d  %7.1f ms  %.2f Tcell/s (fill %.1f trace %.1f)" % (nm, lo, hi, info["cells_per_lane"], pi["waves"], dt*1e3, cells/dt/1e12, tm["fill_ms"], tm["trace_ms"]), flush=True)
