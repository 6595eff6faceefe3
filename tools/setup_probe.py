import time, sys, os
sys.path.insert(0, os.getcwd())
from stringdecomposer_amd import lib, synth
mn, ms = synth.make_monomers(64, seed=1)
for k in range(4):
    t0 = time.perf_counter(); e = lib.Engine(ms); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
    print("engine create %.1f ms, destroy %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
mn, ms = synth.make_monomers(12, seed=1)
for k in range(3):
    t0 = time.perf_counter(); e = lib.Engine(ms); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
    print("12 monomers: engine create %.1f ms, destroy %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
