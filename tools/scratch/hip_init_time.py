#!/usr/bin/env python3
"""Developer probe (GPU box): what a fresh process pays before its first kernel -- HIP runtime start, first
allocation, stream / event creation, loading this library, creating an engine."""
import ctypes as C, os, sys, time
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
t1 = time.perf_counter()
hip.hipInit(0)
t2 = time.perf_counter()
hip.hipSetDevice(0)
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20)
t3 = time.perf_counter()
s = C.c_void_p()
hip.hipStreamCreate(C.byref(s))
t4 = time.perf_counter()
print("dlopen libamdhip64 %.1f ms, hipInit %.1f ms, hipSetDevice + first hipMalloc %.1f ms, hipStreamCreate %.1f ms" % (
    (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t5 = time.perf_counter()
from stringdecomposer_amd import lib, synth
t6 = time.perf_counter()
mn, ms = synth.make_monomers(12, seed=1)
t7 = time.perf_counter()
e = lib.Engine(ms)
t8 = time.perf_counter()
rn, rs = synth.make_reads(ms, 20, read_len=50000, seed=1)
t9 = time.perf_counter()
e.load_reads(rs)
t10 = time.perf_counter()
e.run(); e.total_rows()
t11 = time.perf_counter()
e.run(); e.total_rows()
t12 = time.perf_counter()
print("import package + dlopen libsd_hip %.1f ms, Engine() %.1f ms, load_reads %.1f ms, first run %.1f ms, second run %.1f ms" % (
    (t6 - t5) * 1e3, (t8 - t7) * 1e3, (t10 - t9) * 1e3, (t11 - t10) * 1e3, (t12 - t11) * 1e3))
