#!/usr/bin/env python3
"""Developer A/B harness: fill / traceback kernel times of one library build on the C2 workload.
usage: SD_HIP_LIB=<path/to/libsd_hip_variant.so> python tools/kbench.py [reads] [steps] [monomers]
Checks the rows against a reference run of the default library when SD_KBENCH_CHECK=1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import lib, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mono = int(sys.argv[3]) if len(sys.argv) > 3 else 12
mn, ms = synth.make_monomers(mono, seed=1)
rn, rs = synth.make_reads(ms, reads, read_len=50000, seed=1)
kw = {}
if os.environ.get("SD_SCORING"):   # e.g. SD_SCORING=-2,-3,-4,2 (ins,del,mismatch,match)
    kw["scoring"] = tuple(int(x) for x in os.environ["SD_SCORING"].split(","))
e = lib.Engine(ms, kernel={"auto": 0, "generic": 1, "fast": 2}[os.environ.get("SD_KERNEL", "auto")], **kw)
e.load_reads(rs)
best = None
for _ in range(steps + 1):
    t0 = time.perf_counter()
    e.run()
    n = e.total_rows()
    dt = time.perf_counter() - t0
    t = e.timings()
    if best is None or t["fill_ms"] < best["fill_ms"]:
        best = dict(t, wall_ms=dt * 1e3)
import hashlib
recs = e.fetch()
h = hashlib.sha1(repr(recs).encode()).hexdigest()[:12]
print("%-40s fill %.2f ms  trace %.2f ms  wall %.2f ms  rows %d  P=%s  sha %s" % (
    os.path.basename(lib.LIB_PATH), best["fill_ms"], best["trace_ms"], best["wall_ms"], n,
    e.info()["cells_per_lane"], h), e.info()["cells"], "FL", e.info()["floor_slots"], e.info()["trace"])
