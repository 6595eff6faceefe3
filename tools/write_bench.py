#!/usr/bin/env python3
"""Time sd::write_parts (the file writer of sd_run_files) alone: n parts of b bytes appended twice to a file.
usage: write_bench.py <path> [n_parts] [part_bytes] [threads]   (SD_WRITE_PATH=1: the pwritev path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stringdecomposer_amd import lib
path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 140
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
th = int(sys.argv[4]) if len(sys.argv) > 4 else 16
for rep in range(4):
    t0 = time.perf_counter()
    wrote, ram = lib.write_parts_selftest(path, n, b, threads=th)
    dt = time.perf_counter() - t0
    os.unlink(path)
    print("%s: %d MB in %.1f ms (incl. generating + reading back), tmpfs=%s" % (path, wrote >> 20, dt * 1e3, ram))
