#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI on synthetic reads (developer tool).
usage: python tools/cli_bench.py [reads] [threads] [monomers] [--second-best]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = sys.argv[2] if len(sys.argv) > 2 else "32"
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 12
extra = [a for a in sys.argv[4:]]
mn, ms = synth.make_monomers(nm, seed=1)
rn, rs = synth.make_reads(ms, n, read_len=50000, seed=1)
d = tempfile.mkdtemp()
synth.write_fasta(os.path.join(d, "r.fa"), rn, rs, width=80)
synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
t0 = time.perf_counter()
p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "stringdecomposer"), os.path.join(d, "r.fa"),
                    os.path.join(d, "m.fa"), "-o", os.path.join(d, "out"), "-t", t] + extra,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
dt = time.perf_counter() - t0
print(p.stdout.decode()[-600:])
print("rc=%d  %d reads, %.1f Mbp: %.2f s end to end (%.2f Mbp/s)" % (p.returncode, n, n * 0.05, dt, n * 0.05 / dt))
