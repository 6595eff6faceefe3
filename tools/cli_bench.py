#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI on synthetic reads (developer tool): FASTA files on disk -> raw, final
and _alt TSV files, whole process (interpreter start included).
usage: python tools/cli_bench.py [distinct reads] [threads] [monomers] [repeat] [width] [--second-best ...]
`repeat` writes every read that many times under different names (500 Mbp = 1000 x 10 without generating
10 000 reads); width 0 = single-line FASTA."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = sys.argv[2] if len(sys.argv) > 2 else "32"
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 1
width = int(sys.argv[5]) if len(sys.argv) > 5 else 80
extra = [a for a in sys.argv[6:]]
mn, ms = synth.make_monomers(nm, seed=1)
rn, rs = synth.make_reads(ms, n, read_len=50000, seed=1)
d = tempfile.mkdtemp()
synth.write_fasta(os.path.join(d, "r.fa"), ["%s_%d" % (x, k) for k in range(rep) for x in rn], rs * rep, width=width)
synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
for trial in range(2):
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "stringdecomposer"), os.path.join(d, "r.fa"),
                        os.path.join(d, "m.fa"), "-o", os.path.join(d, "out"), "-t", t] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, SD_TIMING="1"))
    dt = time.perf_counter() - t0
    print("\n".join(l for l in p.stdout.decode().splitlines() if "sd timing" in l or "rror" in l))
    sizes = {f: os.path.getsize(os.path.join(d, "out", f)) for f in sorted(os.listdir(os.path.join(d, "out")))}
    print("rc=%d  %d reads, %.1f Mbp, %s: %.2f s end to end (%.1f Mbp/s)  %s" % (
        p.returncode, n * rep, n * rep * 0.05, " ".join(extra) or "light", dt, n * rep * 0.05 / dt, sizes), flush=True)
