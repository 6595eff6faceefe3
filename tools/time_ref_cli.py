#!/usr/bin/env python3
"""CPU baseline of the --second-best leg (container only: needs /root/reference): the UNMODIFIED reference command
line (tests/golden/make_final_golden.py's recipe: the reference's bin/stringdecomposer, its dp binary, its vendored
edlib behind a ctypes shim) timed on a bounded sample of bench.py's C4 workload -- the first reads of the same
synthetic read set, 64 monomers, --second-best.  Writes profiles/ref_cli_c4_second_best.json, which
`bench.py --config c4-second-best` quotes as cpu_baseline (the reference cannot travel to the GPU box).

usage: python tools/time_ref_cli.py [reads=4] [threads=8]"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_final_golden as g  # noqa: E402
from stringdecomposer_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t = int(sys.argv[2]) if len(sys.argv) > 2 else 8
work = g.prepare()
try:
    mn, ms = synth.make_monomers(64, seed=1)
    rn, rs = synth.make_reads(ms, n, read_len=50000, seed=1)
    rf, mf = os.path.join(work, "reads.fa"), os.path.join(work, "monomers.fa")
    synth.write_fasta(rf, rn, rs, width=80)
    synth.write_fasta(mf, mn, ms)
    out = os.path.join(work, "out")
    env = dict(os.environ, PYTHONPATH=os.path.join(work, "shims"), SD_EDLIB_SO=g.ob.REF_EDLIB)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(work, "ref", "bin", "stringdecomposer"), rf, mf, "-o", out, "-t", str(t),
                        "--second-best"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise SystemExit(p.stdout.decode()[-2000:])
    # split: the dp subprocess alone
    t1 = time.perf_counter()
    subprocess.run([g.ob.REF_DP, rf, mf, str(t), "5000", "500", "-1", "-1", "-1", "1"], stdout=subprocess.DEVNULL, check=True)
    dp_s = time.perf_counter() - t1
    bp = sum(len(x) for x in rs)
    res = {"value": bp / dt, "unit": "bp/s", "cores": t, "kind": "reference",
           "sample": "%d reads x 50000 bp of bench.py's C4 read set (seed 1), 64 monomers, --second-best" % n,
           "seconds": round(dt, 2), "dp_seconds": round(dp_s, 2),
           "where": "build container (%d CPUs); the reference command line cannot travel to the GPU box" % (os.cpu_count() or 0),
           "recipe": "tools/time_ref_cli.py (tests/golden/make_final_golden.py: unmodified reference CLI + dp + vendored edlib via shims)",
           "final_rows": open(os.path.join(out, "final_decomposition.tsv")).read().count("\n")}
    with open(os.path.join(ROOT, "profiles", "ref_cli_c4_second_best.json"), "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print(json.dumps(res))
finally:
    shutil.rmtree(work, ignore_errors=True)
