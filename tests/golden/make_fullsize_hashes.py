#!/usr/bin/env python3
"""sha256 of the raw TSV the REAL reference binary (oracle/_ref/dp, built from /root/reference by oracle/Makefile) prints for
the two full-size inputs of the GPU suite -- so that the driver-run tests compare EVERY row at full size with the reference,
not a sample:
  c2   BASELINE config 2: 1000 synthetic reads x 50 kb, 12 monomers, default scoring (tests/test_gpu_parity.py)
  c5   BASELINE config 5's shape: one 200-Mb sequence, scoring -2,-3,-4,2 (tests/test_gpu_configs.py)
Run in the build container (the GPU box has no /root/reference and needs none: the hashes are data):
    python tests/golden/make_fullsize_hashes.py [threads]      # ~15 minutes on 8 cores
Writes tests/golden/fullsize_sha256.json."""
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from stringdecomposer_amd import synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
assert ob.have_ref_dp(), "build oracle/_ref/dp first (make -C oracle ref)"
out = {}
with tempfile.TemporaryDirectory() as d:
    mn, ms = synth.make_monomers(12, seed=1)
    mfa = os.path.join(d, "m.fa")
    synth.write_fasta(mfa, mn, ms)
    # c2: the inputs of test_full_c2_* (seed 1)
    rn, rs = synth.make_reads(ms, 1000, read_len=50000, seed=1)
    rfa = os.path.join(d, "c2.fa")
    synth.write_fasta(rfa, rn, rs)
    t0 = time.time()
    rc, txt, err = ob.run_ref_dp(rfa, mfa, threads=T)
    assert rc == 0, err[-500:]
    out["c2"] = {"sha256": hashlib.sha256(txt).hexdigest(), "rows": txt.count(b"\n"), "bytes": len(txt),
                 "input": "synth.make_monomers(12, seed=1); synth.make_reads(ms, 1000, read_len=50000, seed=1)",
                 "reference_seconds": round(time.time() - t0, 1)}
    print("c2", out["c2"], flush=True)
    # c5: the sequence of test_c5_200mb_single_sequence_chunk_ranges_and_oracle
    _, r2 = synth.make_reads(ms, 1, read_len=2_000_000, seed=7)
    seq = (r2[0] * 101)[:200_000_000]
    rfa = os.path.join(d, "c5.fa")
    synth.write_fasta(rfa, ["chr"], [seq])
    t0 = time.time()
    rc, txt, err = ob.run_ref_dp(rfa, mfa, threads=T, sc=(-2, -3, -4, 2))
    assert rc == 0, err[-500:]
    out["c5"] = {"sha256": hashlib.sha256(txt).hexdigest(), "rows": txt.count(b"\n"), "bytes": len(txt),
                 "input": "synth.make_reads(ms, 1, read_len=2_000_000, seed=7)[0] repeated to 200 000 000 bp, name 'chr', scoring -2,-3,-4,2",
                 "reference_seconds": round(time.time() - t0, 1)}
    print("c5", out["c5"], flush=True)
out["made_by"] = "tests/golden/make_fullsize_hashes.py: oracle/_ref/dp -t %d (the reference's main.cpp + edlib.cpp compiled by oracle/Makefile)" % T
with open(os.path.join(ROOT, "tests", "golden", "fullsize_sha256.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
