#!/usr/bin/env python3
"""sha256 of the raw TSV the REAL reference binary (oracle/_ref/dp, built from /root/reference by oracle/Makefile) prints for
the full-size inputs of the GPU suite -- so that the driver-run tests compare EVERY row at full size with the reference,
not a sample:
  c2   BASELINE config 2: 1000 synthetic reads x 50 kb, 12 monomers, default scoring (tests/test_gpu_parity.py)
  c4   BASELINE config 4: 64 monomers (128 templates) x 256 reads x 50 kb, default scoring (tests/test_gpu_configs.py)
  c5   BASELINE config 5's shape: one 200-Mb sequence, scoring -2,-3,-4,2 (tests/test_gpu_configs.py).  Since round 6 the
       sequence is 100 INDEPENDENT 2-Mb synthetic reads end to end: 40 000 distinct chunks (rounds 4-5 tiled one 2-Mb read,
       whose chunks repeat with period 400 because 2 000 000 is a multiple of the part size)
Run in the build container (the GPU box has no /root/reference and needs none: the hashes are data):
    python tests/golden/make_fullsize_hashes.py [threads] [c2,c4,c5]     # ~5 + ~25 + ~21 minutes on 8 cores
Writes tests/golden/fullsize_sha256.json (entries of configs that were not asked for are kept)."""
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
WHICH = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2", "c4", "c5"]
PATH = os.path.join(ROOT, "tests", "golden", "fullsize_sha256.json")
assert ob.have_ref_dp(), "build oracle/_ref/dp first (make -C oracle ref)"
out = json.load(open(PATH)) if os.path.exists(PATH) else {}


def c5_sequence(ms, mb=200):
    """The 200-Mb sequence of the C5 tests: 100 independent 2-Mb reads (seed 7) end to end."""
    _, rs = synth.make_reads(ms, mb // 2, read_len=2_000_000, seed=7)
    return b"".join(rs)


def entry(key, rfa, mfa, what, sc=None):
    t0 = time.time()
    rc, txt, err = ob.run_ref_dp(rfa, mfa, threads=T, sc=sc)
    assert rc == 0, err[-500:]
    out[key] = {"sha256": hashlib.sha256(txt).hexdigest(), "rows": txt.count(b"\n"), "bytes": len(txt), "input": what,
                "reference_seconds": round(time.time() - t0, 1), "reference_threads": T}
    print(key, out[key], flush=True)
    with open(PATH, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


with tempfile.TemporaryDirectory() as d:
    mn, ms = synth.make_monomers(12, seed=1)
    mfa = os.path.join(d, "m.fa")
    synth.write_fasta(mfa, mn, ms)
    if "c2" in WHICH:     # the inputs of test_full_c2_* (seed 1)
        rn, rs = synth.make_reads(ms, 1000, read_len=50000, seed=1)
        rfa = os.path.join(d, "c2.fa")
        synth.write_fasta(rfa, rn, rs)
        entry("c2", rfa, mfa, "synth.make_monomers(12, seed=1); synth.make_reads(ms, 1000, read_len=50000, seed=1)")
    if "c5" in WHICH:     # the sequence of test_c5_200mb_single_sequence_chunk_ranges_and_oracle
        rfa = os.path.join(d, "c5.fa")
        synth.write_fasta(rfa, ["chr"], [c5_sequence(ms)])
        entry("c5", rfa, mfa, "b''.join(synth.make_reads(ms, 100, read_len=2_000_000, seed=7)[1]) = 200 000 000 bp of distinct "
              "chunks, name 'chr', scoring -2,-3,-4,2", sc=(-2, -3, -4, 2))
    if "c4" in WHICH:     # the inputs of test_c4_full_size_raw_tsv_equals_the_reference_binary
        mn4, ms4 = synth.make_monomers(64, seed=11)
        m4 = os.path.join(d, "m64.fa")
        synth.write_fasta(m4, mn4, ms4)
        rn, rs = synth.make_reads(ms4, 256, read_len=50000, seed=14)
        rfa = os.path.join(d, "c4.fa")
        synth.write_fasta(rfa, rn, rs)
        entry("c4", rfa, m4, "synth.make_monomers(64, seed=11); synth.make_reads(ms4, 256, read_len=50000, seed=14)")
out["made_by"] = ("tests/golden/make_fullsize_hashes.py: oracle/_ref/dp (the reference's main.cpp + edlib.cpp compiled by "
                  "oracle/Makefile), thread count per entry")
with open(PATH, "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
