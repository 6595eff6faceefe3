#!/usr/bin/env python3
"""BASELINE config 5's shape at full size against the REAL reference binary (GPU box; oracle/_ref/dp travels as a
binary): one sequence of <seq-len> bp, 12 synthetic monomers, scoring -2,-3,-4,2 -- the reference's raw TSV and the
library's, byte for byte.  usage: c5_full_parity.py [seq-len, default 200000000] [reference threads, default 16] [sha256]
With a sha256 (of a raw TSV an earlier run of this tool found identical to the reference's: profiles/r0N_c5_full_parity.json)
the 4 minutes of the reference are skipped and the library's file is compared with that hash.  Prints one JSON line."""
import hashlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import lib, synth
from oracle import binding as ob
L = int(sys.argv[1]) if len(sys.argv) > 1 else 200000000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
KNOWN = sys.argv[3] if len(sys.argv) > 3 else None
sc = (-2, -3, -4, 2)
if not KNOWN and not ob.have_ref_dp():
    raise SystemExit("oracle/_ref/dp is not here")
mn, ms = synth.make_monomers(12, seed=1)
_, rs = synth.make_reads(ms, 1, read_len=L, seed=1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
synth.write_fasta(os.path.join(d, "r.fa"), ["seq0"], rs, width=0)
synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
out = os.path.join(d, "raw.tsv")
t0 = time.perf_counter()
lib.decompose_files(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), out, scoring=sc, threads=16)
t1 = time.perf_counter()
lib.decompose_files(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), out, scoring=sc, threads=16)
t2 = time.perf_counter()
got = open(out, "rb").read()
t3 = time.perf_counter()
if KNOWN:
    times = []
    for _ in range(3):
        ta = time.perf_counter()
        lib.decompose_files(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), out, scoring=sc, threads=16)
        times.append(round(time.perf_counter() - ta, 4))
    print(json.dumps({"workload": "C5 shape: one sequence of %d bp" % L, "hip_files_to_raw_tsv_s": [round(t1 - t0, 3), round(t2 - t1, 3)] + times,
                      "rows": got.count(b"\n"), "raw_tsv_sha256": hashlib.sha256(got).hexdigest(),
                      "identical_to_known_reference_hash": hashlib.sha256(got).hexdigest() == KNOWN}))
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    raise SystemExit(0)
rc, ref, err = ob.run_ref_dp(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), threads=T, sc=sc)
t4 = time.perf_counter()
print(json.dumps({"workload": "C5 shape: one sequence of %d bp, 12 monomers, scoring %s, part 5000 / overlap 500" % (L, ",".join(map(str, sc))),
                  "hip_files_to_raw_tsv_s": [round(t1 - t0, 3), round(t2 - t1, 3)], "hip_bp_per_s": L / (t2 - t1),
                  "reference_dp_threads": T, "reference_rc": rc, "reference_s": round(t4 - t3, 1), "reference_bp_per_s": L / (t4 - t3),
                  "rows": got.count(b"\n"), "raw_tsv_bytes": len(got), "raw_tsv_sha256": hashlib.sha256(got).hexdigest(),
                  "identical_to_reference": rc == 0 and got == ref}))
import shutil
shutil.rmtree(d, ignore_errors=True)
