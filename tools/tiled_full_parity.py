#!/usr/bin/env python3
"""Sets of the tiled multi-wave layout (csrc/sd_fast_wt.hip) against the REAL reference binary on the GPU box
(oracle/_ref/dp), byte for byte: the four sets of tools/tiled_bench.py, <reads> reads of 50 kb each, default scoring
and --ed_thr 40.  usage: tiled_full_parity.py [reads, default 32] [reference threads, default 16]"""
import hashlib, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, synth
from oracle import binding as ob
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if not ob.have_ref_dp():
    raise SystemExit("oracle/_ref/dp is not here")
st = synth.Stream(11, 3)
res = []
for nm, lo, hi in ((30, 330, 350), (100, 400, 500), (12, 900, 1100), (5, 950, 1000)):
    anc = st.below(hi + 16, 4)
    ms = []
    for j in range(nm):
        L = lo + int(st.below(1, hi - lo + 1)[0])
        c = synth.mutate(anc, st, 0.15, 0.02, 0.02)
        while len(c) < L:
            c = np.concatenate([c, st.below(L, 4)])
        ms.append(synth._to_ascii(c[:L]))
    reads = []
    for r in range(NR):
        p, tot = [], 0
        while tot < 50000:
            j = int(st.below(1, nm)[0])
            codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(ms[j], dtype=np.uint8))
            x = synth._to_ascii(synth.mutate(codes, st, 0.05, 0.02, 0.02))
            p.append(x); tot += len(x)
        reads.append(b"".join(p)[:50000])
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    synth.write_fasta(os.path.join(d, "r.fa"), ["r%d" % i for i in range(NR)], reads, width=80)
    synth.write_fasta(os.path.join(d, "m.fa"), ["m%d" % i for i in range(nm)], ms)
    pi = lib.plan_info(ms)
    for ed in (None, 40):
        out = os.path.join(d, "raw.tsv")
        kw = {} if ed is None else {"ed_thr": ed}
        t0 = time.perf_counter()
        lib.decompose_files(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), out, threads=16, **kw)
        t1 = time.perf_counter()
        got = open(out, "rb").read()
        rc, ref, err = ob.run_ref_dp(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), threads=T, ed_thr=ed)
        t2 = time.perf_counter()
        res.append({"set": "%d monomers of %d-%d bp" % (nm, lo, hi), "layout": "%s, P = %d, %d waves" % (pi["cells"], pi["cells_per_lane"], pi["waves"]),
                    "ed_thr": ed, "reads": NR, "hip_s": round(t1 - t0, 3), "reference_s": round(t2 - t1, 1), "rows": got.count(b"\n"),
                    "raw_tsv_sha256": hashlib.sha256(got).hexdigest()[:16], "identical_to_reference": rc == 0 and got == ref})
        print(json.dumps(res[-1]), flush=True)
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps({"all_identical": all(r["identical_to_reference"] for r in res), "cases": len(res)}))
