#!/usr/bin/env python3
"""--second-best on a set of kilobase monomers, whole command-line path (sd_run_files): identities of the blocks on the
device (sd_nw_long.hip) against host threads (SD_NW_LONG_OFF=1).  usage: long_monomer_second_best.py [monomers] [len] [reads]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stringdecomposer_amd import lib, synth
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 64
st = synth.Stream(23, nm)
ms = [synth._to_ascii(st.below(L - 40 + int(st.below(1, 80)[0]), 4)) for _ in range(nm)]
mn = ["m%d" % j for j in range(nm)]
reads = []
for r in range(nr):
    p, tot = [], 0
    while tot < 50000:
        j = int(st.below(1, nm)[0])
        codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(ms[j], dtype=np.uint8))
        x = synth._to_ascii(synth.mutate(codes, st, 0.05, 0.02, 0.02))
        p.append(x); tot += len(x)
    reads.append(b"".join(p)[:50000])
rn = ["r%d" % i for i in range(nr)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    rfa, mfa = os.path.join(d, "r.fa"), os.path.join(d, "m.fa")
    synth.write_fasta(rfa, rn, reads)
    synth.write_fasta(mfa, mn, ms)
    outs = {}
    for tag in ("device", "host"):
        if tag == "host":
            os.environ["SD_NW_LONG_OFF"] = "1"
        else:
            os.environ.pop("SD_NW_LONG_OFF", None)
        best = None
        for rep in range(3):
            o = [os.path.join(d, "%s_%d_%s.tsv" % (tag, rep, x)) for x in ("raw", "final", "alt")]
            t0 = time.perf_counter()
            lib.run_files(rfa, mfa, o[0], o[1], o[2], second_best=True, threads=16)
            dt = time.perf_counter() - t0
            stt = lib.last_run_stats()
            if best is None or dt < best[0]:
                best = (dt, stt)
        outs[tag] = [open(x, "rb").read() for x in o]
        print("%d monomers of ~%d bp, %d reads x 50 kb, --second-best, identities on %s: %.1f ms per job (%.1f Mbp/s); identities %.1f ms, "
              "fill %.1f, wait %.1f" % (nm, L, nr, tag, best[0] * 1e3, nr * 0.05 / best[0], best[1]["text_identity_ms"], best[1]["fill_ms"], best[1]["wait_ms"]), flush=True)
    print("same files:", outs["device"] == outs["host"])
