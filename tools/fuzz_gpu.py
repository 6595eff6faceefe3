#!/usr/bin/env python3
"""Randomised parity campaign (developer tool, needs a GPU): random template sets, scorings,
chunk sizes, N densities and --ed_thr values through libsd_hip vs the CPU oracle.
usage: python tools/fuzz_gpu.py [cases] [seed] [log file to append the summary to]"""
import json
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stringdecomposer_amd import lib, synth
from oracle import binding as oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st = synth.Stream(seed, 99)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rnd(k):
    return int(st.below(1, k)[0])


def monomers(n, lo, hi, pn):
    anc = st.below(hi + 16, 4)
    out = []
    for j in range(n):
        L = lo + rnd(hi - lo + 1)
        codes = synth.mutate(anc, st, 0.05 + 0.2 * st.uniform(1)[0], 0.02, 0.02)
        while len(codes) < L:
            codes = np.concatenate([codes, st.below(L, 4)])
        m = bytearray(synth._to_ascii(codes[:L]))
        if st.uniform(1)[0] < pn and L > 2:
            m[rnd(L)] = ord("N")
        out.append(bytes(m))
    return out


bad = 0
t0 = time.time()
fam = {"fast": 0, "generic": 0}
for case in range(cases):
    shape = [(1, 5, 40), (3, 20, 90), (6, 100, 200), (12, 160, 180), (25, 150, 190), (50, 100, 180),
             (64, 165, 178), (4, 300, 500), (2, 1, 30), (10, 60, 64), (90, 100, 180), (230, 140, 176),
             (16, 150, 190), (20, 160, 180), (14, 120, 175),             # these three: P = 42..64 (sd_fast_fl_long.hip)
             (3, 520, 1000), (2, 1000, 2000), (1, 600, 2040),            # round 3: monomers of up to 2048 bp on the fast family
             (30, 230, 420), (10, 600, 1150), (70, 90, 330), (40, 1, 500),           # round 4: the tiled multi-wave layout (sd_fast_wt.hip)
             (560, 60, 120), (300, 150, 400)][rnd(24)]   # beyond eight waves: generic family, or with --ed_thr the filter-only form
    nm = max(1, shape[0] - rnd(2))
    ms = monomers(nm, shape[1], shape[2], 0.15 if rnd(3) == 0 else 0.0)
    mn = ["m%d" % j for j in range(nm)]
    sc = [(-1, -1, -1, 1), (-2, -3, -4, 2), (-1, -2, -1, 1), (0, 0, 0, 1), (-3, -1, -2, 3), (-5, -1, -3, 4),
          (-1, -1, -2, 2), (-2, -2, -1, 1), (-9, -7, -8, 9), (-1, -1, 2, 1), (-1, 0, -1, 1), (0, -1, -1, 1),
          (-1, -1, -1, -1), (-4, -4, 3, 3)][rnd(14)]
    if rnd(3) == 0:   # random scorings: exercises both cell formats (fp16 / int16) around the range switch
        sc = (-rnd(9), -rnd(9), rnd(13) - 8, rnd(13) - 2)
    if rnd(7) == 0:   # round 6: a positive insertion score (the fast family takes it; a positive deletion score stays generic)
        sc = (1 + rnd(3), -rnd(5), rnd(9) - 6, rnd(6))
    if rnd(4) == 0:   # the largest |del| that still selects fp16 cells for this set (csrc/sd_fast.hip: ub <= 2040)
        Lmax, i_, mm_, ma_ = max(len(m) for m in ms), rnd(3), rnd(7) - 5, rnd(5) - 2
        def ub(d):
            return ((Lmax - 1) * d + 129 * max(0, max(mm_, ma_) + d) + 128 * i_ + 8 * max(i_, d, abs(mm_), abs(ma_)) + 8)
        d = max([x for x in range(0, 40) if ub(x) <= 2040] or [0])
        sc = (-i_, -d, mm_, ma_)
    elif rnd(5) == 0:   # round 6: the largest match score still inside the biased-u16 window of the narrow layout
        Lmax, i_, d_, mm_ = max(len(m) for m in ms), rnd(4), rnd(6), rnd(9) - 6   # (csrc/sd_fast.hip: ub_of(128) <= u16_room)
        def fits(ma_):
            mx = max(i_, d_, abs(mm_), abs(ma_))
            ub = (Lmax - 1) * d_ + 129 * max(0, max(mm_, ma_) + d_) + 128 * i_ + 8 * mx + 8
            return ub <= 15800 - (Lmax + 1) * d_ - 8 * mx
        ok_ = [x for x in range(1, 120) if fits(x)]
        if ok_:
            sc = (-i_, -d_, mm_, max(ok_) - rnd(2))
    part, ov = [(5000, 500), (700, 100), (333, 77), (150, 20), (5000, 0)][rnd(5)]
    reads = []
    for r in range(1 + rnd(3)):
        parts, tot, want = [], 0, [50 + rnd(1500), 1 + rnd(40), 3000 + rnd(4000)][rnd(3) if rnd(4) else 0]
        while tot < want:
            j = rnd(nm)
            codes = np.searchsorted(ACGT, np.frombuffer(ms[j].replace(b"N", b"C"), dtype=np.uint8))
            x = synth._to_ascii(synth.mutate(codes, st, 0.08 * st.uniform(1)[0], 0.04 * st.uniform(1)[0], 0.04 * st.uniform(1)[0])) or b"A"
            if rnd(2):
                x = synth.revcomp_bytes(x)
            parts.append(x)
            if rnd(5) == 0:
                parts.append(synth._to_ascii(st.below(1 + rnd(60), 4)))
            tot = sum(len(p) for p in parts)
        kind = rnd(8)
        if kind == 0:      # exact repeats: fastest score growth
            parts = [ms[rnd(nm)].replace(b"N", b"A")] * (1 + want // max(1, len(ms[0])))
        elif kind == 1:    # unrelated sequence: insertion / mismatch dominated
            parts = [synth._to_ascii(st.below(want, 4))]
        elif kind == 2:    # homopolymer runs
            parts = [bytes([b"ACGT"[rnd(4)]]) * (1 + rnd(300)) for _ in range(1 + want // 150)]
        b = bytearray(b"".join(parts))
        if rnd(3) == 0:
            for p in st.below(1 + rnd(20), len(b)):
                b[int(p)] = ord("N")
        reads.append(bytes(b))
    rn = ["r%d" % i for i in range(len(reads))]
    ed = -1 if rnd(3) else rnd(80)
    if shape[0] >= 300 and rnd(2):   # the big sets mostly with the filter (their fast form exists only then); sometimes one that keeps everything
        ed = rnd(60) if rnd(6) else 400
    try:
        # (one run in four with the narrow layout's cell formats of rounds 1-5: fp16 / int16 instead of biased u16)
        got = lib.decompose(rn, reads, mn, ms, scoring=sc, part_size=part, overlap=ov, ed_thr=ed,
                            threads=1 + rnd(6), max_batch_rows=[0, 0, 200, 1500][rnd(4)],
                            flags=lib.FLAG_NO_U16 if rnd(4) == 0 else 0)
    except lib.SdError as e:
        if e.code == lib.SD_ERR_UNSUPPORTED:
            continue
        raise
    exp = oracle.decompose(rn, reads, mn, ms, threads=32, sc=sc, part=part, overlap=ov, ed_thr=ed)
    if got != exp:
        bad += 1
        print("MISMATCH case", case, "shape", shape, nm, [len(m) for m in ms][:8], sc, part, ov, "ed", ed, flush=True)
        d = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d" % (seed, case))
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(os.path.join(d, "r.fa"), rn, reads)
        synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
        json.dump({"scoring": list(sc), "part_size": part, "overlap": ov, "ed_thr": ed}, open(os.path.join(d, "params.json"), "w"))
summary = "fuzz: seed %d, %d cases, %d mismatches, %.1fs" % (seed, cases, bad, time.time() - t0)
print(summary)
if len(sys.argv) > 3:   # append to a log (copied to profiles/ after the run): which build, which seeds
    import hashlib
    src = hashlib.sha256()
    cs = os.path.join(ROOT, "stringdecomposer_amd", "csrc")
    for fn in sorted(os.listdir(cs)):
        if fn.endswith((".hip", ".hpp")):
            src.update(open(os.path.join(cs, fn), "rb").read())
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[3])), exist_ok=True)
    with open(sys.argv[3], "a") as f:
        f.write("%s  csrc-sha256 %s  %s\n" % (time.strftime("%Y-%m-%d %H:%M:%S"), src.hexdigest()[:16], summary))
sys.exit(1 if bad else 0)
