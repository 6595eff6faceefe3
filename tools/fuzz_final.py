#!/usr/bin/env python3
"""Randomised campaign of the post-DP half of the command line (developer tool, needs a GPU):
random monomer sets (repeated names, N, short and long monomers) and reads through the native streaming
call sd_run_files (device DP, device identities, C++ post-processing) against
  * the CPU oracle for the raw TSV,
  * the Python implementation of convert_tsv (stringdecomposer_amd.main, host NW) for final / _alt,
  * edlib itself (oracle/_ref/libedlib.so, when built) for a sample of the device identities.
usage: python tools/fuzz_final.py [cases] [seed] [log file to append the summary to]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from stringdecomposer_amd import lib, synth, main as sdmain
from oracle import binding as oracle
import edlib_ref

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st = synth.Stream(seed, 4242)


def rnd(k):
    return int(st.below(1, k)[0])


def rand_seq(n):
    return synth._to_ascii(st.below(n, 4))


bad = 0
checked_pairs = 0
t0 = time.time()
tmp = tempfile.mkdtemp(prefix="sd_fuzz_final_")
for case in range(cases):
    nm = 1 + rnd(10)
    lo, hi = [(20, 60), (100, 200), (160, 180), (250, 420), (2, 12), (600, 1400)][rnd(6)]   # the last: host identities (> 512 bp)
    anc = st.below(hi + 8, 4)
    ms, mn = [], []
    for j in range(nm):
        L = lo + rnd(hi - lo + 1)
        codes = synth.mutate(anc, st, 0.03 + 0.2 * st.uniform(1)[0], 0.02, 0.02)
        while len(codes) < L:
            codes = np.concatenate([codes, st.below(L, 4)])
        m = bytearray(synth._to_ascii(codes[:L]))
        if rnd(6) == 0 and L > 2:
            m[rnd(L)] = ord("N")
        ms.append(bytes(m))
        mn.append("m%d" % (rnd(j + 1) if rnd(5) == 0 else j))   # sometimes a repeated name (dict semantics)
    reads, rn = [], []
    for r in range(1 + rnd(4)):
        want = [30 + rnd(300), 500 + rnd(3000), 6000 + rnd(9000)][rnd(3)]
        parts, tot = [], 0
        while tot < want:
            j = rnd(nm)
            codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8),
                                    np.frombuffer(ms[j].replace(b"N", b"C"), dtype=np.uint8))
            x = synth._to_ascii(synth.mutate(codes, st, 0.1 * st.uniform(1)[0], 0.05 * st.uniform(1)[0],
                                             0.05 * st.uniform(1)[0])) or b"A"
            if rnd(2):
                x = synth.revcomp_bytes(x)
            if rnd(7) == 0:
                x = bytes([x[0]]) * (1 + rnd(40)) + x          # homopolymer run: the compressed identities differ
            parts.append(x)
            if rnd(6) == 0:
                parts.append(rand_seq(1 + rnd(80)))
            tot = sum(len(p) for p in parts)
        b = bytearray(b"".join(parts))
        if rnd(4) == 0:
            for p in st.below(1 + rnd(10), len(b)):
                b[int(p)] = ord("N")
        reads.append(bytes(b))
        rn.append("read_%d some description" % r if rnd(3) == 0 else "r%d" % r)
    part, ov = [(5000, 500), (700, 100), (333, 77)][rnd(3)]
    sb = rnd(3) != 0
    thr = [0, 0, 70, 90][rnd(4)]
    rfa, mfa = os.path.join(tmp, "r.fa"), os.path.join(tmp, "m.fa")
    synth.write_fasta(rfa, rn, reads)
    synth.write_fasta(mfa, mn, ms)
    raw, fin = os.path.join(tmp, "o_raw.tsv"), os.path.join(tmp, "o.tsv")
    alt = fin[:-4] + "_alt.tsv"
    try:
        lib.run_files(rfa, mfa, raw, fin, alt, thr, sb, part_size=part, overlap=ov, threads=1 + rnd(8))
    except lib.SdError as e:
        if e.code == lib.SD_ERR_UNSUPPORTED:
            continue
        raise
    why = None
    rmap = sdmain.load_fasta(rfa, "map")
    mons = sdmain.add_rc_monomers(sdmain.load_fasta(mfa))
    exp_raw = oracle.decompose_files(rfa, mfa, threads=8, part=part, overlap=ov)
    got_raw = open(raw, "rb").read()
    if got_raw != exp_raw:
        why = "raw"
    else:
        pfin = os.path.join(tmp, "p.tsv")
        sdmain.convert_tsv(got_raw.decode(), rmap, mons, pfin, thr, not sb, threads=4)
        if open(fin, "rb").read() != open(pfin, "rb").read():
            why = "final"
        elif open(alt, "rb").read() != open(pfin[:-4] + "_alt.tsv", "rb").read():
            why = "alt"
    if why is None and edlib_ref.have_edlib():
        # device identities of a few blocks against edlib itself
        rows = [l.split("\t") for l in got_raw.decode().split("\n")[:-1]]
        for x in st.below(min(6, len(rows)), max(1, len(rows))) if rows else []:
            rw = rows[int(x)]
            rd = rmap[rw[0]].seq.encode()
            seg = rd[int(rw[2]):int(rw[3]) + 1]
            if not seg:
                continue
            starts, ends = np.array([int(rw[2])], dtype=np.int64), np.array([int(rw[3])], dtype=np.int64)
            for hom in (False, True):
                d, m, c = lib.identity_segments(rd, starts, ends, [q.seq for q in mons], hom, 1, device=0)
                for t, q in enumerate(mons):
                    a, b = (edlib_ref.homo(seg.decode()), edlib_ref.homo(q.seq)) if hom else (seg.decode(), q.seq)
                    e = edlib_ref.nw(a, b)
                    checked_pairs += 1
                    if (int(d[0][t]), int(m[0][t]), int(c[0][t])) != tuple(int(v) for v in e):
                        why = "edlib pair"
    if why:
        bad += 1
        print("MISMATCH case", case, why, "monomers", [len(m) for m in ms], mn, "reads", [len(r) for r in reads],
              part, ov, "second_best", sb, "thr", thr, flush=True)
        d = os.path.join(ROOT, "gpurun_out", "fuzz_final_fail_%d_%d" % (seed, case))
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(os.path.join(d, "r.fa"), rn, reads)
        synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
        open(os.path.join(d, "params.txt"), "w").write(repr((part, ov, sb, thr)))
summary = "fuzz_final: seed %d, %d cases, %d mismatches, %d device identities == edlib, %.1fs" % (
    seed, cases, bad, checked_pairs, time.time() - t0)
print(summary)
if len(sys.argv) > 3:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[3])), exist_ok=True)
    with open(sys.argv[3], "a") as f:
        f.write("%s  %s\n" % (time.strftime("%Y-%m-%d %H:%M:%S"), summary))
sys.exit(1 if bad else 0)
