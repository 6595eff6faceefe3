#!/usr/bin/env python3
"""Developer tool (needs a GPU): replay one saved fuzz input against the oracle.
usage: python tools/replay_case.py <dir with r.fa, m.fa, params.json>"""
import json, sys, os
sys.path.insert(0, os.getcwd())
from stringdecomposer_amd import lib
from oracle import binding as oracle
d = sys.argv[1]
rn, rs, _ = lib.fasta_load(d + "/r.fa"); mn, ms, _ = lib.fasta_load(d + "/m.fa")
_p = json.load(open(d + "/params.json"))
sc, part, ov, ed = tuple(_p["scoring"]), _p["part_size"], _p["overlap"], _p["ed_thr"]
got = lib.decompose(rn, rs, mn, ms, scoring=sc, part_size=part, overlap=ov, ed_thr=ed)
exp = oracle.decompose(rn, rs, mn, ms, threads=4, sc=sc, part=part, overlap=ov, ed_thr=ed)
print("replay", d, "OK" if got == exp else "MISMATCH")
