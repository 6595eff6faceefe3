#!/usr/bin/env python3
"""Instruction mix of the largest basic block (the row loop) of a kernel in a hipcc -S listing.
usage: isa_stats.py file.s <substring of the mangled kernel name> [...]   (developer tool)"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r"^(\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(name), s, re.S | re.M)
    if not m:
        print("not found:", name)
        continue
    full, body = m.group(1), m.group(2)
    blocks = re.split(r"\n\.LBB\d+_\d+:[^\n]*\n", body)
    big = max(blocks, key=len)
    ins = [l.strip().split()[0] for l in big.split("\n")
           if l.strip() and not l.strip().startswith((";", ".", "/"))]
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(full[:60], "| largest block:", len(ins), "instr, VALU", valu)
    print("  ", c.most_common(22))
    md = re.search(r"\.amdhsa_kernel %s.*?\.end_amdhsa_kernel" % re.escape(full), s, re.S)
    if md:
        for key in ["next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size"]:
            print("  ", key, re.search(key + r"\s+(\d+)", md.group(0)).group(1))
