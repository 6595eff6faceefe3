import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stringdecomposer_amd import lib, synth
from oracle import binding as oracle
oracle.build()
ms = [b"A", b"ACGTACGTTGCA"]
mn = ["m0", "m1"]
for sc in [(-2, -3, -4, 2), (0, -1, -1, 1)]:
    print(lib.plan_info(ms, scoring=sc))
    for reads in ([b"ACGTTGCA" * 3], [b"A" * 30], [b"G"], [b"AC"], [b"ACGTACGTTGCAACGT"]):
        rn = ["r%d" % i for i in range(len(reads))]
        exp = oracle.decompose(rn, reads, mn, ms, threads=1, sc=sc)
        for fl, nm in ((0, "u16"), (lib.FLAG_NO_U16, "legacy"), (lib.FLAG_TRACE_V1, "u16+v1")):
            got = lib.decompose(rn, reads, mn, ms, kernel=lib.KERNEL_FAST, scoring=sc, flags=fl)
            print(sc, reads[0][:20], nm, "OK" if got == exp else "DIFF")
            if got != exp and nm == "u16":
                print(" exp:", exp.decode().replace("\n", " | ")[:600])
                print(" got:", got.decode().replace("\n", " | ")[:600])
