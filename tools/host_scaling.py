#!/usr/bin/env python3
"""Host stages of the hot path under N concurrent processes (developer tool, CPU only): what an N-GPU node's
host has to sustain.  Every process packs and assembles the C2 read set (1000 x 50 kb) `iters` times after a
common start; prints per-process and aggregate rates for N = 1, 2, 4, 8.
usage: python tools/host_scaling.py [threads per process] [reads] [iters]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(k, threads, reads, iters, bar, q):
    from stringdecomposer_amd import lib as L, synth as S
    mn, ms = S.make_monomers(12, seed=1)
    rn, rs = S.make_reads(ms, reads, read_len=50000, seed=100 + k)
    rset = L.ReadSet(rs)
    L.host_stage_rates(rset, iters=1, threads=threads)   # threads of the pool exist, pages touched
    bar.wait()
    t0 = time.perf_counter()
    r = L.host_stage_rates(rset, iters=iters, threads=threads)
    q.put((k, r, time.perf_counter() - t0))


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ctx = mp.get_context("spawn")
    print("host cores: %d, threads per process: %d, %d reads x 50 kb, %d iterations" % (os.cpu_count(), threads, reads, iters))
    for n in (1, 2, 4, 8):
        bar, q = ctx.Barrier(n), ctx.Queue()
        ps = [ctx.Process(target=worker, args=(k, threads, reads, iters, bar, q)) for k in range(n)]
        for p in ps:
            p.start()
        res = [q.get(timeout=900) for _ in ps]
        for p in ps:
            p.join(60)
        pack = [r[1]["pack_bp_per_s"] for r in res]
        asm = [r[1]["assemble_format_bp_per_s"] for r in res]
        step_ms = [reads * 50000 / a * 1e3 + reads * 50000 / b * 1e3 for a, b in zip(pack, asm)]
        print("N=%d  pack %.1f Gbp/s aggregate (min per process %.2f)  assemble+text %.1f Gbp/s aggregate (min %.2f)  "
              "-> host ms per 50-Mbp step and process: pack %.1f + assemble/text %.1f" % (
                  n, sum(pack) / 1e9, min(pack) / 1e9, sum(asm) / 1e9, min(asm) / 1e9,
                  max(reads * 50000 / a * 1e3 for a in pack), max(reads * 50000 / b * 1e3 for b in asm)), flush=True)


if __name__ == "__main__":
    main()
