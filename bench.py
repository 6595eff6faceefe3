#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: decomposed read-bp/s at 12 monomers x 50 kb reads.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (DP fill + traceback + record compaction + D2H of the compact
records) over one batch of synthetic reads that is already packed and resident in HBM.  At N=1 the
workload is BASELINE.json configs[1] (C2): 1000 reads x 50 kb, 12 monomers (~171 bp), default
scoring.  With N GPUs every rank owns its own 1000 reads (weak scaling; reads are independent, no
data-path collective -- SURVEY.md section 8(e)); torch.distributed is used for the barrier and the
max-over-ranks time only.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      algorithmic bytes of the fill (sum_chunks n*(sumL/4 + 6.25) + 24*rows)
                over the fill kernel's average HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline  the real reference binary (oracle/_ref/dp, kind "reference") or the C oracle
                (kind "port") timed on this box's host cores on a bounded sample of the same reads.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from stringdecomposer_amd import lib, shard, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(mn, ms, rn, rs, gpu_rows_text):
    """Reference CPU path on a bounded sample, all host cores; also re-checks parity on the sample."""
    from oracle import binding as oracle
    cores = os.cpu_count() or 1
    total_bp = sum(len(s) for s in rs)
    sample = "%d reads x %d bp of the benchmark read set (first reads of rank 0)" % (len(rs), len(rs[0]))
    # the reference's -t <threads> at the box's real width: 32, 64, ... up to every host core; the best
    # run is the baseline (its OpenMP driver works in groups of 2*t chunks with a barrier per group and
    # allocates n*(T+1) vectors per chunk, main.cpp:84-102,156-169: it stops scaling long before 256)
    widths = sorted({min(cores, w) for w in (32, 64, 128, 256)} | ({cores} if cores <= 256 else set()))
    sweep, best, parity = [], None, True
    if oracle.have_ref_dp():
        kind = "reference"
        with tempfile.TemporaryDirectory() as d:
            rf, mf = os.path.join(d, "r.fa"), os.path.join(d, "m.fa")
            synth.write_fasta(rf, rn, rs)
            synth.write_fasta(mf, mn, ms)
            for t in widths:
                t0 = time.perf_counter()
                rc, out, err = oracle.run_ref_dp(rf, mf, t)
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError("reference binary failed: " + err.decode(errors="replace")[-300:])
                parity = parity and out == gpu_rows_text
                sweep.append({"threads": t, "seconds": round(dt, 3), "bp_per_s": total_bp / dt})
    else:
        kind = "port"
        oracle.build()
        for t in widths:
            t0 = time.perf_counter()
            out = oracle.decompose(rn, rs, mn, ms, threads=t)
            dt = time.perf_counter() - t0
            parity = parity and out == gpu_rows_text
            sweep.append({"threads": t, "seconds": round(dt, 3), "bp_per_s": total_bp / dt})
    best = max(sweep, key=lambda x: x["bp_per_s"])
    return {"value": best["bp_per_s"], "unit": "bp/s", "cores": best["threads"], "kind": kind,
            "sample": sample, "seconds": best["seconds"], "host_cores_available": cores,
            "thread_sweep": sweep, "parity_on_sample": bool(parity)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1000, help="reads per GPU (C2: 1000)")
    ap.add_argument("--read-len", type=int, default=50000)
    ap.add_argument("--monomers", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kernel", choices=["auto", "generic", "fast"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=32)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started without the launcher: run it as a child process (nothing has touched the GPU yet)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    rank, local_rank, ws = shard.world()
    if args.gpus != ws and ws > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, ws))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsd_hip has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = shard.init_process_group("nccl") if ws > 1 else None
    dev = torch.device("cuda", local_rank)

    # ---- workload: synthetic reads of this rank (read i depends only on (seed, i)) -------------
    mn, ms = synth.make_monomers(args.monomers, seed=args.seed)
    lo, hi = shard.weak_range(args.reads, rank)
    rn, rs = synth.make_reads(ms, hi - lo, read_len=args.read_len, seed=args.seed, first_index=lo)
    bp_rank = sum(len(s) for s in rs)

    kernel = {"auto": 0, "generic": 1, "fast": 2}[args.kernel]
    eng = lib.Engine(ms, device=local_rank, kernel=kernel, threads=max(1, min(32, (os.cpu_count() or 1) // max(ws, 1))))
    t_load = time.perf_counter()
    n_chunks = eng.load_reads(rs)  # chunk + pack + H2D: inputs resident in HBM before timing
    t_load = time.perf_counter() - t_load
    info = eng.info()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.run(stream)
        return eng.total_rows()  # stream sync + D2H of the compact records

    for _ in range(args.warmup):
        step()
    shard.barrier(dist, local_rank if dist else None)
    torch.cuda.synchronize()
    fill_ms = trace_ms = compact_ms = 0.0
    t0 = time.perf_counter()
    rows_out = 0
    for _ in range(args.steps):
        rows_out = step()
        tm = eng.timings()
        fill_ms += tm["fill_ms"]
        trace_ms += tm["trace_ms"]
        compact_ms += tm["compact_ms"]
    torch.cuda.synchronize()
    shard.barrier(dist, local_rank if dist else None)
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dist, dt, dev)
    bp_total = shard.sum_over_ranks(dist, bp_rank, dev)
    K = max(args.steps, 1)

    # ---- roofline of the dominant kernel (fill), rank 0's launch --------------------------------
    sumL, rows = info["sum_template_len"], info["rows"]
    alg_bytes = rows * (sumL / 4.0 + 6.25) + 24.0 * rows_out
    fill_s = fill_ms / K / 1e3 / max(info["fill_launches"], 1)     # avg duration of one fill launch
    alg_per_launch = alg_bytes / max(info["fill_launches"], 1)
    achieved = alg_per_launch / fill_s / 1e9 if fill_s > 0 else 0.0
    traffic = None
    valu = None
    tf = os.path.join(ROOT, "profiles", "fill_traffic.json")
    if os.path.isfile(tf):
        try:
            with open(tf) as f:
                tj = json.load(f)
            if tj.get("workload_rows") == rows and tj.get("kernel_family") == info["family"]:
                traffic = tj.get("hbm_bytes_per_launch")
                if tj.get("SQ_INSTS_VALU_per_launch") and tj.get("cells", info["cells"]) == info["cells"]:
                    # what actually binds the fill (SURVEY 8(d) caveat): VALU issue slots, one wave64
                    # instruction per 4 cycles per SIMD, 4 SIMDs x 256 CUs; clock from GRBM_GUI_ACTIVE
                    cyc = tj["GRBM_GUI_ACTIVE_per_launch"] / 8.0
                    valu = {"wave_insts_per_launch": tj["SQ_INSTS_VALU_per_launch"],
                            "insts_per_row": tj["SQ_INSTS_VALU_per_launch"] / rows,
                            "issue_frac_profiled": tj["SQ_INSTS_VALU_per_launch"] * 4.0 / (1024.0 * cyc),
                            "source": "profiles/fill_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE)"}
        except Exception:
            traffic = None

    out = {
        "metric": "decomposed read-bp/sec (whole node) at 12 monomers x 50kb reads",
        "value": bp_total * K / dt, "unit": "bp/s", "n_gpus": ws, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": info["cells"].split("/")[0],
        "data": "synthetic",
        "config": {"workload": "%s: synthetic %d reads x %d bp per GPU, %d monomers (~171 bp) + reverse complements, "
                               "default scoring -1,-1,-1,1, part 5000 / overlap 500" % (
                                   "C2" if (args.monomers, args.reads, args.read_len) == (12, 1000, 50000) else
                                   "C4 shape (DP only)" if args.monomers == 64 else "custom", args.reads, args.read_len, args.monomers),
                   "reads_per_gpu": args.reads, "read_len": args.read_len, "n_templates": info["n_templates"],
                   "sum_template_len": sumL, "chunks_per_gpu": n_chunks, "rows_per_gpu": rows,
                   "kernel_family": info["family"], "cells_per_lane": info["cells_per_lane"],
                   "cell_arithmetic": info["cells"] + (" (packed pairs holding exact integers)" if info["cells"] == "f16" else ""),
                   "seed": args.seed, "sharding": "reads dealt to ranks in contiguous blocks, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": ("sd_fast_fill_wide" if info["cells_per_lane"] > 64 else "sd_fast_fill") if info["family"] == "fast" else "sd_generic_fill",
                     "algorithmic_bytes_per_launch": alg_per_launch,
                     "avg_launch_ms": fill_s * 1e3, "valu_issue": valu,
                     "cells_per_s": rows * sumL / max(info["fill_launches"], 1) / fill_s if fill_s > 0 else 0.0},
        "kernel_ms_per_step": {"fill": fill_ms / K, "traceback": trace_ms / K, "compact": compact_ms / K},
        "rows_out_per_gpu": rows_out, "hbm_workspace_bytes": info["workspace_bytes"],
        # not the headline: host chunking + 2-bit packing (host threads) + buffer allocation + H2D included
        "host_to_hbm_inclusive": {"load_reads_s": t_load, "bp_per_s": bp_rank / (t_load + dt / K)},
    }
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        k = max(1, min(args.cpu_sample_reads, len(rs)))
        txt = lib.decompose(rn[:k], rs[:k], mn, ms, device=local_rank, kernel=kernel)
        out["cpu_baseline"] = cpu_baseline(mn, ms, rn[:k], rs[:k], txt)
    elif rank == 0:
        out["cpu_baseline"] = None
    eng.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
