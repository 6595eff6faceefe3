#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: decomposed read-bp/s at 12 monomers x 50 kb reads.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic reads, from sequences in host memory to
rows in host memory: chunk table, 2-bit packing, H2D, DP fill, traceback, record
compaction, D2H, per-read assembly (chunk offsets + seam merge).  The same K steps are then repeated
with the batch packed and resident in HBM (one launch per kernel) for the clean per-kernel times
(`device_resident`).  At N=1 the workload is BASELINE.json configs[1] (C2): 1000 reads x 50 kb,
12 monomers (~171 bp), default scoring.  With N GPUs every rank owns its own 1000 reads (weak scaling; reads are independent, no
data-path collective -- SURVEY.md section 8(e)); torch.distributed is used for the barrier and the
max-over-ranks time only.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      the fill kernel against what binds it: VALU wave-instructions per second against the measured
                issue ceiling (bound "valu"); roofline.hbm_notional = algorithmic bytes of the fill (SURVEY 8(d):
                sum_chunks n*(sumL/4 + 6.25) + 24*rows) over the fill's average HIP-event duration against 8 TB/s
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

import stringdecomposer_amd  # noqa: E402

# this process is the benchmark's own: the runtime's queue-thread mode (see the function), chosen before any HIP call;
# the effective value is part of the result line ("amd_direct_dispatch")
AMD_DIRECT_DISPATCH = stringdecomposer_amd.prefer_queue_thread_dispatch()

from stringdecomposer_amd import lib, shard, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU issue ceiling of the fill, MEASURED (tools/ubench_issue.hip -> profiles/r03_ubench_issue.txt): every
# instruction class the fill is made of (v_pk_max/add/maximum3_f16, DPP, SDWA, v_and_or, v_readlane, v_cmp)
# sustains one wave64 instruction per SIMD per 4.07-4.20 cycles at 4 and 8 waves per SIMD -- they do not
# dual-issue (only plain v_add_f32/u32, v_mov, v_max_f16, and/or/xor reach 2.2-2.3, and those do not overlap
# with packed ops either: add_u32 + pk_maximum3 alternating = 3.9 per instruction).  1024 SIMDs, 2.4 GHz.
VALU_CYC_PER_WAVE_INST = 4.1
SHADER_CLOCK_HZ = 2.4e9
N_SIMDS = 1024
VALU_WALL_NS_PER_INST = 1.806   # measured wall time per wave-instruction per SIMD (cell mix, 4 waves per SIMD)
VALU_PEAK_GINST = N_SIMDS * SHADER_CLOCK_HZ / VALU_CYC_PER_WAVE_INST / 1e9


def cpu_baseline(mn, ms, rn, rs, gpu_rows_text):
    """Reference CPU path (`dp -t <threads>`) on a bounded sample of the benchmark reads.  The thread count
    is swept on a sub-sample (16 reads) over 4 / 8 / 16 / 32 / 64 / 128 / 256 threads -- stopping as soon as a
    wider run is clearly slower: the reference's OpenMP driver works in groups of 2*t chunks with a barrier
    per group and allocates n*(T+1) vectors per chunk (main.cpp:84-102,156-169), it does not scale to the
    256 host cores -- and the best width is then timed on the whole sample, whose output is also compared
    with the GPU rows."""
    from oracle import binding as oracle
    cores = os.cpu_count() or 1
    total_bp = sum(len(s) for s in rs)
    sample = "%d reads x %d bp of the benchmark read set (first reads of rank 0)" % (len(rs), len(rs[0]))
    widths = sorted({min(cores, w) for w in (4, 8, 16, 32, 64, 128, 256)})
    sub = min(16, len(rs))
    sub_bp = sum(len(s) for s in rs[:sub])
    have_ref = oracle.have_ref_dp()
    if not have_ref:
        oracle.build()
    with tempfile.TemporaryDirectory() as d:
        rf, rsub, mf = os.path.join(d, "r.fa"), os.path.join(d, "rsub.fa"), os.path.join(d, "m.fa")
        synth.write_fasta(rf, rn, rs)
        synth.write_fasta(rsub, rn[:sub], rs[:sub])
        synth.write_fasta(mf, mn, ms)

        def run(full, t):
            t0 = time.perf_counter()
            if have_ref:
                rc, out, err = oracle.run_ref_dp(rf if full else rsub, mf, t)
                if rc != 0:
                    raise RuntimeError("reference binary failed: " + err.decode(errors="replace")[-300:])
            else:
                out = oracle.decompose(rn if full else rn[:sub], rs if full else rs[:sub], mn, ms, threads=t)
            return out, time.perf_counter() - t0

        sweep, best = [], None
        for t in widths:
            _, dt = run(False, t)
            rate = sub_bp / dt
            sweep.append({"threads": t, "seconds": round(dt, 3), "bp_per_s": rate, "reads": sub})
            if best is None or rate > best[1]:
                best = (t, rate)
            elif rate < 0.7 * best[1]:
                break
        out, dt = run(True, best[0])
    return {"value": total_bp / dt, "unit": "bp/s", "cores": best[0], "kind": "reference" if have_ref else "port",
            "sample": sample, "seconds": round(dt, 3), "host_cores_available": cores,
            "cpu_quota_cores": cpu_quota_cores(),
            "thread_sweep": sweep, "parity_on_sample": bool(out == gpu_rows_text)}


def cpu_baseline_c4_second_best(mn, ms, rn, rs, threads=8):
    """CPU baseline of the --second-best leg, timed HERE (the GPU box's host cores) on the first reads of the benchmark
    set: the real reference binary `dp -t 8` (oracle/_ref/dp) for the raw rows, then what convert_read does with them
    (stringdecomposer/main.py:107-150): every row's segment against all 2 M templates, plain and homopolymer-compressed
    (main.py:87-92), through the reference's vendored edlib (oracle/_ref/libedlib.so; the C++ travels, the reference's
    Python does not: its pandas / Bio driver is restated as the bare loop, which can only flatter it).  Single-threaded
    like the reference's post-processing; kind "reference" when both binaries are present, else "port"."""
    from oracle import binding as oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edlib_ref
    tm = [(n_, s_.decode()) for n_, s_ in zip(mn, ms)] + [(n_ + "'", synth.revcomp_bytes(s_).decode()) for n_, s_ in zip(mn, ms)]
    htm = [edlib_ref.homo(t) for _, t in tm]
    have = oracle.have_ref_dp() and edlib_ref.have_edlib()
    if not oracle.have_ref_dp():
        oracle.build()
    with tempfile.TemporaryDirectory() as d:
        rf, mf = os.path.join(d, "r.fa"), os.path.join(d, "m.fa")
        synth.write_fasta(rf, rn, rs)
        synth.write_fasta(mf, mn, ms)
        t0 = time.perf_counter()
        if oracle.have_ref_dp():
            rc, raw, err = oracle.run_ref_dp(rf, mf, threads)
            if rc != 0:
                raise RuntimeError("reference binary failed: " + err.decode(errors="replace")[-300:])
        else:
            raw = oracle.decompose(rn, rs, mn, ms, threads=threads)
        t_dp = time.perf_counter() - t0
    seqs = {n_: s_.decode() for n_, s_ in zip(rn, rs)}
    t0 = time.perf_counter()
    n_al = 0
    for line in raw.split(b"\n")[:-1]:
        f = line.split(b"\t")
        seg = seqs[f[0].decode()][int(f[2]):int(f[3]) + 1]
        hseg = edlib_ref.homo(seg)
        for (_, t), ht in zip(tm, htm):
            edlib_ref.identity(seg, t)
            edlib_ref.identity(hseg, ht)
            n_al += 2
    t_py = time.perf_counter() - t0
    bp = sum(len(x) for x in rs)
    return {"value": bp / (t_dp + t_py), "unit": "bp/s", "cores": threads, "kind": "reference" if have else "port",
            "sample": "%d reads x %d bp of the benchmark read set: oracle/_ref/dp -t %d (%.2f s) + %d edlib alignments of "
                      "convert_read through oracle/_ref/libedlib.so, one thread (%.2f s)" % (len(rs), len(rs[0]), threads, t_dp, n_al, t_py),
            "seconds": round(t_dp + t_py, 3), "dp_seconds": round(t_dp, 3), "alignments": n_al,
            "alignments_per_s": n_al / t_py if t_py > 0 else None, "cpu_quota_cores": cpu_quota_cores()}


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited.  The GPU
    boxes of this pool show 256 logical CPUs but run under a quota of 16."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, per = float(f.read()), float(g.read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def usable_cores():
    n = os.cpu_count() or 1
    q = cpu_quota_cores()
    return n if q is None else max(1, min(n, int(q + 0.5)))


def bench_c4_second_best(args):
    """BASELINE config 4 with --second-best: the whole drop-in path, FASTA files -> final_decomposition{_raw,,_alt}.tsv
    (stringdecomposer/main.py:186-232: dp + convert_tsv with 2 T python-edlib alignments per row).  A step = one
    sd_run_files call on the C4 files (tmpfs); nothing is resident before the clock starts, text and file writes are
    inside.  The dominant kernel of this leg is the identity kernel (sd_ident_pairs: 19 M alignments per step
    against 4 M DP rows x 128 templates), so `roofline` is its VALU figure; its HBM traffic per pair is from the committed
    PMC profile."""
    import shutil
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsd_hip has no CPU fallback)")
    if args.gpus != 1:
        raise SystemExit("--config c4-second-best is a 1-GPU leg")
    n_mono, n_reads, read_len = 64, 256, 50000
    mn, ms = synth.make_monomers(n_mono, seed=args.seed)
    rn, rs = synth.make_reads(ms, n_reads, read_len=read_len, seed=args.seed)
    bp = sum(len(x) for x in rs)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="sd_bench_c4_", dir=base)
    try:
        rf, mf = os.path.join(d, "reads.fa"), os.path.join(d, "monomers.fa")
        synth.write_fasta(rf, rn, rs, width=80)
        synth.write_fasta(mf, mn, ms)
        outs = [os.path.join(d, x) for x in ("raw.tsv", "final.tsv", "alt.tsv")]
        threads = max(1, min(32, usable_cores()))
        K = max(args.steps, 1)

        def step():
            lib.run_files(rf, mf, outs[0], outs[1], outs[2], second_best=True, threads=threads, device=0)
            return lib.last_run_stats()
        def fresh_outputs():
            # a job writes NEW files; truncating the previous step's 300 MB of tmpfs pages inside open(O_TRUNC) is 20-25 ms
            # of the kernel's time that belongs to no job, so the old outputs are removed between the timed calls
            for x in outs:
                if os.path.exists(x):
                    os.unlink(x)
        for _ in range(max(args.warmup, 1)):
            fresh_outputs()
            step()
        torch.cuda.synchronize()
        dt = 0.0
        acc = {}
        for _ in range(K):
            fresh_outputs()
            t0 = time.perf_counter()
            st = step()
            torch.cuda.synchronize()
            dt += time.perf_counter() - t0
            for k, v in st.items():
                acc[k] = acc.get(k, 0.0) + v
        sizes = {os.path.basename(x): os.path.getsize(x) for x in outs}
        pairs = acc["ident_pairs"] / K
        # identity kernel: VALU instructions per pair and HBM bytes per pair from the committed PMC profile
        prof, insts_pp, bytes_pp = os.path.join(ROOT, "profiles", "ident_traffic.json"), None, None
        if os.path.isfile(prof):
            try:
                with open(prof) as f:
                    pj = json.load(f)
                insts_pp, bytes_pp = pj.get("SQ_INSTS_VALU_per_pair"), pj.get("hbm_bytes_per_pair")
            except Exception:
                pass
        ident_s = acc["ident_ms"] / K / 1e3
        # Round 6: the homopolymer-compressed pairs get their distance alone first; only those whose identity bounds reach a
        # record's two best are aligned in full.  The committed per-pair figures (profiles/ident_traffic.json) are those of a
        # FULL alignment, plain and compressed; profiles/r06_ident_pmc.json holds the instructions and bytes per step of
        # every kernel of the pruned form, taken on this workload.
        homo_pairs, homo_full = acc.get("homo_pairs", 0.0) / K, acc.get("homo_full_pairs", 0.0) / K
        pruned = None
        pj6 = os.path.join(ROOT, "profiles", "r06_ident_pmc.json")
        if homo_pairs > 0 and os.path.isfile(pj6):
            with open(pj6) as f:
                p6 = json.load(f)
            if p6.get("pairs_per_step") == pairs:
                tot = sum(v["SQ_INSTS_VALU_per_step"] for v in p6["kernels"].values())
                hb = sum(v.get("hbm_bytes_per_step", 0.0) for v in p6["kernels"].values())
                pruned = {"wave_insts_per_step": tot, "issue_frac": tot / ident_s / 1e9 / VALU_PEAK_GINST if ident_s > 0 else None,
                          "hbm_bytes_per_step": hb, "hbm_bytes_per_pair": hb / pairs, "traffic_over_algorithmic": hb / pairs / (171 / 4.0 + 4.0),
                          "per_kernel": p6["kernels"], "source": "profiles/r06_ident_pmc.json (rocprofv3 --pmc, separate passes, this workload)"}
        roofline = {"kernel": "sd_ident_pairs<3> (plain: every pair in full) + sd_ident_dist<3> / sd_ident_select / sd_ident_pairs<3> on the "
                              "selected pairs (homopolymer-compressed), per identity slice",
                    "pairs_per_step": pairs, "ident_ms_per_step": ident_s * 1e3,
                    "homopolymer_pairs_per_step": homo_pairs, "homopolymer_pairs_aligned_in_full": homo_full,
                    "homopolymer_pairs_skipped_frac": None if homo_pairs <= 0 else 1.0 - homo_full / homo_pairs,
                    "pruned_form": pruned,
                    "pairs_per_s_in_kernel": pairs / ident_s if ident_s > 0 else None,
                    "algorithmic_bytes_per_pair": 171 / 4.0 + 4.0,
                    "traffic_bytes_per_pair": bytes_pp,
                    "traffic_source": None if bytes_pp is None else "committed profile profiles/ident_traffic.json (rocprofv3 --pmc "
                    "FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE doubled for 16-B-per-lane reads)"}
        if pruned is not None and pruned["issue_frac"] is not None:
            roofline.update({"bound": "valu", "achieved": pruned["wave_insts_per_step"] / ident_s / 1e9, "peak": VALU_PEAK_GINST,
                             "unit": "G wave-inst/s", "frac": pruned["issue_frac"], "traffic_bytes_per_pair": pruned["hbm_bytes_per_pair"],
                             "insts_source": pruned["source"]})
        elif insts_pp and ident_s > 0 and homo_pairs <= 0:
            ach = insts_pp * pairs / ident_s / 1e9
            roofline.update({"bound": "valu", "achieved": ach, "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s",
                             "frac": ach / VALU_PEAK_GINST, "wave_insts_per_pair": insts_pp,
                             "insts_source": "committed profile profiles/ident_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU)"})
        else:
            roofline.update({"bound": "valu", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s", "frac": None})
        # the leg's CPU baseline, timed on THIS box: reference dp + reference edlib on the first two reads; the unmodified
        # reference command line (pandas / Bio driver included) was timed once in the build container, where it can run
        cpu = None if args.no_cpu_baseline else cpu_baseline_c4_second_best(mn, ms, rn[:2], rs[:2])
        cj = os.path.join(ROOT, "profiles", "ref_cli_c4_second_best.json")
        if os.path.isfile(cj) and cpu is not None:
            with open(cj) as f:
                cpu["unmodified_reference_cli_in_build_container"] = json.load(f)
        out = {"metric": "decomposed read-bp/sec at 64 monomers x 50kb reads, --second-best (BASELINE config 4), FASTA files -> "
                         "final_decomposition{_raw,,_alt}.tsv", "value": bp * K / dt, "unit": "bp/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f16 DP cells holding exact integers (the 128-template wide fill), u32 bit vectors (identities)",
               "data": "synthetic",
               "timed_region": "one sd_run_files call per step: FASTA files on tmpfs -> index + alphabet check -> chunk -> pack -> "
                               "H2D -> fill -> traceback -> compaction -> identities in-stream -> D2H -> per-read assembly -> raw / "
                               "final / _alt text -> three NEW files on tmpfs (the previous step's outputs are removed between the timed calls); "
                               "engines kept from the previous job with the same parameters (sd_release_cache drops them)",
               "config": {"workload": "C4: synthetic %d reads x %d bp, %d monomers (~171 bp) + reverse complements, default scoring, "
                                      "--second-best (2 x %d alignments per row)" % (n_reads, read_len, n_mono, 2 * n_mono),
                          "host_threads": threads, "seed": args.seed, "output_bytes": sizes},
               "roofline": roofline,
               "kernel_ms_per_step": {k: acc[k] / K for k in ("fill_ms", "trace_ms", "compact_ms", "ident_ms")},
               "host_ms_per_step": {k: acc[k] / K for k in ("pack_ms", "wait_ms", "raw_text_ms", "post_ms", "io_ms",
                                                            "text_identity_ms", "final_text_ms", "total_ms", "setup_ms", "assemble_ms")},
               "cpu_baseline": cpu}
        print(json.dumps(out), file=_claimed_stdout(), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def bench_strong(args):
    """Strong scaling: ONE job of BASELINE config 3 (a set of reads, 12 monomers, default scoring) or config 5 (one
    long sequence, scoring -2,-3,-4,2) split over the ranks as the multi-process command line splits it
    (shard.strong_share: contiguous blocks of reads / of the global chunk table; no data-path collective).  A step =
    the whole job once; every rank runs its share, the time is the MAX over ranks, value = job bp / that."""
    import torch
    rank, local_rank, ws = shard.world()
    if args.gpus != ws and ws > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, ws))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsd_hip has no CPU fallback)")
    share_gpu = bool(os.environ.get("SD_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = shard.init_process_group("gloo" if share_gpu else "nccl") if ws > 1 else None
    dev = torch.device("cpu") if share_gpu else torch.device("cuda", local_rank)
    bar_dev = None if (share_gpu or not dist) else local_rank
    threads = max(1, min(32, usable_cores() // max(ws, 1)))
    K = max(args.steps, 1)
    mn, ms = synth.make_monomers(12, seed=args.seed)
    if args.config == "c3":
        kind, lo, hi = shard.strong_share("c3", rank, ws, n_reads=args.reads_total)
        rn, rs = synth.make_reads(ms, hi - lo, read_len=args.read_len, seed=args.seed, first_index=lo)
        my_bp = sum(len(x) for x in rs)
        job_bp = shard.sum_over_ranks(dist, my_bp, dev)
        readset = lib.ReadSet(rs)
        st = lib.Stream(ms, device=local_rank, threads=threads, pipe_mode=args.pipe_mode)

        def step():
            st.submit(readset)
            return st.collect()
        workload = "C3: ONE job of %d synthetic reads x %d bp, 12 monomers, default scoring, part 5000 / overlap 500; " \
                   "this rank: reads [%d, %d)" % (args.reads_total, args.read_len, lo, hi)
        scoring = (-1, -1, -1, 1)
    else:
        scoring = (-2, -3, -4, 2)
        kind, lo, hi, n_chunks = shard.strong_share("c5", rank, ws, seq_len=args.seq_len)
        _, rs = synth.make_reads(ms, 1, read_len=args.seq_len, seed=args.seed)
        job_bp = args.seq_len
        st = None

        last = {}

        def step():
            recs, off = lib.decompose_chunk_range(rs, ms, lo, hi, scoring=scoring, device=local_rank, threads=threads)
            last["recs"], last["off"] = recs, off
            return len(recs)
        workload = "C5: ONE sequence of %d bp, 12 monomers, scoring -2,-3,-4,2, part 5000 / overlap 500 (%d chunks); this " \
                   "rank: chunks [%d, %d)" % (args.seq_len, n_chunks, lo, hi)
    for _ in range(max(args.warmup, 1)):
        step()
    a = st.stats() if st else None
    shard.barrier(dist, bar_dev)
    torch.cuda.synchronize()
    c0 = os.times()
    t0 = time.perf_counter()
    out_rows = 0
    for _ in range(K):
        out_rows = step()
    torch.cuda.synchronize()
    my_dt = time.perf_counter() - t0
    c1 = os.times()
    shard.barrier(dist, bar_dev)
    dt = shard.max_over_ranks(dist, my_dt, dev)
    rate = shard.job_rate(dist, job_bp * K, my_dt, dev)
    # What follows the DP in the real multi-process job and is NOT inside the timed steps: records -> raw TSV (chunk offsets,
    # seam merge, SaveBatch text: main.cpp:104-117, 272-302).  Since round 5 every rank does that for its own chunk range
    # (shard._assemble_by_ranks: two exchanges of a few hundred bytes, csrc/sd_seam.hpp) instead of rank 0 for everybody;
    # timed once and reported beside the DP.  One process: the whole-job assembly a single GPU pays, and the same job cut
    # into eight ranges assembled one after the other with the threads a rank of eight gets (what each of eight GPUs' hosts
    # would pay: the slowest range counts).
    serial = None
    if st is None:
        import numpy as np
        shard.barrier(dist, bar_dev)
        if dist is not None and ws > 1:
            import tempfile
            stats = {}
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            raw = os.path.join(shm, "sd_bench_c5_%s.tsv" % os.environ.get("MASTER_PORT", "0"))   # the same name on every rank
            s0 = time.perf_counter()
            done = shard._assemble_by_ranks(dist, rank, ws, lambda: lib.RangeAssembler.from_lists(
                ["seq"], [args.seq_len], mn, lo, hi, last["recs"], last["off"], scoring=scoring, threads=threads),
                raw_tsv_out=raw, stats=stats)
            s1 = time.perf_counter()
            worst = shard.max_over_ranks(dist, s1 - s0, dev)
            if done is not None:
                nbytes = os.path.getsize(raw) if rank == 0 else None
                serial = {"mode": "every rank assembles its own chunk range and writes it into the file at its offset",
                          "wall_ms_max_over_ranks": worst * 1e3, "rank0": stats, "raw_tsv_bytes": nbytes,
                          "note": "once, after the timed steps: records -> raw TSV file (%s); not part of ms_per_step" % shm}
            shard.barrier(dist, bar_dev)
            if rank == 0 and os.path.exists(raw):
                os.remove(raw)
            if done is not None:
                pass
            else:
                serial = {"mode": "not shareable (a share is empty or a crossing piece is shorter than 32 records)"}
        else:
            s1 = time.perf_counter()
            tsv = lib.assemble_tsv(["seq"], [args.seq_len], mn, last["recs"], last["off"], scoring=scoring, threads=threads)
            s2 = time.perf_counter()
            serial = {"mode": "one process", "assemble_raw_tsv_ms": (s2 - s1) * 1e3, "raw_tsv_rows": tsv.count(b"\n"),
                      "raw_tsv_bytes": len(tsv),
                      "note": "once, after the timed steps: this rank's records -> raw TSV text in host memory; not part of ms_per_step"}
            if kind == "chunks" and lo == 0 and hi - lo >= 64:
                t8 = max(1, threads // 8)
                asm, per = [], []
                for g in range(8):
                    glo, ghi = shard.block_range(hi, g, 8)
                    g0 = time.perf_counter()
                    asm.append(lib.RangeAssembler.from_lists(["seq"], [args.seq_len], mn, glo, ghi, last["recs"][last["off"][glo]:last["off"][ghi]],
                                                             last["off"][glo:ghi + 1] - last["off"][glo], scoring=scoring, threads=t8))
                    per.append(time.perf_counter() - g0)
                edges = [x.edge for x in asm]
                texts = []
                for g, x in enumerate(asm):
                    g0 = time.perf_counter()
                    x.text(edges, g)
                    per[g] += time.perf_counter() - g0
                    texts.append(x.bytes())
                st8 = [x.stats() for x in asm]
                for x in asm:
                    x.close()
                serial["eight_ranges_one_after_the_other"] = {
                    "threads_per_range": t8, "ms_per_range": [round(v * 1e3, 2) for v in per], "slowest_ms": max(per) * 1e3,
                    "after_the_exchange_ms": [round(q["text_ms"], 3) for q in st8],
                    "rows_printed_after_the_exchange": [q["rows_printed_after_exchange"] for q in st8],
                    "edge_bytes_per_rank": len(edges[0]), "same_bytes_as_one_process": b"".join(texts) == tsv}
    kern = None
    if st:
        b = st.stats()
        kern = {"fill": (b["fill_ms"] - a["fill_ms"]) / K, "traceback": (b["trace_ms"] - a["trace_ms"]) / K,
                "batches_per_step": (b["batches"] - a["batches"]) / K}
        st.close()
    out = {"metric": "decomposed read-bp/sec (whole node), one job split over the GPUs", "value": rate, "unit": "bp/s",
           "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": lib.plan_info(ms, scoring=scoring)["cells"].split("/")[0], "data": "synthetic",
           "config": {"workload": workload, "job_bp": job_bp, "share": [kind, lo, hi], "scoring": list(scoring),
                      "host_threads": threads, "seed": args.seed,
                      "sharding": "contiguous blocks of %s, no collective (shard.strong_share)" % kind},
           "roofline": None, "cpu_baseline": None, "serial_after_dp": serial,
           "rank0": {"seconds": my_dt, "rows_out_last_step": out_rows, "kernel_ms_per_step": kern,
                     "process_cpu_ms_per_step": ((c1.user - c0.user) + (c1.system - c0.system)) * 1e3 / K},
           "note": "strong-scaling companion of the headline line (which is weak scaling on C2, roofline and cpu_baseline "
                   "there); time = MAX over ranks of the K steps"}
    if rank == 0:
        print(json.dumps(out), file=_claimed_stdout(), flush=True)
    if dist is not None:
        dist.destroy_process_group()


_REAL_STDOUT = None


# Steps outstanding before the oldest one's rows are collected: with 2 the library's three engines all have a batch (the
# next fill is already queued when a fill ends; csrc/sd_engine.hip at Pipeline::NS).  SD_BENCH_DEPTH=1 + SD_PIPE_SLOTS=2 is
# the round-1-4 pipeline (A/B).
BENCH_DEPTH = max(1, int(os.environ.get("SD_BENCH_DEPTH", "2")))

# the sources that decide the instruction streams of the profiled fills / tracebacks (the list of four missed the
# multi-wave fill used by --monomers 150 / 400)
KERNEL_SOURCES = ("sd_fast_fill.hpp", "sd_fast_dev.hpp", "sd_fast_trace2.hip", "sd_fast_wide_fill.hpp", "sd_fast_wn_fill.hpp")


def kernel_source_hashes():
    """sha256 of the sources that decide the instruction stream of the fill / traceback kernels the committed PMC
    profile (profiles/fill_traffic.json) was taken on."""
    import hashlib
    out = {}
    for fn in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "stringdecomposer_amd", "csrc", fn), "rb") as f:
            out[fn] = hashlib.sha256(f.read()).hexdigest()[:16]
    return out


def _claimed_stdout():
    """The process's original stdout; everything else that writes to fd 1 (gloo / RCCL / runtime chatter) goes to stderr,
    so that rank 0 prints exactly ONE line there."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's SCALE runs): every rank owns its own C2 read set; strong: ONE job -- "
                         "--config c3 (a set of reads) or c5 (one 200-Mb sequence) -- split over the ranks exactly as the "
                         "multi-process command line splits it (shard.strong_share)")
    ap.add_argument("--reads-total", type=int, default=100000, help="--scaling strong --config c3: reads of the whole job")
    ap.add_argument("--seq-len", type=int, default=200000000, help="--scaling strong --config c5: length of the sequence")
    ap.add_argument("--config", choices=["c2", "c3", "c5", "c4-second-best"], default="c2",
                    help="c2 (default, the headline): BASELINE configs[1]; c4-second-best: BASELINE config 4 (64 monomers x "
                         "256 reads x 50 kb) through the whole command-line path with --second-best, FASTA files -> the three "
                         "TSV files (1 GPU), roofline of the identity kernel")
    ap.add_argument("--reads", type=int, default=1000, help="reads per GPU (C2: 1000)")
    ap.add_argument("--read-len", type=int, default=50000)
    ap.add_argument("--monomers", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kernel", choices=["auto", "generic", "fast"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-port-baseline", action="store_true",
                    help="accept the C restatement (kind \"port\") as cpu_baseline when oracle/_ref/dp is absent; by default "
                         "the run then exits 3 after printing its line")
    ap.add_argument("--cpu-sample-reads", type=int, default=200,
                    help="reads of the benchmark set the reference CPU path is timed on and compared with (200 x 50 kb = 10 Mbp: "
                         "~25 s of `dp -t 8`)")
    ap.add_argument("--sustain-seconds", type=float, default=8.0,
                    help="N = 1: after the timed region, the same steps for about this long (reported as `sustained`; 0 = skip)")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the timed region (no second pipe mode, no device-resident repeat): for profilers, so "
                         "that per-kernel averages are those of the timed launches")
    ap.add_argument("--ed-thr", type=int, default=-1, help="--ed_thr prefilter (profiling the filter kernels; not the headline)")
    ap.add_argument("--pipe-mode", type=int, choices=[0, 1, 2], default=2,
                    help="kernel streams of the timed region: 2 = the library default (traceback and the next fill move into "
                         "the current fill's drain; kernel event spans overlap), 0 = every kernel in order on one stream "
                         "(clean per-kernel spans), 1 = traceback overlap only; the other of {2, 0} is timed as well and "
                         "reported as other_pipe_mode")
    ap.add_argument("--resident-stream", choices=["null", "new"], default="null")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host threads of a rank (default: its share of the usable CPUs, at most 32); e.g. 2 = what a rank gets on a "
                         "16-CPU box shared by 8 ranks")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="device batches per step (three batches are in flight, across steps; 1 is fastest: the "
                         "persistent kernels want >= 2 chunks per resident wave)")
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
    _claimed_stdout()   # from here on only the result line goes to stdout

    if args.config == "c4-second-best":
        return bench_c4_second_best(args)
    if args.scaling == "strong" or args.config in ("c3", "c5"):
        if args.config not in ("c3", "c5"):
            raise SystemExit("--scaling strong needs --config c3 or c5")
        return bench_strong(args)

    import torch
    rank, local_rank, ws = shard.world()
    if args.gpus != ws and ws > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, ws))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsd_hip has no CPU fallback)")
    share = bool(os.environ.get("SD_BENCH_SHARE_GPU"))   # developer smoke test of the N > 1 path on a 1-GPU box:
    if share:                                             # every rank on GPU 0, host-side (gloo) reductions
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = shard.init_process_group("gloo" if share else "nccl") if ws > 1 else None
    dev = torch.device("cpu") if share else torch.device("cuda", local_rank)
    bar_dev = None if (share or not dist) else local_rank

    # ---- workload: synthetic reads of this rank (read i depends only on (seed, i)) -------------
    mn, ms = synth.make_monomers(args.monomers, seed=args.seed)
    lo, hi = shard.weak_range(args.reads, rank)
    rn, rs = synth.make_reads(ms, hi - lo, read_len=args.read_len, seed=args.seed, first_index=lo)
    bp_rank = sum(len(s) for s in rs)

    kernel = {"auto": 0, "generic": 1, "fast": 2}[args.kernel]
    # host threads of this rank: its share of the CPUs the container may actually use (threads beyond a cgroup
    # quota only get the whole process group throttled)
    threads = args.host_threads if args.host_threads > 0 else max(1, min(32, usable_cores() // max(ws, 1)))
    K = max(args.steps, 1)

    # ---- the timed region: SURVEY.md 8(d) -- sequences in host memory -> chunk -> 2-bit pack -> H2D ->
    # fill -> traceback -> compaction -> D2H -> per-read assembly (chunk offsets, seam merge) -> rows in
    # host memory.  Every step submits the rank's whole read set; the stream cuts it into sub-batches
    # and keeps three of them in flight (fills alternate between two HIP streams), across step boundaries (a long job is a
    # stream of such batches), so packing / upload / assembly run under the kernels of the neighbours.
    readset = lib.ReadSet(rs)

    def timed_steps(pipe_mode, steps, warmup, bracket):
        """K pipelined steps of the stream in the given kernel-stream mode; returns (seconds, rows of the
        last step, stats delta, info)."""
        # pipe_mode (sd_params.reserved[0]): 0 = all kernels of all
        # batches in order on one HIP stream (per-kernel event spans are clean); 2 = the library default:
        # traceback + compaction on a second, lower-priority stream and consecutive fills on two streams, the
        # next fill moving in as the current one drains (5-7 % faster, but the kernels' event spans overlap)
        st = lib.Stream(ms, sub_batches=args.sub_batches, device=local_rank, kernel=kernel, threads=threads,
                        ed_thr=args.ed_thr, pipe_mode=pipe_mode)
        inf = st.info()
        for _ in range(max(warmup, 3)):   # at least one step per pipeline slot: buffers of all engines exist
            st.submit(readset)
            st.collect()
        a = st.stats()
        if bracket:
            shard.barrier(dist, bar_dev)
        torch.cuda.synchronize()
        c0 = os.times()
        t0 = time.perf_counter()
        nrows = 0
        depth = BENCH_DEPTH   # steps kept outstanding before the oldest is collected
        out = 0
        step_t = [] if os.environ.get("SD_BENCH_STEP_TIMES") else None   # developer: wall time of every step on stderr
        for k in range(steps):      # (the order of calls of lib.Stream.imap)
            if step_t is not None:
                step_t.append(time.perf_counter())
            st.submit(readset)
            out += 1
            if out > depth:
                nrows = st.collect()     # rows of an earlier step are in host memory
                out -= 1
        if step_t:
            step_t.append(time.perf_counter())
            print("[bench] step ms:", " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(step_t, step_t[1:])), file=sys.stderr)
        while out > 0:
            nrows = st.collect()
            out -= 1
        torch.cuda.synchronize()
        if bracket:
            shard.barrier(dist, bar_dev)
        sec = time.perf_counter() - t0
        c1 = os.times()
        b = st.stats()
        st.close()
        dd = {k: b[k] - a[k] for k in b}
        # CPU time of this process (all host threads, user + system) over the timed region: what one rank asks
        # of the host per step -- the N-GPU node has to supply N times that within one step time
        dd["host_cpu_ms_per_step"] = ((c1.user - c0.user) + (c1.system - c0.system)) * 1e3 / max(steps, 1)
        return sec, nrows, dd, inf

    dt, rows_out, d, info = timed_steps(args.pipe_mode, args.steps, args.warmup, True)
    dt = shard.max_over_ranks(dist, dt, dev)
    bp_total = shard.sum_over_ranks(dist, bp_rank, dev)
    n_chunks = lib.chunk_table_size([len(x) for x in rs])
    other = None
    if ws == 1 and not args.timed_only:   # the other kernel-stream mode, same K steps (secondary figure)
        om = 2 if args.pipe_mode == 0 else 0
        odt, _, od, _ = timed_steps(om, args.steps, min(args.warmup, 1), False)
        other = {"pipe_mode": om, "bp_per_s": bp_rank * K / odt, "ms_per_step": odt / K * 1e3,
                 "kernel_event_ms_per_step": {"fill": od["fill_ms"] / K, "traceback": od["trace_ms"] / K}}

    # ---- the headline's steps again, for seconds instead of K steps (secondary figure: does the rate of the 0.3-s timed
    # region hold at the clocks and temperatures of a long job?  It also keeps the GPU busy for long enough that a sampler
    # beside the process -- the driver's gpu_busy -- sees it: the other legs of this script are mostly the CPU baseline) ----
    sustained = None
    if ws == 1 and not args.timed_only and args.sustain_seconds > 0:
        S = int(min(5000, max(K, round(args.sustain_seconds / max(dt / K, 1e-6)))))
        sdt, srows, _, _ = timed_steps(args.pipe_mode, S, 1, False)
        sustained = {"steps": S, "seconds": round(sdt, 3), "ms_per_step": sdt / S * 1e3, "bp_per_s": bp_rank * S / sdt,
                     "vs_timed_region": (dt / K) / (sdt / S), "same_rows_as_headline": srows == rows_out}

    # ---- the same K steps with the batch already packed and resident in HBM, one launch per kernel, no
    # overlap: clean per-kernel HIP-event times (device_resident; NOT the headline) ------------------
    eng = lib.Engine(ms, device=local_rank, kernel=kernel, threads=threads, ed_thr=args.ed_thr)
    t_load = time.perf_counter()
    eng.load_reads(rs)
    t_load = time.perf_counter() - t_load
    tstream = torch.cuda.current_stream().cuda_stream
    if args.resident_stream == "new":   # developer A/B: a created stream instead of the null stream
        _keep = torch.cuda.Stream()
        tstream = _keep.cuda_stream
    res_steps = 0 if args.timed_only else args.steps
    for _ in range(min(args.warmup, 1) if res_steps else 0):
        eng.run(tstream)
        eng.total_rows()
    r_fill = r_trace = r_cmp = 0.0
    torch.cuda.synchronize()
    tr0 = time.perf_counter()
    for _ in range(res_steps):
        eng.run(tstream)
        eng.total_rows()
        tm = eng.timings()
        r_fill += tm["fill_ms"]
        r_trace += tm["trace_ms"]
        r_cmp += tm["compact_ms"]
    torch.cuda.synchronize()
    dtr = time.perf_counter() - tr0
    einfo = eng.info()
    res_recs = eng.total_rows() if res_steps else None   # records of the resident batch (before the per-read seam merge)
    eng.close()

    # ---- the same launch with the other cell formats of the layout: saturating int16 (SD_FLAG_NO_F16; roofline.int16_cells,
    # reported since round 4) and, where the headline runs the biased-u16 cells of round 6, the exact-integer fp16 cells
    # of rounds 1-5 (SD_FLAG_NO_U16; roofline.f16_cells) -- same rows required of both ----
    def cell_format_leg(flags):
        e16 = lib.Engine(ms, device=local_rank, kernel=kernel, threads=threads, flags=flags)
        e16.load_reads(rs)
        i16 = e16.info()
        n16 = max(2, min(res_steps, 5))
        e16.run(tstream)
        rows16 = e16.total_rows()
        f16ms = t16ms = 0.0
        for _ in range(n16):
            e16.run(tstream)
            e16.total_rows()
            tm = e16.timings()
            f16ms += tm["fill_ms"]
            t16ms += tm["trace_ms"]
        torch.cuda.synchronize()
        e16.close()
        return {"cells": i16["cells"], "fill_ms": f16ms / n16, "traceback_ms": t16ms / n16, "steps": n16,
                "rows_out": rows16, "launched": "alone, batch resident in HBM (as isolated_*)"}
    int16_leg = f16_leg = None
    if ws == 1 and not args.timed_only and info["cells"].split("/")[0] in ("f16", "u16") and args.ed_thr < 0:
        int16_leg = cell_format_leg(lib.FLAG_NO_F16)
        if info["cells"] == "u16":
            f16_leg = cell_format_leg(lib.FLAG_NO_U16)
            if f16_leg["cells"] != "f16":
                f16_leg = None

    # ---- roofline of the dominant kernel (fill) over the timed region, rank 0 ---------------------
    sumL, rows = info["sum_template_len"], einfo["rows"]
    alg_bytes = rows * (sumL / 4.0 + 6.25) + 24.0 * rows_out       # SURVEY.md 8(d), per step
    launches = max(d["fill_launches"], 1.0)
    fill_s = d["fill_ms"] / 1e3 / launches                          # avg duration of one fill launch
    alg_per_launch = alg_bytes * K / launches
    achieved = alg_per_launch / fill_s / 1e9 if fill_s > 0 else 0.0
    res_fill_s = r_fill / max(res_steps, 1) / 1e3 / max(einfo["fill_launches"], 1)
    res_achieved = alg_bytes / max(einfo["fill_launches"], 1) / res_fill_s / 1e9 if res_fill_s > 0 else 0.0
    traffic = None
    valu = None
    profile_problems = []
    tf = os.path.join(ROOT, "profiles", "fill_traffic.json")
    if os.path.isfile(tf):
        try:
            with open(tf) as f:
                tj = json.load(f)
            # The committed counters belong to ONE build of the kernels: instructions per row and bytes per launch change
            # with the row loop.  The profile carries the hashes of the kernel sources it was taken on
            # (tools/collect_traffic.py); a different source -> no instruction count, no traffic, and a note.
            want = tj.get("kernel_sources_sha256")
            have = kernel_source_hashes()
            if want != have:
                stale = sorted(k for k in have if (want or {}).get(k) != have[k])
                profile_problems.append("profiles/fill_traffic.json was taken on other kernel sources (%s): valu_issue and traffic "
                                        "withheld; rerun tools/collect_traffic.py on the GPU box" % ", ".join(stale))
                raise LookupError("stale profile")
            # the C2 entry at the top level, further workloads (the 64-monomer wide kernel) under "other_workloads"
            for tj in [tj] + list(tj.get("other_workloads", [])):
                same_kernel = (tj.get("kernel_family") == info["family"] and
                               tj.get("cells", info["cells"]).split("/")[0] == info["cells"].split("/")[0] and
                               tj.get("cells_per_lane", 35) == info["cells_per_lane"] and tj.get("sum_template_len", 4096) == sumL)
                if same_kernel and tj.get("workload_rows") == rows and tj.get("hbm_bytes_per_launch"):
                    traffic = tj.get("hbm_bytes_per_launch")
                if same_kernel and tj.get("SQ_INSTS_VALU_per_launch") and tj.get("workload_rows"):
                    # what actually binds the fill (SURVEY 8(d) caveat): VALU issue slots.  The instruction count of a row
                    # is a property of (kernel instance, template set) -- every row of every chunk runs the same slot loop
                    # and tail -- counted once by rocprofv3 --pmc SQ_INSTS_VALU on this workload (committed profile) and
                    # scaled by this run's rows; the durations are measured in this run.
                    per_row = tj["SQ_INSTS_VALU_per_launch"] / tj["workload_rows"]
                    valu = {"wave_insts_per_launch": per_row * rows / max(d["fill_launches"] / K, 1.0),
                            "insts_per_row": per_row,
                            "cycles_per_wave_inst_per_simd": VALU_CYC_PER_WAVE_INST,
                            "ceiling_source": "profiles/r03_ubench_issue.txt (tools/ubench_issue.hip, this GPU model)",
                            "issue_frac_profiled": None if not tj.get("GRBM_GUI_ACTIVE_per_launch") else
                            tj["SQ_INSTS_VALU_per_launch"] * VALU_CYC_PER_WAVE_INST / (N_SIMDS * tj["GRBM_GUI_ACTIVE_per_launch"] / 8.0),
                            "source": "instructions per row: committed profile profiles/fill_traffic.json (rocprofv3 --pmc "
                                      "SQ_INSTS_VALU, device-resident single launch of %s) x the rows of this "
                                      "run; durations: this run" % tj.get("kernel_instance", "the fill")}
                    break
        except Exception:
            traffic = None
    kname = ("sd_fast_fill_wide" if info["cells_per_lane"] > 64 else "sd_fast_fill") if info["family"] == "fast" else "sd_generic_fill"

    # contract figure (hbm_notional): SURVEY 8(d) algorithmic bytes of a fill launch / its HIP-event duration, launches
    # of the timed region (in the default stream mode a fill's span contains the neighbouring batch's traceback and
    # the drain hand-over; isolated_* = the same kernel launched alone, from the device_resident steps).  The fill is
    # NOT HBM bound: it moves 0.28x the algorithmic bytes (pointers are recomputed by the traceback, not stored) and
    # is limited by VALU issue slots, so the primary figure is wave-instructions per second against the measured
    # issue ceiling of the chip; the notional HBM fraction north_star asks for is carried next to it.
    hbm_notional = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": alg_per_launch,
                    "isolated_frac": res_achieved / HBM_PEAK_GBS if res_steps else None,
                    "path_frac": alg_bytes * K / dt / 1e9 / HBM_PEAK_GBS if ws == 1 else None,
                    "note": "algorithmic bytes (2 bit per cell + per-row words + records) / fill duration against 8 TB/s"}
    if valu is not None and fill_s > 0:
        v_ach = valu["wave_insts_per_launch"] / fill_s / 1e9
        v_iso = valu["wave_insts_per_launch"] / res_fill_s / 1e9 if res_steps and res_fill_s > 0 else None
        roofline = {"bound": "valu", "achieved": v_ach, "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s",
                    "peak_note": "1024 SIMDs x 2.4 GHz / 4.1 cycles per wave64 instruction: the issue rate MEASURED on this GPU for "
                                 "packed / DPP / SDWA / VOP3 instructions (profiles/r03_ubench_issue.txt); the micro-architecture "
                                 "guide's 2-cycle SIMD-32 figure would double the peak and halve every fraction quoted against it",
                    "frac": v_ach / VALU_PEAK_GINST,
                    "stream_span_frac": v_ach / VALU_PEAK_GINST,
                    "isolated_frac": None if v_iso is None else v_iso / VALU_PEAK_GINST}
        # `peak` prices the 4.1 cycles at the 2.4-GHz peak clock; under this load the chip holds 2.27-2.33 GHz (the
        # micro-benchmark's wall time per instruction, profiles/r03_ubench_issue.txt: 1.806 ns per SIMD for the fill's
        # instruction mix at 4 waves per SIMD): what the machine delivered there, for reference
        # In the default stream mode the launches of consecutive batches overlap (a fill starts in the previous fill's
        # drain and shares the machine with the previous traceback), so `frac` -- instructions over the fill's own
        # HIP-event span, as the contract asks -- falls when the overlap grows even if the step gets faster.  The figure
        # that cannot be gamed by overlap: the vector instructions of BOTH kernels of a step over the step time.
        roofline["frac_note"] = ("frac / achieved = work over time: the vector wave-instructions of the fill AND the traceback of the "
                                 "timed steps over the timed region (step_frac; without a traceback count: the fill alone, "
                                 "isolated_frac); stream_span_frac = fill instructions / the fill's in-stream HIP-event span -- "
                                 "spans of consecutive batches overlap in stream mode 2, so it measures occupancy of the stream, "
                                 "not work; isolated_frac = the fill launched alone on the resident batch")
        if tj.get("traceback_SQ_INSTS_VALU_per_launch") and ws == 1 and args.sub_batches == 1:
            both = valu["wave_insts_per_launch"] + tj["traceback_SQ_INSTS_VALU_per_launch"] * rows / tj["workload_rows"]
            roofline["step_frac"] = both * K / dt / 1e9 / VALU_PEAK_GINST
            valu["traceback_wave_insts_per_launch"] = tj["traceback_SQ_INSTS_VALU_per_launch"] * rows / tj["workload_rows"]
            roofline["achieved"] = both * K / dt / 1e9
            roofline["frac"] = roofline["step_frac"]
        elif v_iso is not None:
            roofline["achieved"] = v_iso
            roofline["frac"] = v_iso / VALU_PEAK_GINST
        valu["wall_ceiling_ginst"] = N_SIMDS / VALU_WALL_NS_PER_INST
        valu["isolated_frac_of_wall_ceiling"] = None if v_iso is None else v_iso / valu["wall_ceiling_ginst"]
    else:   # no instruction count for this workload / binary: only the notional figure
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "isolated_frac": res_achieved / HBM_PEAK_GBS if res_steps else None}
    roofline.update({
        "traffic": traffic,
        # the counters' HBM bytes of a launch over the launch's own time (alone): what the fill really asks of the 8 TB/s
        "hbm_measured_frac": None if (traffic is None or not res_steps or res_fill_s <= 0) else traffic / res_fill_s / 1e9 / HBM_PEAK_GBS,
        "traffic_source": None if traffic is None else
        "committed profile: profiles/fill_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
        "device-resident single launch), not measured in this run",
        "kernel": kname, "avg_launch_ms": fill_s * 1e3, "launches_timed": launches,
        "isolated_avg_launch_ms": res_fill_s * 1e3 if res_steps else None,
        "valu_issue": valu, "hbm_notional": hbm_notional,
        "binding_resource": "VALU issue slots: packed-f16 / DPP / SDWA wave64 instructions issue once per 4.1 cycles per "
                            "SIMD on gfx950 (measured), the fill is %s of them per row" % (
                                "%.0f" % valu["insts_per_row"] if valu else "116 (C2) / 584 (128 templates)"),
        "cells_per_s": rows * sumL * K / launches / fill_s if fill_s > 0 else 0.0})
    if int16_leg is not None:
        # same algorithmic bytes per launch, the integer-cell kernel's duration launched alone
        int16_leg["hbm_notional_frac"] = alg_bytes / max(einfo["fill_launches"], 1) / (int16_leg["fill_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS
        int16_leg["same_rows_as_headline"] = bool(int16_leg.pop("rows_out") == res_recs)   # records of the same resident batch
        roofline["int16_cells"] = int16_leg
    if f16_leg is not None:
        f16_leg["hbm_notional_frac"] = alg_bytes / max(einfo["fill_launches"], 1) / (f16_leg["fill_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS
        f16_leg["same_rows_as_headline"] = bool(f16_leg.pop("rows_out") == res_recs)
        roofline["f16_cells"] = f16_leg

    out = {
        "metric": "decomposed read-bp/sec (whole node) at 12 monomers x 50kb reads",
        "value": bp_total * K / dt, "unit": "bp/s", "n_gpus": ws, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": info["cells"].split("/")[0],
        "data": "synthetic",
        "timed_region": "sequences in host memory -> chunk -> 2-bit pack -> H2D -> fill -> traceback -> compaction -> "
                        "D2H -> per-read assembly -> rows in host memory (SURVEY 8(d)); %d device batch(es) per step, %d "
                        "batches in flight (pinned staging, copies on their own streams), pipelined across steps: a step's "
                        "rows are collected when %d later step(s) have been submitted; every step's rows are in host memory "
                        "inside the timed region; kernel streams: %s" % (
                            args.sub_batches, BENCH_DEPTH + 1, BENCH_DEPTH, "one, in order (pipe_mode 0)" if args.pipe_mode == 0 else
                                         "overlapped (pipe_mode %d, the library default is 2)" % args.pipe_mode),
        "config": {"workload": "%s: synthetic %d reads x %d bp per GPU, %d monomers (~171 bp) + reverse complements, "
                               "default scoring -1,-1,-1,1, part 5000 / overlap 500" % (
                                   "C2" if (args.monomers, args.reads, args.read_len) == (12, 1000, 50000) else
                                   "C4 shape (DP only)" if args.monomers == 64 else "custom", args.reads, args.read_len, args.monomers),
                   "reads_per_gpu": args.reads, "read_len": args.read_len, "n_templates": info["n_templates"],
                   "sum_template_len": sumL, "chunks_per_gpu": n_chunks, "rows_per_gpu": rows,
                   "kernel_family": info["family"], "cells_per_lane": info["cells_per_lane"],
                   "cell_arithmetic": info["cells"] + (" (packed pairs holding exact integers)" if info["cells"] == "f16" else
                                                       " (packed pairs of biased unsigned 16-bit integers, exact over +-15 k)" if info["cells"] == "u16" else ""),
                   "sub_batches": args.sub_batches, "host_threads": threads, "ed_thr": args.ed_thr,
                   "seed": args.seed, "sharding": "reads dealt to ranks in contiguous blocks, no collective"},
        # contract figure: SURVEY 8(d) algorithmic bytes of a fill launch / its HIP-event duration, launches of
        # the timed region (in the default stream mode a fill's span contains the neighbouring batch's traceback
        # and the drain hand-over; isolated_* = the same kernel launched alone, from the device_resident steps)
        "roofline": roofline,
        "kernel_ms_per_step": {"fill": d["fill_ms"] / K, "traceback": d["trace_ms"] / K, "compact": d["compact_ms"] / K,
                               "note": "HIP-event spans per batch, summed; batches on the two streams overlap"},
        "host_ms_per_step": {"pack_upload_enqueue": d["host_pack_ms"] / K, "wait_for_device": d["host_wait_ms"] / K,
                             "d2h_assemble": d["host_assemble_ms"] / K,
                             "process_cpu_ms": d["host_cpu_ms_per_step"],
                             "cpu_quota_cores": cpu_quota_cores(),
                             "note": "wall ms of the host stages of one step; process_cpu_ms = CPU time of all host "
                                     "threads of this rank per step (user + system)"},
        "other_pipe_mode": other,
        "sustained": sustained,
        "amd_direct_dispatch": AMD_DIRECT_DISPATCH,
        # batches repeated with integer cells because the fp16 range guard tripped: must be 0 (checked below)
        "f16_guard_trips": lib.guard_trips(),
        "rows_out_per_gpu": rows_out, "hbm_workspace_bytes": einfo["workspace_bytes"] * (BENCH_DEPTH + 1),
        "hbm_workspace_per_engine_bytes": einfo["workspace_bytes"],   # (one engine per batch in flight)
        # kernels only, batch packed and resident in HBM before the clock starts, one launch per kernel
        "device_resident": None if not res_steps else {
            "bp_per_s": bp_rank * res_steps / dtr, "ms_per_step": dtr / res_steps * 1e3,
            "kernel_ms_per_step": {"fill": r_fill / res_steps, "traceback": r_trace / res_steps, "compact": r_cmp / res_steps},
            "fill_roofline_frac": res_achieved / HBM_PEAK_GBS, "load_reads_s": t_load},
    }
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        k = max(1, min(args.cpu_sample_reads, len(rs)))
        txt = lib.decompose(rn[:k], rs[:k], mn, ms, device=local_rank, kernel=kernel)
        out["cpu_baseline"] = cpu_baseline(mn, ms, rn[:k], rs[:k], txt)
    elif rank == 0:
        out["cpu_baseline"] = None
    # The line is only a result if the fp16 guard never tripped, and -- where a CPU baseline was asked for -- if that
    # baseline is the REAL reference binary and its rows equal the HIP path's on the sample: otherwise the line is
    # still printed (with "invalid") and the process exits non-zero.
    problems = []
    if out["f16_guard_trips"] != 0:
        problems.append("fp16 range guard tripped %d time(s)" % out["f16_guard_trips"])
    cb = out.get("cpu_baseline")
    if cb is not None and not args.allow_port_baseline:
        if cb["kind"] != "reference":
            problems.append("cpu_baseline.kind is %r (oracle/_ref/dp missing): not the reference binary" % cb["kind"])
        if not cb["parity_on_sample"]:
            problems.append("rows of the HIP path differ from the CPU baseline's on the sample")
    elif cb is not None and not cb["parity_on_sample"]:
        problems.append("rows of the HIP path differ from the CPU baseline's on the sample")
    if int16_leg is not None and not int16_leg["same_rows_as_headline"]:
        problems.append("integer-cell run produced a different number of rows")
    if f16_leg is not None and not f16_leg["same_rows_as_headline"]:
        problems.append("fp16-cell run produced a different number of rows")
    if problems:
        out["invalid"] = problems
    if profile_problems:   # (the measured value stands; only the figures derived from the committed counters are withheld)
        out["profile_problems"] = profile_problems
    if rank == 0:
        print(json.dumps(out), file=_claimed_stdout(), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if problems:
        print("bench.py: INVALID RESULT: " + "; ".join(problems), file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
