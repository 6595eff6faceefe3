#!/usr/bin/env python3
"""What follows the DP when ONE sequence is sharded by chunk range over N processes (BASELINE config 5's multi-GPU form):
records -> raw TSV file, (a) every rank its own range (round 5, shard._assemble_by_ranks), (b) gathered on rank 0
(SD_SHARD_GATHER=1, rounds 1-4).  Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P tools/c5_ranks_files.py [seq-len, default 200000000] [sha256 of the expected raw TSV]
All ranks share GPU 0 here (one-GPU box), so the DP times are serialised and NOT the point; the time from "every rank has
its records" to "file complete" is.  Rank 0 prints one JSON line."""
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stringdecomposer_amd import lib, shard, synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
KNOWN = sys.argv[2] if len(sys.argv) > 2 else None
rank, local_rank, ws = shard.world()
dist = shard.init_process_group("gloo")
sc = (-2, -3, -4, 2)
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (usable_cores: the cgroup CPU quota, not the 256 logical CPUs the box shows)
threads = max(1, min(32, bench.usable_cores() // ws))
shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
d = os.path.join(shm, "sd_c5_ranks_%s" % os.environ.get("MASTER_PORT", "0"))
if rank == 0:
    os.makedirs(d, exist_ok=True)
    mn, ms = synth.make_monomers(12, seed=1)
    _, rs = synth.make_reads(ms, 1, read_len=L, seed=1)
    synth.write_fasta(os.path.join(d, "r.fa"), ["seq0"], rs, width=0)
    synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
shard.barrier(dist)
rfa, mfa, out = os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), os.path.join(d, "raw.tsv")
marks = {}


def range_fn(reads_fa, monomers_fa, rk, world, **kw):
    res = lib.decompose_files_range(reads_fa, monomers_fa, rk, world, **kw)
    shard.barrier(dist)                 # every rank has its records: what follows is the part this tool measures
    marks["t"] = time.perf_counter()
    return res


_run_files = lib.RangeAssembler.run_files


def run_files_then_wait(*a, **kw):
    # the shipped path: DP + first step of the assembly in one library call.  The processes share one GPU here, so their
    # DPs end one after the other; the clock starts when the last one has, and the first step's own time (stats:
    # begin_ms, part of the call) is added back below
    h = _run_files(*a, **kw)
    shard.barrier(dist)
    marks["t"] = time.perf_counter()
    return h


lib.RangeAssembler.run_files = staticmethod(run_files_then_wait)


res = {}
for mode in ("ranks", "ranks", "gather", "gather"):
    if mode == "gather":
        os.environ["SD_SHARD_GATHER"] = "1"
    else:
        os.environ.pop("SD_SHARD_GATHER", None)
    st = {}
    shard.barrier(dist)
    t0 = time.perf_counter()
    shard.decompose_files_sharded(rfa, mfa, out, dist, range_fn=range_fn if mode == "gather" else None, scoring=sc,
                                  threads=threads, device=0, assemble_stats=st)
    shard.barrier(dist)
    t1 = time.perf_counter()
    after = shard.max_over_ranks(dist, t1 - marks["t"] + st.get("begin_ms", 0.0) * 1e-3)
    if rank == 0:
        with open(out, "rb") as f:
            h = hashlib.sha256(f.read()).hexdigest()
        res.setdefault(mode, []).append({"records_to_file_ms": round(after * 1e3, 2), "whole_call_ms": round((t1 - t0) * 1e3, 1),
                                         "sha256": h, "rank0_stats": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}})
shard.barrier(dist)
if rank == 0:
    hs = {r["sha256"] for m in res.values() for r in m}
    print(json.dumps({"workload": "one sequence of %d bp over %d processes (sharing GPU 0), %d host threads each" % (L, ws, threads),
                      "every_rank_its_own_range": res["ranks"], "gathered_on_rank_0": res["gather"],
                      "same_file_both_ways": len(hs) == 1, "equals_known_hash": (KNOWN in hs and len(hs) == 1) if KNOWN else None}))
    import shutil
    shutil.rmtree(d, ignore_errors=True)
dist.destroy_process_group()
