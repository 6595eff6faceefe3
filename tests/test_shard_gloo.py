"""N>1 path on CPU: world_size-2 gloo processes.  Checks that the shards partition the read set
exactly (so concatenating per-rank rows in rank order equals the single-process result), that
weak-scaling shards regenerate the same reads as a single process would, and that the timing
reduction is a MAX over ranks."""
import hashlib
import os
import socket

import pytest
import torch.multiprocessing as mp

from stringdecomposer_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = shard.init_process_group("gloo")
    mn, ms = synth.make_monomers(12, seed=1)
    lo, hi = shard.block_range(7, rank, ws)
    names, seqs = synth.make_reads(ms, hi - lo, read_len=3000, seed=1, first_index=lo)
    wlo, whi = shard.weak_range(3, rank)
    wn, wseqs = synth.make_reads(ms, whi - wlo, read_len=2000, seed=1, first_index=wlo)
    shard.barrier(dist)
    t = shard.max_over_ranks(dist, 1.0 + rank)
    tot = shard.sum_over_ranks(dist, sum(len(s) for s in seqs))
    q.put((rank, lo, hi, names, [hashlib.sha1(s).hexdigest() for s in seqs],
           wn, [hashlib.sha1(s).hexdigest() for s in wseqs], t, tot))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    mn, ms = synth.make_monomers(12, seed=1)
    names, seqs = synth.make_reads(ms, 7, read_len=3000, seed=1)
    assert [r[1] for r in res] == [0, 4] and [r[2] for r in res] == [4, 7]
    assert sum((r[3] for r in res), []) == names
    assert sum((r[4] for r in res), []) == [hashlib.sha1(s).hexdigest() for s in seqs]
    wn, wseqs = synth.make_reads(ms, 6, read_len=2000, seed=1)
    assert sum((r[5] for r in res), []) == wn
    assert sum((r[6] for r in res), []) == [hashlib.sha1(s).hexdigest() for s in wseqs]
    assert all(r[7] == 2.0 for r in res)           # MAX over ranks
    assert all(r[8] == 7 * 3000 for r in res)      # SUM over ranks


@pytest.mark.parametrize("n,ws", [(0, 1), (1, 1), (10, 3), (1000, 8), (7, 8)])
def test_block_range_partitions(n, ws):
    got = []
    for r in range(ws):
        lo, hi = shard.block_range(n, r, ws)
        assert 0 <= hi - lo <= n // ws + 1
        got += list(range(lo, hi))
    assert got == list(range(n))
