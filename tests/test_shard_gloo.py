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


# ---- one job sharded by chunk range ----------------------------------------------------
def _checker_range_fn(read_seqs, mono_seqs, lo, hi, scoring=(-1, -1, -1, 1), part_size=5000, overlap=500, **_):
    """Records of chunks [lo, hi) computed by the CPU oracle (test infrastructure): stands in for the
    HIP path so that the sharding / gather / assembly logic runs on a machine without a GPU."""
    import numpy as np
    from oracle import binding as oracle
    from stringdecomposer_amd import lib
    tmpl = [m if isinstance(m, str) else m.decode() for m in mono_seqs]
    tmpl = tmpl + [oracle.reverse_complement(m) for m in tmpl]
    table = []
    for s in read_seqs:
        table += [(s, off, ln) for off, ln in lib.chunk_plan(len(s), part_size, overlap)]
    recs, off = [], [0]
    for s, o, ln in table[lo:hi]:
        chunk = s[o:o + ln]
        recs += [(t, a, b, int(sc)) for t, a, b, sc in
                 oracle.align_chunk(chunk if isinstance(chunk, str) else chunk.decode(), tmpl, scoring)]
        off.append(len(recs))
    return np.array(recs, dtype=lib._rec_dtype()), np.array(off, dtype=np.int64)


def _shard_worker(rank, ws, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = shard.init_process_group("gloo")
    mn, ms = synth.make_monomers(3, seed=2)
    ms = [m[:60] for m in ms]
    # one long sequence (5 chunks) + short reads: the long one must split across the two ranks
    names, seqs = synth.make_reads(ms, 3, read_len=700, seed=2)
    n2, s2 = synth.make_reads(ms, 1, read_len=2450, seed=3)
    names, seqs = ["long"] + names, list(s2) + list(seqs)
    out = shard.decompose_sharded(names, seqs, mn, ms, dist=dist, range_fn=_checker_range_fn,
                                  scoring=(-1, -2, -1, 1), part_size=500, overlap=100, threads=2)
    q.put((rank, out))
    dist.destroy_process_group()


def test_one_job_sharded_by_chunk_range_world_size_2():
    from oracle import binding as oracle
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_shard_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[1] is None
    mn, ms = synth.make_monomers(3, seed=2)
    ms = [m[:60] for m in ms]
    names, seqs = synth.make_reads(ms, 3, read_len=700, seed=2)
    n2, s2 = synth.make_reads(ms, 1, read_len=2450, seed=3)
    names, seqs = ["long"] + names, list(s2) + list(seqs)
    exp = oracle.decompose(names, seqs, mn, ms, sc=(-1, -2, -1, 1), part=500, overlap=100)
    assert res[0] == exp
    # and the single-process form of the same driver
    one = shard.decompose_sharded(names, seqs, mn, ms, range_fn=_checker_range_fn,
                                  scoring=(-1, -2, -1, 1), part_size=500, overlap=100)
    assert one == exp


# ---- the file form used by the multi-process command line --------------------------------------------
def _checker_files_range_fn(reads_fa, monomers_fa, rank, ws, scoring=(-1, -1, -1, 1), part_size=5000, overlap=500, **_):
    """lib.decompose_files_range with the CPU oracle standing in for the device (test infrastructure); a read
    named 'bad...' makes the rank that owns one of its chunks fail like the alphabet check would."""
    from stringdecomposer_amd import lib
    names, seqs, _ = lib.fasta_load(reads_fa)
    mnames, mseqs, _ = lib.fasta_load(monomers_fa)
    n = lib.chunk_table_size([len(s) for s in seqs], part_size, overlap)
    lo, hi = shard.block_range(n, rank, ws)
    table = []
    for nm, s in zip(names, seqs):
        table += [nm] * len(lib.chunk_plan(len(s), part_size, overlap))
    for nm in table[lo:hi]:
        if nm.startswith("bad"):
            raise lib.SdError(lib.SD_ERR_SYMBOL, "ERROR: Sequence %s contains undefined symbol (not ACGT): x" % nm)
    recs, off = _checker_range_fn(seqs, mseqs, lo, hi, scoring=scoring, part_size=part_size, overlap=overlap)
    return recs, off, lo, hi, n


def _files_worker(rank, ws, port, q, d, bad):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stringdecomposer_amd import lib
    dist = shard.init_process_group("gloo")
    out = os.path.join(d, "raw_%s.tsv" % ("bad" if bad else "ok"))
    try:
        ok = shard.decompose_files_sharded(os.path.join(d, "bad.fa" if bad else "r.fa"), os.path.join(d, "m.fa"), out, dist,
                                           range_fn=_checker_files_range_fn, scoring=(-1, -2, -1, 1), part_size=500,
                                           overlap=100, threads=2)
        q.put((rank, "ok", ok))
    except lib.SdError as e:
        q.put((rank, "err", (e.code, e.msg)))
    dist.destroy_process_group()


def test_file_form_sharded_world_size_2_and_uniform_failure(tmp_path):
    """decompose_files_sharded (what `torch.distributed.run ... bin/stringdecomposer` executes): rank 0 writes
    the raw TSV of the whole job; a failure in ONE rank's share is raised by every rank (nobody is left
    waiting in the gather) with that rank's message."""
    from oracle import binding as oracle
    mn, ms = synth.make_monomers(3, seed=2)
    ms = [m[:60] for m in ms]
    names, seqs = synth.make_reads(ms, 3, read_len=700, seed=2)
    n2, s2 = synth.make_reads(ms, 1, read_len=2450, seed=3)
    names, seqs = ["long"] + names, list(s2) + list(seqs)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "r.fa"), names, seqs, width=70)
    synth.write_fasta(os.path.join(d, "bad.fa"), names[:3] + ["bad1"], seqs)
    synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
    for bad in (False, True):
        ws, port = 2, _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_files_worker, args=(r, ws, port, q, d, bad)) for r in range(ws)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=180) for _ in range(ws))
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        if not bad:
            assert [r[1:] for r in res] == [("ok", True), ("ok", None)]
            with open(os.path.join(d, "raw_ok.tsv"), "rb") as f:
                assert f.read() == oracle.decompose(names, seqs, mn, ms, sc=(-1, -2, -1, 1), part=500, overlap=100)
        else:
            msg = "ERROR: Sequence bad1 contains undefined symbol (not ACGT): x"
            assert [r[1:] for r in res] == [("err", (255, msg)), ("err", (255, msg))]


# ---- read-granular form: every rank runs its reads completely ------------------------------------------
def _checker_run_files_range(reads_fa, monomers_fa, rank, ws, raw_out, final_out, alt_out, min_identity=0,
                             second_best=False, lr_coef=None, scoring=(-1, -1, -1, 1), part_size=5000, overlap=500, **_):
    """lib.run_files_range with the CPU oracle standing in for the DP and the host identities for the device
    kernel (test infrastructure): same split of the reads, same three part files."""
    from oracle import binding as oracle
    from stringdecomposer_amd import lib
    names, seqs, _ = lib.fasta_load(reads_fa)
    nch = [len(lib.chunk_plan(len(s), part_size, overlap)) for s in seqs]
    total, cum = sum(nch), [0]
    for k in nch:
        cum.append(cum[-1] + k)
    if max(nch) * 2 * ws > total:
        raise lib.SdError(lib.SD_ERR_UNSUPPORTED, "read set cannot be split by reads")
    import bisect
    bound = lambda g: bisect.bisect_left(cum, total * g // ws)  # noqa: E731
    lo, hi = bound(rank), (len(seqs) if rank + 1 == ws else bound(rank + 1))
    mnames, mseqs, _ = lib.fasta_load(monomers_fa)
    raw = oracle.decompose(names[lo:hi], seqs[lo:hi], mnames, mseqs, sc=scoring, part=part_size, overlap=overlap) if hi > lo else b""
    with open(raw_out, "wb") as f:
        f.write(raw)
    lib.convert_raw_tsv(raw_out, reads_fa, monomers_fa, final_out, alt_out, min_identity, second_best,
                        lr_coef or (-31.48494996, 0.41784018, 0.69186882), device=-1, threads=2)
    return lo, hi, len(seqs), sum(nch[lo:hi])


def _run_files_worker(rank, ws, port, q, d, reads_name):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stringdecomposer_amd import lib
    dist = shard.init_process_group("gloo")
    out = os.path.join(d, "o_" + reads_name)
    try:
        ok = shard.run_files_sharded(os.path.join(d, reads_name), os.path.join(d, "m.fa"), out + "_raw.tsv", out + ".tsv",
                                     out + "_alt.tsv", dist, run_fn=_checker_run_files_range, second_best=True,
                                     scoring=(-1, -2, -1, 1), part_size=500, overlap=100, threads=2)
        q.put((rank, "ok", ok))
    except lib.SdError as e:
        q.put((rank, "err", (e.code, e.msg)))
    dist.destroy_process_group()


def test_read_granular_sharding_world_size_2(tmp_path):
    """run_files_sharded: every rank runs its group of reads completely and the part files land in the three
    outputs at their offsets == the single-process result; a read set dominated by one sequence is refused
    uniformly ("unsplittable") so that the launcher can fall back to chunk-range sharding."""
    from oracle import binding as oracle
    from stringdecomposer_amd import lib
    mn, ms = synth.make_monomers(3, seed=2)
    ms = [m[:60] for m in ms]
    names, seqs = synth.make_reads(ms, 7, read_len=900, seed=5)
    seqs[2] = seqs[2][:333]
    n2, s2 = synth.make_reads(ms, 1, read_len=30000, seed=6)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "many.fa"), names, seqs, width=70)
    synth.write_fasta(os.path.join(d, "one.fa"), ["chr"] + names[:2], list(s2) + seqs[:2])
    synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
    for reads_name in ("many.fa", "one.fa"):
        ws, port = 2, _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_run_files_worker, args=(r, ws, port, q, d, reads_name)) for r in range(ws)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=180) for _ in range(ws))
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        out = os.path.join(d, "o_" + reads_name)
        if reads_name == "one.fa":
            assert [r[1:] for r in res] == [("ok", "unsplittable")] * 2
            assert not os.path.exists(out + ".tsv") and not any(".part" in f for f in os.listdir(d))
            continue
        assert [r[1:] for r in res] == [("ok", True)] * 2
        raw = oracle.decompose(names, seqs, mn, ms, sc=(-1, -2, -1, 1), part=500, overlap=100)
        with open(out + "_raw.tsv", "rb") as f:
            assert f.read() == raw
        exp_raw = os.path.join(d, "exp_raw.tsv")
        with open(exp_raw, "wb") as f:
            f.write(raw)
        lib.convert_raw_tsv(exp_raw, os.path.join(d, "many.fa"), os.path.join(d, "m.fa"), os.path.join(d, "exp.tsv"),
                            os.path.join(d, "exp_alt.tsv"), 0, True, device=-1, threads=2)
        for a, b in ((out + ".tsv", "exp.tsv"), (out + "_alt.tsv", "exp_alt.tsv")):
            with open(a, "rb") as f, open(os.path.join(d, b), "rb") as g:
                assert f.read() == g.read()
        assert not any(".part" in f for f in os.listdir(d))


def _run_fn_rank1_oserror(reads_fa, monomers_fa, rank, ws, raw_out, final_out, alt_out, **kw):
    """rank 1 fails with something that is NOT an SdError after writing a part file; rank 0 succeeds."""
    r = _checker_run_files_range(reads_fa, monomers_fa, rank, ws, raw_out, final_out, alt_out, **kw)
    if rank == 1:
        raise OSError(28, "No space left on device")
    return r


def _oserror_worker(rank, ws, port, q, d):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stringdecomposer_amd import lib
    dist = shard.init_process_group("gloo")
    out = os.path.join(d, "o")
    try:
        shard.run_files_sharded(os.path.join(d, "many.fa"), os.path.join(d, "m.fa"), out + "_raw.tsv", out + ".tsv",
                                out + "_alt.tsv", dist, run_fn=_run_fn_rank1_oserror, second_best=False,
                                scoring=(-1, -1, -1, 1), part_size=500, overlap=100, threads=2)
        q.put((rank, "ok", None))
    except lib.SdError as e:
        q.put((rank, "err", (e.code, e.msg)))
    shard.barrier(dist)     # both ranks are still in step: a rank that had left the exchange would hang here
    dist.destroy_process_group()


def test_a_non_sd_error_on_one_rank_is_raised_on_every_rank(tmp_path):
    """only SdError used to reach the status exchange; an OSError on one rank left the others waiting in
    all_gather_object until the gloo timeout, with .partN files behind.  Now any exception travels as SD_ERR_INTERNAL
    and the part files are removed on every path."""
    from stringdecomposer_amd import lib
    mn, ms = synth.make_monomers(3, seed=2)
    ms = [m[:60] for m in ms]
    names, seqs = synth.make_reads(ms, 6, read_len=900, seed=5)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "many.fa"), names, seqs, width=70)
    synth.write_fasta(os.path.join(d, "m.fa"), mn, ms)
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_oserror_worker, args=(r, ws, port, q, d)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["err", "err"]
    assert all(r[2][0] == lib.SD_ERR_INTERNAL and "OSError" in r[2][1] for r in res)
    assert not any(".part" in f for f in os.listdir(d))


def _strong_worker(rank, ws, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = shard.init_process_group("gloo")
    c3 = shard.strong_share("c3", rank, ws, n_reads=1001)
    c5 = shard.strong_share("c5", rank, ws, seq_len=2_000_123)
    rate = shard.job_rate(dist, 1000.0, 1.0 + rank)     # the job is as slow as its slowest rank
    q.put((rank, c3, c5, rate))
    dist.destroy_process_group()


def test_strong_scaling_shares_world_size_2():
    """`bench.py --scaling strong`: ONE job split like the multi-process command line splits it -- blocks of reads
    (C3) or of the global chunk table (C5) that tile the job exactly -- and a whole-job rate over the MAX time."""
    from stringdecomposer_amd import lib
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_strong_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [("reads", 0, 501), ("reads", 501, 1001)]
    n_chunks = lib.chunk_table_size([2_000_123])
    assert n_chunks == len(lib.chunk_plan(2_000_123, 5000, 500))
    assert res[0][2] == ("chunks", 0, (n_chunks + 1) // 2, n_chunks) and res[1][2] == ("chunks", (n_chunks + 1) // 2, n_chunks, n_chunks)
    assert all(abs(r[3] - 500.0) < 1e-9 for r in res)


def _convert_worker(rank, ws, port, q, d, second_best):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stringdecomposer_amd import lib
    dist = shard.init_process_group("gloo")
    tag = "sb" if second_best else "light"
    try:
        shard.convert_sharded(os.path.join(d, "raw.tsv"), os.path.join(d, "r.fa"), os.path.join(d, "m.fa"),
                              os.path.join(d, "final_%s.tsv" % tag), os.path.join(d, "alt_%s.tsv" % tag), dist,
                              second_best=second_best, device=-1, threads=2)
        # a failure on ONE rank (its raw file is missing) is everybody's
        try:
            shard.convert_sharded(os.path.join(d, "raw.tsv") if rank != 1 else os.path.join(d, "nope.tsv"),
                                  os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), os.path.join(d, "x.tsv"),
                                  os.path.join(d, "xa.tsv"), dist, second_best=second_best, device=-1, threads=1)
            q.put((rank, "no error"))
        except lib.SdError as e:
            q.put((rank, e.code))
    finally:
        shard.barrier(dist)
        dist.destroy_process_group()


@pytest.mark.parametrize("second_best", [False, True])
def test_convert_tsv_shared_by_three_ranks_equals_one_process(tmp_path, second_best):
    """SURVEY 8(e) / VERDICT r03 item 3c: the job of one huge sequence is sharded by chunk range, rank 0 assembles the raw
    TSV -- and convert_tsv (main.py:168-184) is then shared again: every rank converts the rows that begin in its byte
    range of the raw file (rows are independent, main.py:95-150) into part files, copied into the final / _alt files at
    their offsets.  Three gloo ranks (host identities, no device) against the same conversion in one process, on the raw
    rows of the reference's test read (ONE read: every range is a piece of it); a failure on one rank is raised on all,
    and no part file stays behind."""
    import shutil
    from conftest import GOLDEN
    from stringdecomposer_amd import lib
    from oracle import binding as oracle
    d = str(tmp_path)
    shutil.copy(os.path.join(GOLDEN, "test_data", "read.fa"), os.path.join(d, "r.fa"))
    shutil.copy(os.path.join(GOLDEN, "test_data", "DXZ1_star_monomers.fa"), os.path.join(d, "m.fa"))
    raw = oracle.decompose_files(os.path.join(d, "r.fa"), os.path.join(d, "m.fa"), threads=8)
    with open(os.path.join(d, "raw.tsv"), "wb") as f:
        f.write(raw)
    tag = "sb" if second_best else "light"
    lib.convert_raw_tsv(os.path.join(d, "raw.tsv"), os.path.join(d, "r.fa"), os.path.join(d, "m.fa"),
                        os.path.join(d, "one_final.tsv"), os.path.join(d, "one_alt.tsv"), second_best=second_best,
                        device=-1, threads=4)
    ws, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_convert_worker, args=(r, ws, port, q, d, second_best)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(ws))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [lib.SD_ERR_IO] * ws
    for a, b in (("one_final.tsv", "final_%s.tsv" % tag), ("one_alt.tsv", "alt_%s.tsv" % tag)):
        with open(os.path.join(d, a), "rb") as fa, open(os.path.join(d, b), "rb") as fb:
            one, many = fa.read(), fb.read()
        assert one == many and (one or a == "one_alt.tsv")
    assert open(os.path.join(d, "one_final.tsv"), "rb").read().count(b"\n") == 557
    assert not any(".part" in f for f in os.listdir(d))
    # byte ranges that cut lines anywhere: 1..7 ranges of a small file tile it exactly (one process, no exchange)
    small = raw[: raw.index(b"\n", 3000) + 1]
    with open(os.path.join(d, "small.tsv"), "wb") as f:
        f.write(small)
    lib.convert_raw_tsv(os.path.join(d, "small.tsv"), os.path.join(d, "r.fa"), os.path.join(d, "m.fa"),
                        os.path.join(d, "s_final.tsv"), os.path.join(d, "s_alt.tsv"), second_best=second_best, device=-1)
    want = open(os.path.join(d, "s_final.tsv"), "rb").read()
    for w in (1, 2, 5, 7, 64):
        got = b""
        for g in range(w):
            lib.convert_raw_tsv_range(os.path.join(d, "small.tsv"), os.path.join(d, "r.fa"), os.path.join(d, "m.fa"),
                                      os.path.join(d, "p_final.tsv"), os.path.join(d, "p_alt.tsv"), g, w,
                                      second_best=second_best, device=-1)
            got += open(os.path.join(d, "p_final.tsv"), "rb").read()
        assert got == want, w


# ---- every rank assembles its own range (csrc/sd_seam.hpp) --------------------------------------------
_SEAM_READS = [2500, 260_000, 40, 30_000]      # a chromosome over all ranks, small reads either side of it
_SEAM_PART, _SEAM_OV = 100, 30


def _seam_job():
    """Synthetic per-chunk records (no DP): neighbours overlap often enough for every branch of main.cpp:287-302."""
    import random
    import numpy as np
    from stringdecomposer_amd import lib
    rnd = random.Random(77)
    recs, off = [], [0]
    for rl in _SEAM_READS:
        for (_o, ln) in lib.chunk_plan(rl, _SEAM_PART, _SEAM_OV):
            pos, k = 0, 0
            while pos < ln - 5 and k <= 60:
                w = rnd.randint(3, 40)
                e = min(ln - 1, pos + w)
                recs.append((rnd.randrange(6), pos, e, rnd.randint(-5, 30)))
                pos = max(0, e - rnd.randint(0, w)) if rnd.random() < 0.3 else e + 1 + rnd.randint(0, 3)
                k += 1
            off.append(len(recs))
    return np.array(recs, dtype=lib._rec_dtype()), np.array(off, dtype=np.int64)


def _seam_range_fn(read_seqs, mono_seqs, lo, hi, **_):
    recs, off = _seam_job()
    return recs[off[lo]:off[hi]], off[lo:hi + 1] - off[lo]


def _seam_files_fn(reads_fa, monomers_fa, rank, ws, **_):
    recs, off = _seam_job()
    n = len(off) - 1
    lo, hi = shard.block_range(n, rank, ws)
    return recs[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], lo, hi, n


def _seam_worker(rank, ws, port, q, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = shard.init_process_group("gloo")
    names = ["r%d" % i for i in range(len(_SEAM_READS))]
    seqs = ["A" * n for n in _SEAM_READS]
    mono = ["m0", "m1", "m2"]
    st1, st2 = {}, {}
    out = shard.decompose_sharded(names, seqs, mono, ["ACGT"] * 3, dist=dist, range_fn=_seam_range_fn,
                                  part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2, assemble_stats=st1)
    raw = os.path.join(tmp, "raw.tsv")
    ok = shard.decompose_files_sharded(os.path.join(tmp, "r.fa"), os.path.join(tmp, "m.fa"), raw, dist,
                                       range_fn=_seam_files_fn, part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2,
                                       assemble_stats=st2)
    os.environ["SD_SHARD_GATHER"] = "1"       # the rank-0 assembly of the same job
    old = shard.decompose_sharded(names, seqs, mono, ["ACGT"] * 3, dist=dist, range_fn=_seam_range_fn,
                                  part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2)
    q.put((rank, out, ok, old, st1, st2))
    dist.destroy_process_group()


@pytest.mark.parametrize("ws", [2, 3])
def test_every_rank_writes_its_own_part_of_the_raw_tsv(ws, tmp_path):
    """decompose_sharded / decompose_files_sharded over gloo: the ranks exchange 160-byte edges and text sizes, each
    makes the text of its own chunk range and writes it into the file at its offset; bytes equal the serial assembly of
    all records and the gather path (SD_SHARD_GATHER) of the same job."""
    from stringdecomposer_amd import lib
    with open(tmp_path / "r.fa", "w") as f:
        for i, n in enumerate(_SEAM_READS):
            f.write(">r%d\n%s\n" % (i, "A" * n))
    with open(tmp_path / "m.fa", "w") as f:
        for i in range(3):
            f.write(">m%d\nACGT\n" % i)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_seam_worker, args=(r, ws, port, q, str(tmp_path))) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    recs, off = _seam_job()
    want = lib.assemble_tsv(["r%d" % i for i in range(len(_SEAM_READS))], _SEAM_READS, ["m0", "m1", "m2"], recs, off,
                            part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2)
    assert res[0][1] == want and all(r[1] is None for r in res[1:])
    assert res[0][3] == want
    assert res[0][2] is True and all(r[2] is None for r in res[1:])
    assert (tmp_path / "raw.tsv").read_bytes() == want
    for r in res:       # the rank-local path ran on every rank, for both forms
        assert r[4].get("text_bytes", 0) > 0 and r[5].get("text_bytes", 0) > 0, r[4]
    assert sum(r[5]["text_bytes"] for r in res) == len(want)


class _FailingAssembler:
    """A RangeAssembler whose `bytes()` or `records()` raises on one rank (those two calls could fail on one
    rank only, MemoryError on a 50-MB share, and the others then waited in gather_object for ever)."""

    def __init__(self, inner, fail_at, refuse, lo, hi, n):
        self._a, self._fail, self._refuse = inner, fail_at, refuse
        self.chunk_lo, self.chunk_hi, self.n_chunks = lo, hi, n
        self._share = None

    @property
    def edge(self):
        return None if self._refuse else self._a.edge       # an edge of None makes every rank take the fall-back

    def text(self, edges, rank):
        return self._a.text(edges, rank)

    def bytes(self):
        if self._fail == "bytes":
            raise MemoryError("injected")
        return self._a.bytes()

    def records(self):
        if self._fail == "records":
            raise MemoryError("injected")
        return self._share       # (from_lists leaves the records with its caller)

    def stats(self):
        return self._a.stats()

    def close(self):
        self._a.close()


def _fail_point_worker(rank, ws, port, q, point):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stringdecomposer_amd import lib
    dist = shard.init_process_group("gloo")
    names = ["r%d" % i for i in range(len(_SEAM_READS))]
    recs, off = _seam_job()
    lo, hi = shard.block_range(len(off) - 1, rank, ws)

    def make():
        a = lib.RangeAssembler.from_lists(names, _SEAM_READS, ["m0", "m1", "m2"], lo, hi, recs[off[lo]:off[hi]],
                                          off[lo:hi + 1] - off[lo], part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2)
        f = _FailingAssembler(a, point if rank == 1 else None, point == "records", lo, hi, len(off) - 1)
        f._share = (recs[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
        return f
    try:
        done = shard._assemble_by_ranks(dist, rank, ws, make, refused={})
        q.put((rank, "ok", done is None))
    except lib.SdError as e:
        q.put((rank, "err", (e.code, e.msg)))
    shard.barrier(dist)      # every rank left _assemble_by_ranks at the same point of the collective sequence
    dist.destroy_process_group()


@pytest.mark.parametrize("point", ["bytes", "records"])
def test_a_failure_of_one_rank_behind_the_edge_exchange_is_raised_on_every_rank(point):
    from stringdecomposer_amd import lib
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fail_point_worker, args=(r, ws, port, q, point)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["err", "err"], res
    assert all(r[2][0] == lib.SD_ERR_INTERNAL and "MemoryError" in r[2][1] for r in res)


def _unshared_worker(rank, ws, port, q, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ws),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["SD_SHARD_FAKE_UNSHARED"] = "1"      # rank 1 behaves as if its output directory were another node's
    dist = shard.init_process_group("gloo")
    st = {}
    raw = os.path.join(tmp, "raw.tsv")
    ok = shard.decompose_files_sharded(os.path.join(tmp, "r.fa"), os.path.join(tmp, "m.fa"), raw, dist,
                                       range_fn=_seam_files_fn, part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2,
                                       assemble_stats=st)
    q.put((rank, ok, st.get("gathered_because_not_one_file_system", False)))
    dist.destroy_process_group()


def test_ranks_that_do_not_share_the_output_directory_gather_on_rank_0(tmp_path):
    """every rank writes its own byte range of the raw TSV, which is one file only on one file system.  A rank
    that does not find rank 0's token beside the output makes ALL ranks send their texts to rank 0 instead (decided in
    the exchange of the text sizes, before anybody writes): same bytes, no holes, no silent success."""
    from stringdecomposer_amd import lib
    with open(tmp_path / "r.fa", "w") as f:
        for i, n in enumerate(_SEAM_READS):
            f.write(">r%d\n%s\n" % (i, "A" * n))
    with open(tmp_path / "m.fa", "w") as f:
        for i in range(3):
            f.write(">m%d\nACGT\n" % i)
    ws, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_unshared_worker, args=(r, ws, port, q, str(tmp_path))) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(ws))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    recs, off = _seam_job()
    want = lib.assemble_tsv(["r%d" % i for i in range(len(_SEAM_READS))], _SEAM_READS, ["m0", "m1", "m2"], recs, off,
                            part_size=_SEAM_PART, overlap=_SEAM_OV, threads=2)
    assert (tmp_path / "raw.tsv").read_bytes() == want
    assert all(r[2] for r in res), res
    assert not [f for f in os.listdir(tmp_path) if "ranks-share" in f]
