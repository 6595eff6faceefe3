"""Multi-GPU sharding of the hot path: independent chunks, no data-path collective.

A chunk's DP depends only on its <= part+overlap bases and the (replicated) template set
(reference main.cpp:88-96).  Two granularities:
  * bench.py (weak scaling): reads are dealt to ranks in contiguous blocks, every rank runs the
    single-GPU engine on its block;
  * decompose_sharded (one job, strong scaling): the chunks of ALL reads form one global table and
    rank g takes the contiguous chunk range block_range(n_chunks, g, G) -- a single 200-Mb sequence
    (BASELINE config 5) spreads over the GPUs exactly like a million reads.  Every rank
    then turns ITS records into ITS part of the raw TSV -- chunk offsets, seam merge, text (host only): the
    merge of a read that crosses range boundaries needs eight records from either side of a boundary, which
    the ranks exchange as 160-byte edges (_assemble_by_ranks, csrc/sd_seam.hpp), and the ranks write their
    texts into the output file at their offsets.  Only when a share is empty or a crossing piece is
    shorter than 32 records are the compact records (24 B per ~171 bp) gathered on rank 0 instead.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the barrier / max-over-ranks
timing and for those small exchanges; there is no collective on the data path.
"""
import os


def world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def block_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of n_items for `rank`: sizes differ by at most one, order kept."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def weak_range(items_per_rank, rank):
    """Weak scaling: every rank owns items_per_rank consecutive items of an unbounded stream."""
    return rank * items_per_rank, (rank + 1) * items_per_rank


def strong_share(config, rank, world_size, n_reads=0, seq_len=0, part_size=5000, overlap=500):
    """What `rank` does of ONE job split over `world_size` GPUs (strong scaling; `bench.py --scaling strong`), exactly
    as the multi-process command line splits it -- no data-path collective in either form:
      "c3"  a set of reads (BASELINE config 3): contiguous blocks of reads, as run_files_sharded deals them when all
            reads have the same length -> ("reads", lo, hi);
      "c5"  one huge sequence (BASELINE config 5): contiguous blocks of the global chunk table (main.cpp:70-81), as
            decompose_files_sharded -> ("chunks", lo, hi, n_chunks)."""
    if config == "c3":
        lo, hi = block_range(int(n_reads), rank, world_size)
        return ("reads", lo, hi)
    if config == "c5":
        from . import lib
        n_chunks = lib.chunk_table_size([int(seq_len)], part_size, overlap)
        lo, hi = block_range(n_chunks, rank, world_size)
        return ("chunks", lo, hi, n_chunks)
    raise ValueError("strong_share: config must be c3 or c5")


def job_rate(dist, total_units, my_seconds, device="cpu"):
    """Whole-job rate of a strong-scaling run: the job's units over the time of its slowest rank."""
    return float(total_units) / max_over_ranks(dist, my_seconds, device)


def init_process_group(backend=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns the module or None."""
    rank, local_rank, ws = world()
    if ws <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return dist


def barrier(dist, device=None):
    if dist is not None:
        if device is not None:
            dist.barrier(device_ids=[device])
        else:
            dist.barrier()


def max_over_ranks(dist, value, device="cpu"):
    """MAX over ranks of a python float (the job is as slow as its slowest rank)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def _name_reads(msg, read_names):
    """The chunk-range entry point knows reads by index only ("Sequence #12 ..."): put the read's
    name back so that the text equals the single-process / reference message (main.cpp:335)."""
    import re

    def sub(m):
        i = int(m.group(1))
        nm = read_names[i] if 0 <= i < len(read_names) else m.group(0)
        return "Sequence " + (nm.decode() if isinstance(nm, bytes) else str(nm))
    return re.sub(r"Sequence #(\d+)", sub, msg)


def _edges_ok(edges):
    from . import lib
    return all(e is not None and lib.SeamEdge.from_buffer_copy(e).ok for e in edges)


def _assemble_by_ranks(dist, rank, ws, make, raw_tsv_out=None, stats=None, refused=None):
    """Every rank makes the raw TSV text of its own chunk range (lib.RangeAssembler: `make()` builds this rank's).
    Two small exchanges -- the 160-byte edges, then the text sizes -- and, with `raw_tsv_out`, all ranks write into
    the file at their offsets (each sets the file's size to the same total first, none waits for another); without it
    the texts are gathered on rank 0.  Returns None when the job cannot be shared this way (decided identically on
    all ranks from the edges: gather the records instead; `refused`, a dict, then receives the share's records when
    the assembler holds them), else (True, text-or-None).  A failure on any rank is raised on every rank."""
    import time
    from . import lib
    failure, a, edge = None, None, None
    t0 = time.perf_counter()
    # Every rank writes its own byte range of `raw_tsv_out`: that is one file only where the ranks see one file system.  On
    # a multi-node launch without a shared output directory rank 0's file would keep zero-filled holes and the job would
    # still report success.  So rank 0 leaves a token beside the output before the DP, the token travels with
    # the edges, every rank looks for it after that exchange and says what it found in the exchange of the text sizes: if
    # any rank did not see it, the texts are gathered on rank 0, which writes the file alone -- as in rounds 1-4.
    token, probe = None, None
    if raw_tsv_out is not None:
        probe = raw_tsv_out + ".ranks-share-this-directory"
        if rank == 0:
            try:
                token = os.urandom(16).hex()
                with open(probe, "w") as f:
                    f.write(token)
            except OSError:
                token = None
    try:
        a = make()
        edge = a.edge
    except Exception as e:
        failure = _status_of(e)
    t1 = time.perf_counter()
    box = [None] * ws
    dist.all_gather_object(box, (failure, edge, token))
    t2 = time.perf_counter()
    first = next((b[0] for b in box if b[0] is not None), None)
    sees_file = True
    if raw_tsv_out is not None:
        try:
            with open(probe) as f:
                sees_file = box[0][2] is not None and f.read() == box[0][2]
        except OSError:
            sees_file = False
        if os.environ.get("SD_SHARD_FAKE_UNSHARED") == str(rank):   # (test hook: this rank behaves as if on another node)
            sees_file = False
    try:
        if first is not None:
            raise lib.SdError(*first)
        edges = [b[1] for b in box]
        if not _edges_ok(edges):
            if refused is not None and hasattr(a, "chunk_lo"):
                # the caller goes on to a gather: a rank that cannot hand its records back (SdError, MemoryError on a
                # large share) must not leave the others waiting there -- every rank learns every rank's status first
                try:
                    refused["recs"], refused["off"] = a.records()
                    refused["range"] = (a.chunk_lo, a.chunk_hi, a.n_chunks)
                except Exception as e:
                    failure = _status_of(e)
                _raise_first(dist, ws, failure)
            return None
        n = None
        try:
            n = a.text(edges, rank)
        except Exception as e:
            failure = _status_of(e)
        t3 = time.perf_counter()
        sizes = [None] * ws
        dist.all_gather_object(sizes, (failure, n, sees_file))
        first = next((b[0] for b in sizes if b[0] is not None), None)
        if first is not None:
            raise lib.SdError(*first)
        shared = all(b[2] for b in sizes)
        sizes = [b[1] for b in sizes]
        t4 = time.perf_counter()
        text = None
        if raw_tsv_out is not None and not shared:
            # not one file system: the texts travel to rank 0, which writes the whole file
            mine = None
            try:
                mine = a.bytes()
            except Exception as e:
                failure = _status_of(e)
            _raise_first(dist, ws, failure)
            got = [None] * ws if rank == 0 else None
            dist.gather_object(mine, got, dst=0)
            if rank == 0:
                try:
                    with open(raw_tsv_out, "wb") as f:
                        f.write(b"".join(got))
                except Exception as e:
                    failure = _status_of(e)
            _raise_first(dist, ws, failure)
            if stats is not None:
                stats["gathered_because_not_one_file_system"] = True
        elif raw_tsv_out is not None:
            try:
                a.write(raw_tsv_out, sum(sizes[:rank]), sum(sizes))
            except Exception as e:
                failure = _status_of(e)
            _raise_first(dist, ws, failure)     # the file is complete when this returns
        else:
            mine = None
            try:
                mine = a.bytes()
            except Exception as e:
                failure = _status_of(e)
            _raise_first(dist, ws, failure)     # nobody enters the gather unless everybody has its text
            got = [None] * ws if rank == 0 else None
            dist.gather_object(mine, got, dst=0)
            text = b"".join(got) if rank == 0 else None
        if stats is not None:
            stats.update(a.stats())
            stats.update({"begin_wall_ms": (t1 - t0) * 1e3, "edge_exchange_ms": (t2 - t1) * 1e3,
                          "text_wall_ms": (t3 - t2) * 1e3, "size_exchange_ms": (t4 - t3) * 1e3,
                          "write_or_gather_ms": (time.perf_counter() - t4) * 1e3, "text_bytes": sizes[rank]})
        return (True, text)
    finally:
        if a is not None:
            a.close()
        if probe is not None and rank == 0:
            try:
                os.remove(probe)
            except OSError:
                pass


def decompose_sharded(read_names, read_seqs, mono_names, mono_seqs, dist=None, range_fn=None, **params):
    """Raw TSV (bytes) of the whole job on rank 0, None on the other ranks.

    Every rank holds the full read set (it parsed the same FASTA), computes the records of its own
    contiguous range of the global chunk table on its GPU and sends them to rank 0.
    `range_fn(read_seqs, mono_seqs, lo, hi, **params) -> (recs, rec_off)` defaults to the HIP path
    (lib.decompose_chunk_range); the CPU tests inject a checker-based one."""
    import numpy as np
    from . import lib
    rank, local_rank, ws = world() if dist is not None else (0, 0, 1)
    asm_stats = params.pop("assemble_stats", None)   # a dict that receives the stage times of the rank-local assembly
    part = int(params.get("part_size", 5000))
    overlap = int(params.get("overlap", 500))
    read_lens = [len(s) for s in read_seqs]
    n_chunks = lib.chunk_table_size(read_lens, part, overlap)
    lo, hi = block_range(n_chunks, rank, ws)
    if range_fn is None:
        params = dict(params, device=params.get("device", local_rank))
        range_fn = lib.decompose_chunk_range
    # A rank whose range holds an undefined symbol (or whose device fails) must not leave the others
    # blocked in the gather: every rank first learns every rank's status and all of them raise the
    # error of the lowest-numbered failing rank -- its range comes first in the chunk table, so that
    # is the read the reference (which checks reads in input order, main.cpp:329-341) would report.
    failure = None
    recs = off = None
    try:
        recs, off = range_fn(read_seqs, mono_seqs, lo, hi, **params)
    except lib.SdError as e:
        failure = (e.code, _name_reads(e.msg, read_names))
    if dist is not None and ws > 1:
        status = [None] * ws
        dist.all_gather_object(status, failure)
        failure = next((s for s in status if s is not None), None)
    if failure is not None:
        raise lib.SdError(*failure)
    recs = np.ascontiguousarray(recs)
    off = np.ascontiguousarray(off, dtype=np.int64)
    keep = {k: v for k, v in params.items() if k in ("scoring", "part_size", "overlap", "threads")}
    if dist is not None and ws > 1 and not os.environ.get("SD_SHARD_GATHER"):   # (knob: the rank-0 assembly of rounds 1-4)
        done = _assemble_by_ranks(dist, rank, ws, lambda: lib.RangeAssembler.from_lists(
            read_names, read_lens, mono_names, lo, hi, recs, off, **keep), stats=asm_stats)
        if done is not None:
            return done[1]
    if dist is None or ws == 1:
        parts = [(lo, recs, off)]
    else:
        box = [None] * ws if rank == 0 else None
        dist.gather_object((lo, recs, off), box, dst=0)
        if rank != 0:
            return None
        parts = sorted(box, key=lambda t: t[0])
    all_recs = np.concatenate([p[1] for p in parts]) if parts else recs
    offs = [np.zeros(1, dtype=np.int64)]
    base = 0
    for _, r, o in parts:
        offs.append(o[1:] + base)
        base += len(r)
    all_off = np.concatenate(offs)
    assert len(all_off) == n_chunks + 1
    return lib.assemble_tsv(read_names, read_lens, mono_names, all_recs, all_off, **keep)


def decompose_files_sharded(reads_fa, monomers_fa, raw_tsv_out, dist, range_fn=None, **params):
    """The multi-process command line: every rank maps the FASTA files itself (sequences never enter Python),
    runs its contiguous share of the global chunk table on its GPU and sends the compact records to rank 0,
    which writes the raw TSV file.  Returns True on rank 0, None elsewhere.  A failure on any rank (an
    undefined symbol in its share, a device error) is raised on EVERY rank: the lowest failing rank's, i.e.
    the first offending read in file order, as the reference would report it."""
    import numpy as np
    from . import lib
    rank, local_rank, ws = world()
    asm_stats = params.pop("assemble_stats", None)
    params = dict(params, device=params.get("device", local_rank))
    fn = range_fn or lib.decompose_files_range
    keep = {k: v for k, v in params.items() if k in ("scoring", "part_size", "overlap", "threads")}
    gather_only = bool(os.environ.get("SD_SHARD_GATHER"))   # developer knob: the rank-0 assembly of rounds 1-4
    failure, res = None, None
    if range_fn is None and not gather_only:
        # DP of the share and the first step of the assembly in ONE library call: the FASTA is indexed once and the
        # records stay in the library (a failure of the DP travels with the edges and is raised on every rank)
        refused = {}
        done = _assemble_by_ranks(dist, rank, ws, lambda: lib.RangeAssembler.run_files(reads_fa, monomers_fa, rank, ws, **params),
                                  raw_tsv_out=raw_tsv_out, stats=asm_stats, refused=refused)
        if done is not None:
            return True if rank == 0 else None
        recs, off = refused["recs"], refused["off"]
        lo, hi, n_chunks = refused["range"]
    else:
        try:
            res = fn(reads_fa, monomers_fa, rank, ws, **params)
        except Exception as e:   # ANY failure is exchanged: a rank that left the collective sequence would hang the others
            failure = _status_of(e)
        _raise_first(dist, ws, failure)
        recs, off, lo, hi, n_chunks = res
        if not gather_only:
            done = _assemble_by_ranks(dist, rank, ws, lambda: lib.RangeAssembler.from_files(
                reads_fa, monomers_fa, rank, ws, recs, off, **keep), raw_tsv_out=raw_tsv_out, stats=asm_stats)
            if done is not None:
                return True if rank == 0 else None
    box = [None] * ws if rank == 0 else None
    dist.gather_object((lo, recs, off), box, dst=0)
    failure = None
    if rank == 0:
        try:
            parts = sorted(box, key=lambda t: t[0])
            all_recs = np.concatenate([p[1] for p in parts])
            offs, base = [np.zeros(1, dtype=np.int64)], 0
            for _, r, o in parts:
                offs.append(o[1:] + base)
                base += len(r)
            all_off = np.concatenate(offs)
            assert len(all_off) == n_chunks + 1
            lib.assemble_files_tsv(reads_fa, monomers_fa, all_recs, all_off, raw_tsv_out, **keep)
        except Exception as e:
            failure = _status_of(e)
    _raise_first(dist, ws, failure)   # rank 0's assembly failing is everybody's failure too
    return True if rank == 0 else None


def _status_of(e):
    """(code, message) of an exception for the status exchange; anything that is not an SdError travels as
    SD_ERR_INTERNAL with its type in the text."""
    from . import lib
    if isinstance(e, lib.SdError):
        return (e.code, e.msg)
    return (lib.SD_ERR_INTERNAL, "%s: %s" % (type(e).__name__, e))


def _raise_first(dist, ws, failure):
    """All ranks exchange their status; the lowest failing rank's failure is raised on every rank."""
    from . import lib
    status = [None] * ws
    dist.all_gather_object(status, failure)
    first = next((s for s in status if s is not None), None)
    if first is not None:
        raise lib.SdError(*first)
    return status


def _copy_into(src, dst, offset):
    """Copy the file `src` into the existing file `dst` at byte `offset` (ranks do this concurrently)."""
    n = os.path.getsize(src)
    with open(src, "rb") as fi, open(dst, "r+b") as fo:
        done = 0
        use_cfr = hasattr(os, "copy_file_range")
        while done < n:
            if use_cfr:
                try:
                    k = os.copy_file_range(fi.fileno(), fo.fileno(), min(n - done, 1 << 30), done, offset + done)
                except OSError:
                    use_cfr, k = False, 0
                if k:
                    done += k
                    continue
                use_cfr = False
            buf = os.pread(fi.fileno(), min(n - done, 8 << 20), done)
            if not buf:
                break
            os.pwrite(fo.fileno(), buf, offset + done)
            done += len(buf)
    if done != n:
        raise IOError("short copy of %s" % src)


def _assemble_parts(dist, rank, ws, outs, parts):
    """Every rank holds one part file per output: sizes are exchanged, rank 0 creates the outputs at their final
    size, then all ranks copy their parts in at their offsets.  Every step that can fail locally reports through the
    status exchange (raised on every rank)."""
    sizes, failure = [None] * ws, None
    try:
        mine = [os.path.getsize(p) for p in parts]
    except Exception as e:
        mine, failure = None, _status_of(e)
    dist.all_gather_object(sizes, mine)
    if failure is None and rank == 0 and all(sz is not None for sz in sizes):
        try:
            for k, p in enumerate(outs):
                with open(p, "wb") as f:
                    f.truncate(sum(sz[k] for sz in sizes))
        except Exception as e:
            failure = _status_of(e)
    _raise_first(dist, ws, failure)     # also the barrier in front of the copies
    try:
        for k, p in enumerate(outs):
            _copy_into(parts[k], p, sum(sz[k] for sz in sizes[:rank]))
    except Exception as e:
        failure = _status_of(e)
    _raise_first(dist, ws, failure)


def convert_sharded(raw_tsv, reads_fa, monomers_fa, final_out, alt_out, dist, conv_fn=None, **params):
    """convert_tsv (main.py:168-184) of a raw TSV that rank 0 has written, by ALL ranks: every rank converts the rows
    that begin in its byte range of the raw file (sd_convert_raw_tsv_range: ranges cut at line starts; rows are
    independent) into part files, which are then copied into the final / _alt files at their offsets.  The job of
    one huge sequence (chunk-range sharding) no longer leaves its whole post-processing to rank 0.  A failure on any
    rank is raised on every rank; no .partN file stays behind.  Returns True."""
    from . import lib
    rank, local_rank, ws = world()
    fn = conv_fn or lib.convert_raw_tsv_range
    outs = [final_out, alt_out]
    parts = ["%s.part%d" % (p, rank) for p in outs]
    failure = None
    try:
        try:
            fn(raw_tsv, reads_fa, monomers_fa, parts[0], parts[1], rank, ws, **params)
        except Exception as e:
            failure = _status_of(e)
        _raise_first(dist, ws, failure)
        _assemble_parts(dist, rank, ws, outs, parts)
        return True
    finally:
        for p in parts:
            try:
                os.remove(p)
            except OSError:
                pass


def run_files_sharded(reads_fa, monomers_fa, raw_out, final_out, alt_out, dist, run_fn=None, **params):
    """The multi-process command line on a set of reads: the reads are dealt to the ranks in contiguous groups of
    about equal chunk counts and EVERY rank runs its group completely on its GPU -- DP, identities, raw / final /
    _alt TSV text (sd_run_files_range) -- into part files; the parts are then copied, all ranks at once, into
    the three output files at their offsets.  No rank does another rank's post-processing and nothing but file
    sizes crosses between processes.  Returns True on every rank, or the string "unsplittable" (nothing
    written) when one read holds more than half a rank's share -- a single chromosome -- and the job has to be
    sharded by chunk range instead (decompose_files_sharded).  A failure on any rank is raised on every rank
    (the lowest failing rank's: the first offending read in file order)."""
    from . import lib
    rank, local_rank, ws = world()
    params = dict(params, device=params.get("device", local_rank))
    fn = run_fn or lib.run_files_range
    outs = [raw_out, final_out, alt_out]
    parts = ["%s.part%d" % (p, rank) for p in outs]
    failure = None
    try:
        try:
            fn(reads_fa, monomers_fa, rank, ws, *parts, **params)
        except Exception as e:   # not only SdError: a rank that raised past the exchange would leave the others waiting
            failure = _status_of(e)
        status = [None] * ws
        dist.all_gather_object(status, failure)
        first = next((s for s in status if s is not None), None)
        if first is not None:
            if all(s is not None and s[0] == lib.SD_ERR_UNSUPPORTED for s in status):
                return "unsplittable"
            raise lib.SdError(*first)
        # every later step that can fail locally (stat, truncate, copy) reports through the same exchange
        _assemble_parts(dist, rank, ws, outs, parts)
        return True
    finally:
        for p in parts:   # whatever happened, no .partN file stays behind
            try:
                os.remove(p)
            except OSError:
                pass
