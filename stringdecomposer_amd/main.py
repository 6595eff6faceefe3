#!/usr/bin/env python3
"""Drop-in command line of StringDecomposer on MI355X.

Mirrors the reference driver stringdecomposer/main.py (same positional arguments, options,
defaults, output files and log line), with the one process boundary of the reference --
`subprocess.run([SD_BIN, ...], stdout=raw_file)` at main.py:194 -- replaced by a ctypes call into
libsd_hip.so (HIP kernels for gfx950).  Post-processing of the raw TSV into
final_decomposition.tsv / _alt.tsv follows main.py:29-184 without Bio / python-edlib / pandas.

Documented difference: the reference CLI silently ignores -s/--scoring (it always launches the
binary with 10 arguments, and main.cpp:381 parses scores only when argc == 10, i.e. 9 arguments).
Here -s is honoured (the documented intent, README / main.py:210); pass --ref-compat to reproduce
the reference CLI's effective behaviour (default scores whatever -s says).

The command line runs natively end to end (lib.run_files -> sd_run_files): FASTA ingest, DP, raw TSV,
identities (device kernel) and the final / _alt TSVs are produced batch by batch inside the library.
convert_read / print_read / convert_tsv below are the same post-processing written on numpy arrays:
the module-level API of the reference, kept importable and used by the tests as a second implementation.
"""
import argparse
import logging
import os
import pathlib
import sys



class _LazyNumpy:
    """numpy is only needed by the module-level convert_* API (the Python implementation of the post-processing,
    kept for the tests and for callers of the reference's functions); the command line runs natively and should
    not pay ~0.1-0.2 s of import time for it."""

    def __getattr__(self, name):
        import numpy
        globals()["np"] = numpy
        return getattr(numpy, name)


np = _LazyNumpy()

from . import lib
from . import shard

CUR_DIR = os.path.dirname(os.path.abspath(__file__))
LOGREG_FILE = os.path.join(CUR_DIR, "models", "ont_logreg_model.txt")
# main.py:25-26 -- intercept, coef(identity), coef(identity - second best)
DEFAULT_LR = [-31.48494996, 0.41784018, 0.69186882]


def _lr_coef():
    if os.path.isfile(LOGREG_FILE):
        with open(LOGREG_FILE) as f:
            return list(map(float, f.readline().strip().split()))
    return list(DEFAULT_LR)


def get_logger(filename, logger_name="StringDecomposer", level=logging.INFO, filemode="a", stdout=True):
    """py/standard_logger.py:5-28."""
    logger = logging.getLogger(logger_name)
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    fh = logging.FileHandler(filename, mode=filemode)
    formatter = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    fh.setFormatter(formatter)
    logger.addHandler(fh)
    if stdout:
        sh = logging.StreamHandler(sys.stdout)
        sh.setFormatter(formatter)
        logger.addHandler(sh)
    return logger


class Record:
    """The three SeqRecord attributes main.py uses: .id/.name (first header token) and .seq."""
    __slots__ = ("id", "name", "seq")

    def __init__(self, rid, seq, name=None):
        self.id = rid
        self.name = name if name is not None else rid
        self.seq = seq


def read_fasta(filename):
    """Bio.SeqIO.parse(filename, 'fasta') + .upper() (main.py:63-74): id = first word of the
    title, sequence = remaining lines with whitespace removed, upper-cased."""
    recs = []
    name, chunks = None, []
    with open(filename) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    recs.append(Record(name, "".join(chunks).upper()))
                title = line[1:].rstrip()
                name = title.split(None, 1)[0] if title.split() else ""
                chunks = []
            elif name is not None:
                chunks.append("".join(line.split()))
    if name is not None:
        recs.append(Record(name, "".join(chunks).upper()))
    return recs


def load_fasta(filename, tp="list"):
    recs = read_fasta(filename)
    if tp == "map":
        d = {}
        for r in recs:
            if r.id in d:
                raise ValueError("Duplicate key '%s'" % r.id)
            d[r.id] = r
        return d
    return recs


_COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def add_rc_monomers(monomers):
    """main.py:81-86: interleaved m0, m0', m1, m1', ..."""
    res = []
    for m in monomers:
        res.append(m)
        res.append(Record(m.id + "'", m.seq.translate(_COMP)[::-1], m.name + "'"))
    return res


def convert_to_homo(seq):
    """main.py:87-92: homopolymer compression."""
    res = []
    prev = None
    for c in seq:
        if c != prev:
            res.append(c)
            prev = c
    return "".join(res)


def _strip_star(p):
    return p[:-1] if p.endswith("*") else p


def aai_batch(pairs, threads):
    """main.py:38-60 for a list of (query, target): percent identity from the NW alignment."""
    q = [_strip_star(a) for a, _ in pairs]
    t = [_strip_star(b) for _, b in pairs]
    res = lib.nw_identity_batch(q, t, threads=threads)
    out = []
    for ed, matches, cols in res:
        if ed == -1:
            out.append(0)
        else:
            a = 0.0
            a += matches       # sum of the '=' run lengths (integers; exact in a double)
            a /= cols
            out.append(a * 100)
    return out


def _identity_percent(dist, matches, cols):
    """main.py:38-60 on arrays: 0 for an empty side (edist -1), else '=' columns / CIGAR columns * 100."""
    with np.errstate(divide="ignore", invalid="ignore"):
        a = matches.astype(np.float64) / cols
        a = a * 100
    return np.where(dist == -1, 0.0, a)


def classify(score, second_best_score, coef):
    """main.py:95-104 (pandas DataFrame.dot == numpy dot over [1, score, score - second_best]):
    True where the block keeps '+', False where it becomes '?'."""
    s = np.asarray(score, dtype=np.float64)
    s2 = np.asarray(second_best_score, dtype=np.float64)
    if s.size == 0:
        return np.zeros(0, dtype=bool)
    X = np.stack([np.ones_like(s), s, s - s2], axis=1)
    return X.dot(np.array(coef, dtype=np.float64)) > 0


def _identities(seq, starts, ends, monomers, light, threads, own_names=None):
    """The NW identities convert_read needs for the blocks [starts, ends] of `seq`:
    light -> (scores,) against each block's own monomer; else -> (vals, hvals) against all monomers,
    plain and homopolymer-compressed."""
    seqs = [m.seq for m in monomers]
    if light:
        by_name = {}
        for x, m in enumerate(monomers):
            by_name[m.name] = x  # the reference keeps the last monomer of a given name
        pair = np.array([by_name[nm] for nm in own_names], dtype=np.int32)
        return (_identity_percent(*lib.identity_segments(seq, starts, ends, seqs, False, threads, pair_tmpl=pair)),)
    return (_identity_percent(*lib.identity_segments(seq, starts, ends, seqs, False, threads)),
            _identity_percent(*lib.identity_segments(seq, starts, ends, seqs, True, threads)))


def _convert_read_arrays(decomposition, read, monomers, light, threads, coef, pre=None):
    """main.py:107-150 on arrays: one entry per block.  Returns a dict of columns (lists / arrays).
    `pre` = this read's slice of identities computed for a batch of reads (see convert_tsv)."""
    n = len(decomposition)
    starts = np.array([d["start"] for d in decomposition], dtype=np.int64)
    ends = np.array([d["end"] for d in decomposition], dtype=np.int64)
    own_names = [d["m"] for d in decomposition]
    col = {"m": own_names, "start": starts, "end": ends, "alt_keys": None}
    if pre is None:
        pre = _identities(read.seq, starts, ends, monomers, light, threads, own_names)
    if light:
        col["score"] = pre[0]
        col["second_best"] = ["None"] * n
        col["second_best_score"] = np.full(n, -1.0)
        col["homo_best"] = ["None"] * n
        col["homo_best_score"] = np.full(n, -1.0)
        col["homo_second_best"] = ["None"] * n
        col["homo_second_best_score"] = np.full(n, -1.0)
    else:
        names = [m.name for m in monomers]
        vals, hvals = pre
        # scores is a dict keyed by monomer name in the reference: a repeated name keeps its first
        # position and its last value
        first, last = {}, {}
        for x, nm in enumerate(names):
            first.setdefault(nm, x)
            last[nm] = x
        keys = list(first)
        kidx = {k: x for x, k in enumerate(keys)}
        kcol = np.array([last[k] for k in keys], dtype=np.int64)
        kvals = vals[:, kcol]                                  # [n, len(keys)] in dict order
        for nm in own_names:
            if nm not in kidx:
                raise KeyError(nm)
        own = np.array([kidx[nm] for nm in own_names], dtype=np.int64)
        rows = np.arange(n)
        masked = kvals.copy()
        masked[rows, own] = -np.inf
        if len(keys) > 1 and "" not in kidx:
            sb = np.argmax(masked, axis=1)                     # first maximum among the other names
            col["second_best"] = [keys[x] for x in sb.tolist()]
            col["second_best_score"] = masked[rows, sb]
        else:  # `not secondbest` quirk of main.py:127 for an empty name / a single key
            sbn, sbs = [], []
            for i in range(n):
                secondbest, secondbest_score = None, -1
                for kx, m in enumerate(keys):
                    if kx != own[i]:
                        if not secondbest or secondbest_score < kvals[i, kx]:
                            secondbest, secondbest_score = m, float(kvals[i, kx])
                sbn.append(str(secondbest))
                sbs.append(secondbest_score)
            col["second_best"] = sbn
            col["second_best_score"] = np.array(sbs, dtype=np.float64)
        horder = np.argsort(-hvals, axis=1, kind="stable")
        h0, h1 = horder[:, 0], horder[:, 1]
        col["score"] = kvals[rows, own]
        col["homo_best"] = [names[x] for x in h0.tolist()]
        col["homo_best_score"] = hvals[rows, h0]
        col["homo_second_best"] = [names[x] for x in h1.tolist()]
        col["homo_second_best_score"] = hvals[rows, h1]
        col["alt_keys"], col["alt_vals"], col["own_key"] = keys, kvals, own
    keep = classify(col["score"], col["second_best_score"], coef)
    col["q"] = ["+" if k else "?" for k in keep.tolist()]
    return col


def convert_read(decomposition, read, monomers, light, threads, coef):
    """main.py:107-150.  Returns the list of per-block dicts of the reference (scores as floats)."""
    if not decomposition:
        return []
    c = _convert_read_arrays(decomposition, read, monomers, light, threads, coef)
    res = []
    for i in range(len(decomposition)):
        res.append({"m": c["m"][i], "start": str(int(c["start"][i])), "end": str(int(c["end"][i])),
                    "score": float(c["score"][i]),
                    "second_best": c["second_best"][i],
                    "second_best_score": -1 if light else float(c["second_best_score"][i]),
                    "homo_best": c["homo_best"][i], "homo_best_score": -1 if light else float(c["homo_best_score"][i]),
                    "homo_second_best": c["homo_second_best"][i],
                    "homo_second_best_score": -1 if light else float(c["homo_second_best_score"][i]),
                    "alt": {} if c["alt_keys"] is None else (c["alt_keys"], c["alt_vals"][i]), "q": c["q"][i]})
    return res


def print_read(fout, fout_alt, dec, read, monomers, identity_th, light, threads, coef, pre=None, alt_acc=None):
    """main.py:153-165.  With alt_acc (a list) the _alt rows are not written but appended to it as
    (read name, keys, starts, ends, own key, values) for one formatting call per batch of reads."""
    if not dec:
        return
    c = _convert_read_arrays(dec, read, monomers, light, threads, coef, pre)
    keep = np.nonzero(c["score"] >= identity_th)[0]
    f2 = "{:.2f}".format
    ks = keep.tolist()
    sc = [f2(x) for x in c["score"][keep].tolist()]
    s2 = [f2(x) for x in c["second_best_score"][keep].tolist()]
    h1 = [f2(x) for x in c["homo_best_score"][keep].tolist()]
    h2 = [f2(x) for x in c["homo_second_best_score"][keep].tolist()]
    st = c["start"][keep].tolist()
    en = c["end"][keep].tolist()
    name = read.name
    m, sbn, hb, hsb, q = c["m"], c["second_best"], c["homo_best"], c["homo_second_best"], c["q"]
    fout.write("".join(["%s\t%s\t%d\t%d\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" %
                        (name, m[i], st[j], en[j], sc[j], sbn[i], s2[j], hb[i], h1[j], hsb[i], h2[j], q[i])
                        for j, i in enumerate(ks)]))
    if c["alt_keys"] is not None and len(ks) and alt_acc is not None:
        alt_acc.append((name, c["alt_keys"], c["start"][keep], c["end"][keep], c["own_key"][keep], c["alt_vals"][keep]))
    elif c["alt_keys"] is not None and len(ks):  # one row per block and monomer name, formatted natively
        fout_alt.write(lib.format_alt_rows(name, c["alt_keys"], c["start"][keep], c["end"][keep],
                                           c["own_key"][keep], c["alt_vals"][keep],
                                           max(1, min(threads, len(ks) // 512))))


def convert_tsv(decomposition, reads, monomers, outfile, identity_th, light, threads=1):
    """main.py:168-184.  Reads are post-processed in the order of the raw file; the NW identities are
    computed for batches of reads at a time (one native call per ~64 k blocks instead of one per read)."""
    coef = _lr_coef()
    per_read = []            # [(read name, [block dicts])] in file order
    prev_read = None
    for ln in decomposition.split("\n")[:-1]:
        read, monomer, start, end = ln.split("\t")[:4]
        read = read.split()[0]
        monomer = monomer.split()[0]
        if read != prev_read:
            per_read.append((read, []))
        prev_read = read
        per_read[-1][1].append({"m": monomer, "start": int(start), "end": int(end)})
    with open(outfile[:-len(".tsv")] + "_alt.tsv", "w") as fout_alt:
        with open(outfile, "w") as fout:
            i = 0
            while i < len(per_read):
                j, blocks = i, 0
                while j < len(per_read) and (j == i or blocks + len(per_read[j][1]) <= 65536):
                    blocks += len(per_read[j][1])
                    j += 1
                batch = per_read[i:j]
                # one sequence for the batch: blocks never cross a read, so shifted coordinates are exact
                seqs, off, pos = [], [], 0
                for name, _ in batch:
                    sq = reads[name].seq
                    off.append(pos)
                    seqs.append(sq)
                    pos += len(sq)
                starts = np.array([d["start"] + o for (_, dec), o in zip(batch, off) for d in dec], dtype=np.int64)
                ends = np.array([d["end"] + o for (_, dec), o in zip(batch, off) for d in dec], dtype=np.int64)
                own = [d["m"] for _, dec in batch for d in dec]
                pre = _identities("".join(seqs), starts, ends, monomers, light, threads, own)
                at = 0
                alt_acc = []
                for name, dec in batch:
                    sl = slice(at, at + len(dec))
                    print_read(fout, fout_alt, dec, reads[name], monomers, identity_th, light, threads, coef,
                               tuple(p[sl] for p in pre), alt_acc)
                    at += len(dec)
                if alt_acc:  # the key list is the same for every read (it depends on the monomers only)
                    fout_alt.write(lib.format_alt_rows(
                        [a[0] for a in alt_acc], alt_acc[0][1], np.concatenate([a[2] for a in alt_acc]),
                        np.concatenate([a[3] for a in alt_acc]), np.concatenate([a[4] for a in alt_acc]),
                        np.concatenate([a[5] for a in alt_acc]), threads,
                        row_read=np.concatenate([np.full(len(a[2]), x, dtype=np.int32) for x, a in enumerate(alt_acc)])))
                i = j


def run(sequences, monomers, num_threads, scoring, batch_size, raw_file, ed_thr, overlap, logger,
        ref_compat=False, device=0, kernel=0, final_file=None, min_identity=0, second_best=False, records_file=None):
    """main.py:186-197 with the subprocess replaced by libsd_hip.so.

    Single process with final_file given: ONE native call (sd_run_files) streams the job through the
    device and writes the raw, final and _alt TSVs batch by batch -- nothing is re-read, returns True.
    Otherwise (a multi-GPU launch, or no final_file): writes the raw TSV and returns its text (rank 0)."""
    ins, dels, mm, match = [int(x) for x in scoring.split(",")]
    if ref_compat:
        ins, dels, mm, match = -1, -1, -1, 1  # what the reference binary does with 10 argv (main.cpp:381)
    try:
        lib.load()
    except lib.SdError as e:
        logger.info("The HIP library of String Decomposer is not available. Did you forget to run `make`? Aborting. (%s)" % e.msg)
        sys.exit(1)
    logger.info(" ".join(["Run", lib.LIB_PATH, "with parameters", sequences, monomers, str(num_threads),
                          str(batch_size), str(overlap), scoring]))
    rank, local_rank, ws = shard.world()
    if ws > 1:
        # launched with `python -m torch.distributed.run --nproc-per-node G bin/stringdecomposer ...`:
        # one process per GPU, each takes a contiguous range of the global chunk table (shard.py);
        # every rank writes the rows of its own range into the output (shard._assemble_by_ranks) -- no device collective.
        dist = shard.init_process_group("gloo")
        dev = local_rank % max(lib.device_count(), 1)
        common = dict(scoring=(ins, dels, mm, match), part_size=int(batch_size), overlap=int(overlap), ed_thr=int(ed_thr),
                      threads=int(num_threads), kernel=kernel, device=dev, flags=lib.FLAG_PROGRESS)
        try:
            ok = None
            if final_file is not None:
                # a set of reads: every rank runs its group of reads completely (DP + post-processing + text)
                ok = shard.run_files_sharded(sequences, monomers, raw_file, final_file,
                                             final_file[:-len(".tsv")] + "_alt.tsv", dist, min_identity=min_identity,
                                             second_best=second_best, lr_coef=_lr_coef(), **common)
            if ok is True:
                return True if rank == 0 else None
            # a single huge sequence (or no final file wanted): shard by chunk range; every rank makes the raw TSV text
            # of its own range (seam merge across the range boundaries from exchanged edges) and writes it at its offset
            asm_stats = {}
            ok = shard.decompose_files_sharded(sequences, monomers, raw_file, dist, assemble_stats=asm_stats, **common)
            if asm_stats:
                logger.info("raw TSV: this rank wrote %d bytes of it (merge + text %.1f ms before, %.1f ms after the exchange of "
                            "edges, write %.1f ms)" % (asm_stats["text_bytes"], asm_stats["begin_ms"], asm_stats["text_ms"],
                                                       asm_stats["write_ms"]))
            if final_file is not None:
                # ... and convert_tsv (main.py:168-184) is shared again: every rank converts the rows of its byte range of
                # the raw file (identities on its own GPU), the parts are copied into the final / _alt files
                shard.barrier(dist)   # the raw file is complete
                shard.convert_sharded(raw_file, sequences, monomers, final_file, final_file[:-len(".tsv")] + "_alt.tsv", dist,
                                      min_identity=min_identity, second_best=second_best, lr_coef=_lr_coef(), device=dev,
                                      threads=max(1, int(num_threads)))
                return True if rank == 0 else None
        finally:
            shard.barrier(dist)
            dist.destroy_process_group()
        return "raw file written" if ok else None
    if final_file is not None:
        lib.run_files(sequences, monomers, raw_file, final_file, final_file[:-len(".tsv")] + "_alt.tsv",
                      min_identity=min_identity, second_best=second_best, lr_coef=_lr_coef(),
                      scoring=(ins, dels, mm, match), part_size=int(batch_size), overlap=int(overlap),
                      ed_thr=int(ed_thr), threads=int(num_threads), device=device, kernel=kernel,
                      flags=lib.FLAG_PROGRESS,   # the dp binary's progress lines on stderr (main.cpp:82,115,393)
                      records_out=records_file)
        return True
    lib.decompose_files(sequences, monomers, raw_file, scoring=(ins, dels, mm, match),
                        part_size=int(batch_size), overlap=int(overlap), ed_thr=int(ed_thr),
                        threads=int(num_threads), device=device, kernel=kernel)
    with open(raw_file, "r") as f:
        raw_decomposition = "".join(f.readlines())
    return raw_decomposition


def main(argv=None):
    parser = argparse.ArgumentParser(description="Decomposes string into blocks alphabet")
    parser.add_argument("sequences", help="fasta-file with long reads or genomic sequences")
    parser.add_argument("monomers", help="fasta-file with monomers")
    parser.add_argument("-t", "--threads", help="number of threads (by default 1)", default="1", required=False)
    parser.add_argument("-o", "--out-dir", help="output directory (by default .)", default=".", required=False)
    parser.add_argument("--out-file", help='output tsv-file (by default "final_decomposition")',
                        default="final_decomposition", required=False)
    parser.add_argument("-i", "--min-identity",
                        help="only monomer alignments with percent identity >= MIN_IDENTITY are printed (by default MIN_IDENTITY=0)",
                        type=int, default=0, required=False)
    parser.add_argument("-s", "--scoring",
                        help='set scoring scheme for SD in the format "insertion,deletion,mismatch,match" (by default "-1,-1,-1,1")',
                        default="-1,-1,-1,1", required=False)
    parser.add_argument("-b", "--batch-size", help="set size of the batch in parallelization (by default 5000)",
                        type=str, default="5000", required=False)
    parser.add_argument("--second-best", dest="second_best",
                        help="generate second best monomer and homopolymer scores", action="store_true")
    parser.add_argument("--ed_thr",
                        help="align only monomers with edit distance less then ed_thr for each segment (by default align all monomers)",
                        default=-1, type=int, required=False)
    parser.add_argument("-v", "--overlap", help="set size of batch overlap (by default 500)", type=str,
                        default="500", required=False)
    # opt-in extras of this build (never change defaults)
    parser.add_argument("--ref-compat", action="store_true",
                        help="reproduce the reference CLI exactly: -s/--scoring is ignored (default scores)")
    parser.add_argument("--device", type=int, default=0, help="HIP device ordinal (by default 0)")
    parser.add_argument("--kernel", choices=["auto", "generic", "fast"], default="auto",
                        help="device kernel family (by default auto)")
    parser.add_argument("--records", action="store_true",
                        help="also write <out-file>_raw.sdr: the rows of the raw tsv as a binary record stream "
                             "(stringdecomposer_amd.formats.read_records)")
    args = parser.parse_args(argv)
    pathlib.Path(args.out_dir).mkdir(parents=True, exist_ok=True)

    logfn = os.path.join(args.out_dir, "stringdecomposer.log")
    if shard.world()[0] == 0:
        logger = get_logger(logfn, logger_name="StringDecomposer")
    else:  # ranks > 0 of a multi-GPU launch compute their share and stay silent
        logger = logging.getLogger("StringDecomposer.rank%d" % shard.world()[0])
        logger.addHandler(logging.NullHandler())
        logger.propagate = False
    logger.info(f"cmd: {sys.argv}")

    raw_decomp_fn = os.path.join(args.out_dir, args.out_file + "_raw.tsv")
    convert_tsv_fn = os.path.join(args.out_dir, args.out_file + ".tsv")
    records_fn = os.path.join(args.out_dir, args.out_file + "_raw.sdr") if args.records else None
    kernel = {"auto": 0, "generic": 1, "fast": 2}[args.kernel]
    try:
        raw_decomposition = run(args.sequences, args.monomers, args.threads, args.scoring, args.batch_size,
                                raw_decomp_fn, args.ed_thr, args.overlap, logger,
                                ref_compat=args.ref_compat, device=args.device, kernel=kernel,
                                final_file=convert_tsv_fn, min_identity=int(args.min_identity),
                                second_best=args.second_best,
                                records_file=records_fn if shard.world()[2] == 1 else None)
    except lib.SdError as e:
        # the reference dies with CalledProcessError after the binary printed its message on stderr
        sys.stderr.write(e.msg + "\n")
        logger.info("String Decomposer failed: " + e.msg)
        sys.exit(e.code if 0 < e.code < 256 else 1)
    if raw_decomposition is None:
        return  # not rank 0
    logger.info("Saved raw decomposition to " + raw_decomp_fn)
    if records_fn and shard.world()[2] > 1:
        # a multi-GPU launch: the ranks wrote text parts; rank 0 restates the finished raw file as the record stream
        from . import formats
        names = lib.fasta_load(args.monomers)[0]
        sc = (-1, -1, -1, 1) if args.ref_compat else tuple(int(x) for x in args.scoring.split(","))
        formats.write_records(records_fn, formats.raw_to_records(
            formats.read_raw(raw_decomp_fn), names + [n + "'" for n in names], scoring=sc,
            part_size=int(args.batch_size), overlap=int(args.overlap), ed_thr=int(args.ed_thr)))
    if records_fn:
        logger.info("Saved the binary record stream to " + records_fn)
    logger.info("Transforming raw alignments...")
    if raw_decomposition is not True:
        # multi-GPU launch: rank 0 holds the raw TSV of the whole job; convert_tsv (main.py:168-184) natively,
        # identities on this rank's GPU
        try:
            lib.convert_raw_tsv(raw_decomp_fn, args.sequences, args.monomers, convert_tsv_fn,
                                convert_tsv_fn[:-len(".tsv")] + "_alt.tsv", int(args.min_identity), args.second_best,
                                _lr_coef(), device=shard.world()[1] % max(lib.device_count(), 1),
                                threads=max(1, int(args.threads)))
        except lib.SdError as e:
            sys.stderr.write("post-processing failed: " + e.msg + "\n")
            logger.info("Transformation failed (%s); the raw decomposition is in %s" % (e.msg, raw_decomp_fn))
            sys.exit(e.code if 0 < e.code < 256 else 1)
    logger.info("Transformation finished. Results can be found in " + convert_tsv_fn)

    logger.info("Thank you for using StringDecomposer!")


if __name__ == "__main__":
    main()
