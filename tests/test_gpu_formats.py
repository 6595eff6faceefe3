"""SURVEY 8(f) rank 4 on the GPU: the files the HIP path writes, read back through the on-disk formats API
(stringdecomposer_amd/formats.py; reference: main.py:168-184 reads the raw TSV, README.md:77-83 is the column spec of
the final TSV) and compared with the goldens of the reference binary / the unmodified reference command line; and the
binary record stream (<out>_raw.sdr) written by the same run.  `pytest -m gpu`."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT, load_case

from stringdecomposer_amd import formats, lib

pytestmark = pytest.mark.gpu


def _sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("name", ["td_light", "td_second_best", "td_second_best_i95", "syn64_second_best", "tiled_second_best"])
def test_cli_files_parse_and_re_emit_through_the_formats_api(name, tmp_path):
    """bin/stringdecomposer ... --records on a golden job of the unmodified reference command line: each of the three
    text files parses with formats.read_* and re-emits byte for byte (= the golden), the starred _alt rows are the
    final rows, and the record stream gives the raw TSV back -- through the Python reader and through the C-ABI."""
    d = os.path.join(GOLDEN, "final", name)
    with open(os.path.join(d, "params.json")) as f:
        c = json.load(f)
    out = str(tmp_path / "o")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "stringdecomposer")] +
                       [os.path.join(GOLDEN, x) for x in c["inputs"]] + ["-o", out, "-t", "8", "--records"] + c["args"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    fn = lambda x: os.path.join(out, "final_decomposition" + x)   # noqa: E731
    rd = lambda x: open(fn(x), "rb").read()                       # noqa: E731
    # raw TSV: parse -> re-emit -> the reference binary's bytes
    raw = formats.read_raw(fn("_raw.tsv"))
    raw_text = formats.format_raw(raw).encode()
    assert raw_text == rd("_raw.tsv") and _sha(raw_text) == c["raw_sha256"]
    for read, rr in formats.by_read(raw):   # the derived columns are SaveBatch's (main.cpp:277-283)
        assert formats.raw_rows(read, [(r.monomer, r.start, r.end, r.score) for r in rr]) == rr
    # final TSV: parse -> re-emit -> the reference command line's golden
    fin = formats.read_final(fn(".tsv"))
    with open(os.path.join(d, "final.tsv"), "rb") as f:
        gold_final = f.read()
    assert formats.format_final(fin).encode() == gold_final == rd(".tsv")
    assert len(fin) == c["final_rows"]
    # _alt TSV
    alt = formats.read_alt(fn("_alt.tsv"))
    alt_text = formats.format_alt(alt).encode()
    assert alt_text == rd("_alt.tsv") and _sha(alt_text) == c["alt_sha256"] and len(alt_text) == c["alt_bytes"]
    gz = os.path.join(d, "alt.tsv.gz")
    if os.path.exists(gz):
        with gzip.open(gz, "rb") as f:
            assert f.read() == alt_text
    if "--second-best" in c["args"]:
        starred = [(a.read, a.monomer, a.start, a.end, a.identity) for a in alt if a.best]
        assert starred == [(r.read, r.monomer, r.start, r.end, r.identity) for r in fin]
    # the binary record stream of the same run
    rec = formats.read_records(fn("_raw.sdr"))
    assert formats.records_to_raw_tsv(rec).encode() == raw_text
    names = lib.fasta_load(os.path.join(GOLDEN, c["inputs"][1]))[0]
    assert rec.templates == names + [n + "'" for n in names]
    assert rec.scoring == (-1, -1, -1, 1) and rec.ed_thr == -1
    rnames, rseqs, _ = lib.fasta_load(os.path.join(GOLDEN, c["inputs"][0]))
    assert [(r[0], r[1]) for r in rec.reads] == [(n, len(s)) for n, s in zip(rnames, rseqs)]
    lib.records_to_raw_tsv(fn("_raw.sdr"), str(tmp_path / "back.tsv"), threads=4)
    with open(str(tmp_path / "back.tsv"), "rb") as f:
        assert f.read() == raw_text
    # the finite final rows agree with the raw rows on the columns both hold
    keep = {(r.read, r.start, r.end) for r in fin}
    assert keep <= {(r.read, r.start, r.end) for r in raw}


@pytest.mark.parametrize("name", ["td_default", "td_part700_ov100", "td_s_-2_-3_-4_2", "td_edthr_10", "syn12_boundary_lengths",
                                  "syn12_N_multiline", "syn64_10kb", "weird_templates"])
def test_record_stream_from_the_device_equals_the_reference_binary(name, tmp_path):
    """sd_decompose_files_records (DP on the device -> record stream, no text anywhere) against the fixtures of the
    reference binary: records_to_raw_tsv of the stream is the golden raw TSV; header fields are the job's argv."""
    c = load_case(name)
    sc = tuple(c["scoring"]) if c["scoring"] else (-1, -1, -1, 1)
    ed = -1 if c["ed_thr"] is None else c["ed_thr"]
    p = str(tmp_path / "a.sdr")
    lib.decompose_files_records(c["reads"], c["monomers"], p, scoring=sc, part_size=c["part"], overlap=c["overlap"],
                                ed_thr=ed, threads=4)
    rec = formats.read_records(p)
    assert formats.records_to_raw_tsv(rec).encode() == c["raw"]
    assert (rec.scoring, rec.part_size, rec.overlap, rec.ed_thr) == (sc, c["part"], c["overlap"], ed)
    n = lib.read_records(p)
    assert int(n["row_off"][-1]) == len(n["rows"]) == c["rows"]
    out = str(tmp_path / "a.tsv")
    lib.records_to_raw_tsv(p, out, threads=2)
    with open(out, "rb") as f:
        assert f.read() == c["raw"]


def test_record_stream_of_a_job_cut_into_many_device_batches(tmp_path):
    """Reads that span device batches: the stream is written read by read as batches complete (sd_run_files_records
    with a small row budget) and still equals the one-shot raw TSV."""
    from stringdecomposer_amd import synth
    mn, ms = synth.make_monomers(12, seed=7)
    rn, rs = synth.make_reads(ms, 9, read_len=23000, seed=8)
    rfa, mfa = str(tmp_path / "r.fa"), str(tmp_path / "m.fa")
    synth.write_fasta(rfa, rn, rs, width=70)
    synth.write_fasta(mfa, mn, ms)
    o = [str(tmp_path / x) for x in ("raw.tsv", "final.tsv", "alt.tsv", "raw.sdr")]
    lib.run_files(rfa, mfa, o[0], o[1], o[2], threads=4, max_batch_rows=12000, records_out=o[3])
    with open(o[0], "rb") as f:
        raw = f.read()
    assert raw == lib.decompose(rn, rs, mn, ms, threads=4)
    rec = formats.read_records(o[3])
    assert formats.records_to_raw_tsv(rec).encode() == raw
    assert [r[0] for r in rec.reads] == [x.decode() if isinstance(x, bytes) else x for x in rn]
