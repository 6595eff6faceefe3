"""Readers / writers of the three on-disk TSV formats (stringdecomposer_amd/formats.py): round trips on
the reference's own golden file and on the raw fixtures produced by the reference binary."""
import os

import pytest

from conftest import GOLDEN, case_names, load_case

from stringdecomposer_amd import formats
from stringdecomposer_amd import main as sdmain


@pytest.mark.parametrize("name", case_names())
def test_raw_round_trip_on_reference_fixtures(name):
    text = load_case(name)["raw"].decode()
    rows = formats.parse_raw(text)
    assert formats.format_raw(rows) == text
    # derived columns follow SaveBatch: gap = start - previous end of the same read (0 before the first)
    for read, rr in formats.by_read(rows):
        rebuilt = formats.raw_rows(read, [(r.monomer, r.start, r.end, r.score) for r in rr])
        assert rebuilt == rr


def test_final_round_trip_on_reference_golden():
    path = os.path.join(GOLDEN, "test_data", "final_decomposition_fc89af8.tsv")
    rows = formats.read_final(path)
    with open(path) as f:
        assert formats.format_final(rows) == f.read()
    assert len(rows) == 557 and sum(r.reliability == "?" for r in rows) == 2
    assert all(r.end >= r.start and 0 <= r.identity <= 100 for r in rows)


def test_alt_round_trip_and_consistency_with_final(tmp_path):
    c = load_case("td_default")
    td = os.path.join(GOLDEN, "test_data")
    reads = sdmain.load_fasta(os.path.join(td, "read.fa"), "map")
    mons = sdmain.add_rc_monomers(sdmain.load_fasta(os.path.join(td, "DXZ1_star_monomers.fa")))
    out = str(tmp_path / "d.tsv")
    sdmain.convert_tsv(c["raw"].decode(), reads, mons, out, 0, False, threads=2)
    alt = formats.read_alt(out[:-4] + "_alt.tsv")
    with open(out[:-4] + "_alt.tsv") as f:
        assert formats.format_alt(alt) == f.read()
    fin = formats.read_final(out)
    starred = [a for a in alt if a.best]
    assert [(a.read, a.monomer, a.start, a.end, a.identity) for a in starred] == \
           [(r.read, r.monomer, r.start, r.end, r.identity) for r in fin]


def test_malformed_lines_are_reported_with_position():
    with pytest.raises(formats.FormatError) as e:
        formats.parse_raw("r\tm\t1\t2\t3.000000\t1\n")
    assert ":1:" in str(e.value)
    with pytest.raises(formats.FormatError):
        formats.parse_alt("r\tm\t1\t2\t99.00\tx\n")
    assert formats.parse_final("") == []
