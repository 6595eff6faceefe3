"""CPU-only checks of the product library: it loads, exports the whole C-ABI of include/sd_hip.h,
and its host-side pieces (chunk plan, seam merge, TSV formatting, FASTA ingest, NW identity,
final-TSV post-processing) agree with the oracle / the reference's golden files.
No device compute happens here."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import CASES, GOLDEN, ROOT, case_names, load_case

from stringdecomposer_amd import lib, main as sdmain, synth


def test_library_loads_and_exports_header_symbols():
    L = lib.load()
    with open(os.path.join(ROOT, "include", "sd_hip.h")) as f:
        hdr = f.read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(lib.EXPORTS)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert b"gfx950" in L.sd_version()


def test_no_cpu_fallback_without_device():
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    mn, ms = synth.make_monomers(12, seed=1)
    with pytest.raises(lib.SdError) as e:
        lib.Engine(ms)
    assert e.value.code == lib.SD_ERR_NO_DEVICE
    with pytest.raises(lib.SdError) as e:
        lib.decompose(["r"], [b"ACGT" * 10], mn, ms)
    assert e.value.code == lib.SD_ERR_NO_DEVICE


def test_param_validation_is_host_side():
    mn, ms = synth.make_monomers(2, seed=1)
    with pytest.raises(lib.SdError) as e:
        lib.Engine(ms, part_size=0)
    assert e.value.code == lib.SD_ERR_PARAM
    with pytest.raises(lib.SdError) as e:
        lib.Engine([b"ACGTX"])
    assert e.value.code == lib.SD_ERR_SYMBOL
    with pytest.raises(lib.SdError) as e:
        lib.Engine(ms, scoring=(-1000, -1, -1, 1))  # reaches the reference's INF sentinel
    assert e.value.code == lib.SD_ERR_UNSUPPORTED


@pytest.mark.parametrize("length", [0, 1, 2, 499, 500, 501, 4999, 5000, 5001, 5499, 5500, 5501,
                                    10499, 10500, 50000, 94871])
@pytest.mark.parametrize("po", [(5000, 500), (700, 100), (333, 77), (100, 0), (10, 500)])
def test_chunk_plan_matches_oracle(oracle, length, po):
    assert lib.chunk_plan(length, *po) == oracle.chunk_plan(length, *po)


def test_seam_merge_matches_oracle(oracle):
    st = synth.Stream(5, 5)
    for trial in range(200):
        n = int(st.below(1, 40)[0])
        recs, pos = [], 0
        for _ in range(n):
            a = pos + int(st.below(1, 200)[0]) - 60
            b = a + int(st.below(1, 260)[0])
            recs.append((int(st.below(1, 24)[0]), a, b, int(st.below(1, 300)[0]) - 100))
            pos = b
        got = lib.seam_merge(recs)
        exp = [(t, s, e, int(sc)) for (t, s, e, sc) in oracle.postprocess([(t, s, e, float(sc)) for t, s, e, sc in recs])]
        assert got == exp


def test_format_rows_matches_reference_text():
    c = load_case("td_default")
    lines = c["raw"].decode().splitlines()
    read_name = lines[0].split("\t")[0]
    names = sorted({ln.split("\t")[1] for ln in lines})
    rows = []
    for ln in lines:
        f = ln.split("\t")
        rows.append((names.index(f[1]), int(f[2]), int(f[3]), int(float(f[4]))))
    assert lib.format_rows(read_name, names, rows) == c["raw"]
    # negative scores / zero
    txt = lib.format_rows("r", ["m"], [(0, 0, 0, -166), (0, 1, 5, 0)])
    assert txt == b"r\tm\t0\t0\t-166.000000\t0\t0\nr\tm\t1\t5\t0.000000\t1\t4\n"


def test_fasta_load_semantics(tmp_path):
    p = tmp_path / "a.fa"
    p.write_bytes(b">r1 some description\nACGT\n\nNNAC\n>r2\tx\nGG\n")
    names, seqs, has_n = lib.fasta_load(str(p))
    assert names == ["r1", "r2"] and seqs == [b"ACGTNNAC", b"GG"] and has_n
    for name in case_names(include_errors=True):
        if not name.startswith("err_"):
            continue
        c = load_case(name)
        with pytest.raises(lib.SdError) as e:
            lib.fasta_load(c["reads"])
        assert e.value.code == lib.SD_ERR_SYMBOL
        assert e.value.msg.strip() == c["stderr_tail"][0].strip()
    with pytest.raises(lib.SdError) as e:
        lib.fasta_load(str(tmp_path / "missing.fa"))
    assert e.value.code == lib.SD_ERR_IO


def _rand_seq(st, n):
    return bytes(b"ACGT"[i] for i in st.below(n, 4))


def test_nw_identity_matches_oracle_and_edlib(oracle):
    st = synth.Stream(9, 9)
    qs, ts = [], []
    for trial in range(300):
        L = int(st.below(1, 230)[0]) + 1
        t = synth._ACGT[st.below(L, 4)].tobytes()
        mode = trial % 4
        if mode == 0:
            codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer(t, dtype=np.uint8))
            q = synth._to_ascii(synth.mutate(codes, st, 0.1, 0.05, 0.05)) or b"A"
        elif mode == 1:
            q = synth._ACGT[st.below(int(st.below(1, 300)[0]) + 1, 4)].tobytes()   # unrelated
        elif mode == 2:
            q = t[: max(1, L // 2)]
        else:
            q = t[::-1]
        qs.append(q)
        ts.append(t)
    qs += [b"", b"ACGT", b"A" * 700, b"ACGT" * 40]
    ts += [b"ACGT", b"", b"A" * 170, b"TGCA" * 40]
    # bit-vector word boundaries, N symbols, a symbol outside ACGTN (full-matrix path), long pairs
    for L in (1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 1000):
        a = synth._ACGT[st.below(L, 4)].tobytes()
        b = synth._ACGT[st.below(L + 3, 4)].tobytes()
        qs += [a, a, a[:-1] + b"N", a[: L // 2] + b"R" + a[L // 2:]]
        ts += [b, a, a, a]
    got = lib.nw_identity_batch(qs, ts, threads=3)
    for q, t, g in zip(qs, ts, got):
        assert g == oracle.nw_identity(q, t), (q, t)
    if os.path.isfile(oracle.REF_EDLIB):
        ed = ctypes.CDLL(oracle.REF_EDLIB)

        class Cfg(ctypes.Structure):
            _fields_ = [("k", ctypes.c_int), ("mode", ctypes.c_int), ("task", ctypes.c_int),
                        ("eq", ctypes.c_void_p), ("neq", ctypes.c_int)]

        class Res(ctypes.Structure):
            _fields_ = [("status", ctypes.c_int), ("editDistance", ctypes.c_int),
                        ("endLocations", ctypes.POINTER(ctypes.c_int)),
                        ("startLocations", ctypes.POINTER(ctypes.c_int)),
                        ("numLocations", ctypes.c_int),
                        ("alignment", ctypes.POINTER(ctypes.c_ubyte)),
                        ("alignmentLength", ctypes.c_int), ("alphabetLength", ctypes.c_int)]

        ed.edlibAlign.restype = Res
        ed.edlibAlign.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, Cfg]
        ed.edlibFreeAlignResult.argtypes = [Res]
        for q, t, g in zip(qs, ts, got):
            if not q or not t:
                continue
            r = ed.edlibAlign(q, len(q), t, len(t), Cfg(-1, 0, 2, None, 0))   # NW, path
            m = sum(1 for i in range(r.alignmentLength) if r.alignment[i] == 0)
            assert (r.editDistance, m, r.alignmentLength) == g
            ed.edlibFreeAlignResult(r)


def _edlib_nw(ed):
    class Cfg(ctypes.Structure):
        _fields_ = [("k", ctypes.c_int), ("mode", ctypes.c_int), ("task", ctypes.c_int),
                    ("eq", ctypes.c_void_p), ("neq", ctypes.c_int)]

    class Res(ctypes.Structure):
        _fields_ = [("status", ctypes.c_int), ("editDistance", ctypes.c_int),
                    ("endLocations", ctypes.POINTER(ctypes.c_int)),
                    ("startLocations", ctypes.POINTER(ctypes.c_int)),
                    ("numLocations", ctypes.c_int),
                    ("alignment", ctypes.POINTER(ctypes.c_ubyte)),
                    ("alignmentLength", ctypes.c_int), ("alphabetLength", ctypes.c_int)]

    ed.edlibAlign.restype = Res
    ed.edlibAlign.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, Cfg]
    ed.edlibFreeAlignResult.argtypes = [Res]

    def align(q, t):
        r = ed.edlibAlign(q, len(q), t, len(t), Cfg(-1, 0, 2, None, 0))   # NW, path
        a = np.ctypeslib.as_array(r.alignment, shape=(r.alignmentLength,))
        out = (r.editDistance, int((a == 0).sum()), r.alignmentLength)
        ed.edlibFreeAlignResult(r)
        return out
    return align


def test_nw_identity_of_long_pairs_follows_edlibs_hirschberg_split(oracle):
    """Once (2*8+4) * ceil(|query| / 64) * |target| + 8 * |target| reaches 1 MB edlib aligns by Hirschberg's split of the
    target instead of its block traceback (edlib.cpp:1186-1400; ~19.6 kb against a 171-bp monomer, reachable with
    -b >= 19000): the host identities (csrc/sd_post.hip) and the oracle restate both; checked against each other and,
    where oracle/_ref holds it, against the reference's vendored edlib itself.  Also symbols outside ACGTN and a
    100-kb query (the 65 000-bp limit of round 2 is gone)."""
    st = synth.Stream(11, 3)

    def rnd(n, k=4):
        return synth._ACGT[st.below(n, k)].tobytes()
    qs, ts = [], []
    for trial in range(16):
        tl = [171, 600, 2000, 90][trial % 4]
        t = rnd(tl, 4 if trial % 5 else 2)
        ql = [20000, 6000, 3000, 70000][trial % 4] + int(st.below(1, 3000)[0])
        mode = (trial // 4) % 4
        if mode == 0:
            q = rnd(ql)
        elif mode == 1:
            codes = np.searchsorted(synth._ACGT, np.frombuffer(t * (ql // tl + 1), dtype=np.uint8))[:ql]
            q = synth._to_ascii(synth.mutate(codes, st, 0.1, 0.05, 0.05))
        elif mode == 2:
            q = rnd(ql, 2)
        else:
            q = rnd(ql // 2) + t + rnd(ql // 2, 3)
        assert 20 * ((len(q) + 63) // 64) * len(t) + 8 * len(t) >= 1 << 20
        qs.append(q)
        ts.append(t)
    qs += [b"ACGTRYK" * 30, b"\x01\x02\x03ACGT" * 20, b"A" * 100000, b"N" * 21000 + b"ACGT"]
    ts += [b"ACGTRRK" * 28, b"\x03\x02\x01TGCA" * 22, b"A" * 170 + b"C", rnd(171)]
    got = lib.nw_identity_batch(qs, ts, threads=4)
    for q, t, g in zip(qs[:8] + qs[16:], ts[:8] + ts[16:], got[:8] + got[16:]):
        assert tuple(g) == tuple(oracle.nw_identity(q, t)), (len(q), len(t))
    if os.path.isfile(oracle.REF_EDLIB):
        align = _edlib_nw(ctypes.CDLL(oracle.REF_EDLIB))
        for q, t, g in zip(qs, ts, got):
            assert align(q, t) == tuple(g), (len(q), len(t))


def test_final_tsv_postprocessing_reproduces_reference_golden(tmp_path):
    """The reference's own golden file (made with --second-best, reference Makefile:17-19)."""
    c = load_case("td_default")
    td = os.path.join(GOLDEN, "test_data")
    reads = sdmain.load_fasta(os.path.join(td, "read.fa"), "map")
    mons = sdmain.add_rc_monomers(sdmain.load_fasta(os.path.join(td, "DXZ1_star_monomers.fa")))
    out = str(tmp_path / "final_decomposition.tsv")
    sdmain.convert_tsv(c["raw"].decode(), reads, mons, out, 0, False, threads=4)
    with open(out, "rb") as f, open(os.path.join(td, "final_decomposition_fc89af8.tsv"), "rb") as g:
        assert f.read() == g.read()
    # the _alt file (main.py:161-165): one row per block x monomer, recomputed here with the oracle's NW
    oracle_b = __import__("oracle.binding", fromlist=["x"])
    with open(out[:-4] + "_alt.tsv") as f:
        alt = [ln.rstrip("\n").split("\t") for ln in f]
    with open(out) as f:
        main_rows = [ln.rstrip("\n").split("\t") for ln in f]
    assert len(alt) == len(main_rows) * len(mons)
    rseq = next(iter(reads.values())).seq
    for bi in (0, 1, len(main_rows) // 2, len(main_rows) - 1):
        row = main_rows[bi]
        for x, m in enumerate(mons):
            a = alt[bi * len(mons) + x]
            ed, mt, cols = oracle_b.nw_identity(rseq[int(row[2]):int(row[3]) + 1], m.seq)
            assert a[:4] == [row[0], m.name, row[2], row[3]]
            assert a[4] == "{:.2f}".format(mt / cols * 100)
            assert a[5] == ("*" if m.name == row[1] else "-")
    # light mode: sha recorded from the unmodified reference CLI
    sdmain.convert_tsv(c["raw"].decode(), reads, mons, out, 0, True, threads=2)
    with open(out, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == \
            "de9d4cc554051d842022f0a75db18a7d4ce7b3aa97af9cadc6ef960c7ea8ee85"
    assert os.path.getsize(out[:-4] + "_alt.tsv") == 0


def test_convert_read_dict_semantics_with_repeated_names(oracle):
    """main.py:118-146 keeps per-name dicts: a repeated monomer name keeps its first position and
    last value; second best = first maximum among the other names; homo list is a stable sort."""
    st = synth.Stream(5, 77)
    base = synth._ACGT[st.below(60, 4)].tobytes().decode()
    def mut(sq, k):
        b = list(sq)
        for p in st.below(k, len(b)):
            b[int(p)] = "ACGT"[int(st.below(1, 4)[0])]
        return "".join(b)
    mons = [sdmain.Record("a", mut(base, 3)), sdmain.Record("b", mut(base, 6)),
            sdmain.Record("a", mut(base, 9)), sdmain.Record("c", mut(base, 6))]
    mons = sdmain.add_rc_monomers(mons)
    read = sdmain.Record("r", "".join(mut(base, 5) for _ in range(6)))
    dec = [{"m": ["a", "b", "c", "a'", "b", "c"][i], "start": 60 * i, "end": 60 * i + 59} for i in range(6)]
    got = sdmain.convert_read(dec, read, mons, False, 2, sdmain._lr_coef())

    def ident(q, t):
        ed, m, c = oracle.nw_identity(q, t)
        return 0 if ed == -1 else m / c * 100

    for d, g in zip(dec, got):
        seg = read.seq[d["start"]:d["end"] + 1]
        scores = {}
        for m in mons:
            scores[m.name] = ident(seg, m.seq)
        sb, sbs = None, -1
        for m in scores:
            if m != d["m"]:
                if not sb or sbs < scores[m]:
                    sb, sbs = m, scores[m]
        hs = sorted([[m.name, ident(sdmain.convert_to_homo(seg), sdmain.convert_to_homo(m.seq))]
                     for m in mons], key=lambda x: -x[1])
        assert g["score"] == scores[d["m"]] and g["second_best"] == str(sb) and g["second_best_score"] == sbs
        assert [g["homo_best"], g["homo_best_score"]] == hs[0]
        assert [g["homo_second_best"], g["homo_second_best_score"]] == hs[1]
        keys, row = g["alt"]
        assert list(keys) == list(scores) and row.tolist() == [scores[k] for k in keys]


def test_native_alt_row_formatting_equals_python_format():
    """sd_format_alt_rows prints "%.2f"; Python prints "{:.2f}": both are the correctly rounded decimal of
    the double, also at the x.xx5 ties that a decimal reading would round the other way."""
    rng = np.random.default_rng(5)
    n, nk = 700, 5
    vals = rng.uniform(0, 100, size=(n, nk))
    vals[:50, 0] = np.arange(50) + 0.125            # exact ties in binary: round-half-even
    vals[50:100, 1] = np.arange(50) + 0.005         # inexact: whichever side the double falls on
    vals[100:150, 2] = (np.arange(50) * 7 + 1) / 3 * 100 / 57
    vals[150, 3] = 0.0
    vals[151, 3] = 100.0
    keys = ["m%d" % k if k != 3 else "m3'" for k in range(nk)]
    starts = np.arange(n) * 171
    ends = starts + 170
    own = rng.integers(0, nk, size=n)
    got = lib.format_alt_rows("read/1", keys, starts, ends, own, vals, threads=3)
    exp = "".join("read/1\t%s\t%d\t%d\t%s\t%s\n" % (keys[k], starts[i], ends[i], "{:.2f}".format(vals[i, k]),
                                                   "*" if k == own[i] else "-")
                  for i in range(n) for k in range(nk))
    assert got == exp
    assert lib.format_alt_rows("r", keys, [], [], [], np.zeros((0, nk)), threads=2) == ""


def test_pack_bases_matches_the_device_format():
    """2-bit packing (SWAR, 8 bases per 64-bit step) against a plain restatement of the format the
    kernels read: base i at bits 2*(i & 15) of word i >> 4, A,C,G,T = 0..3, N = 0 + mask bit."""
    st = synth.Stream(8, 8)
    for n in [1, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 1000, 5500]:
        for with_n in (False, True):
            b = bytearray(synth._ACGT[st.below(n, 4)].tobytes())
            if with_n:
                for p in st.below(max(1, n // 9), n):
                    b[int(p)] = ord("N")
            w, m, has_n = lib.pack_bases(bytes(b))
            code = {65: 0, 67: 1, 71: 2, 84: 3, 78: 0}
            ew = np.zeros((n + 15) // 16, dtype=np.uint32)
            em = np.zeros((n + 31) // 32, dtype=np.uint32)
            for i, ch in enumerate(b):
                ew[i >> 4] |= np.uint32(code[ch] << (2 * (i & 15)))
                if ch == 78:
                    em[i >> 5] |= np.uint32(1 << (i & 31))
            assert (w == ew).all() and (m == em).all() and has_n == (b"N" in bytes(b)), (n, with_n)


def test_host_pool_parallel_loops_are_exact():
    """The persistent host thread pool behind the library's parallel loops (packing, assembly, TSV
    formatting): many short loops in a row, from several Python threads at once (ctypes releases the
    GIL; the pool runs one loop at a time), still assemble every read."""
    from concurrent.futures import ThreadPoolExecutor
    n = 700
    names = ["read%d" % i for i in range(n)]
    lens = [100 + (i % 50) for i in range(n)]
    recs = np.zeros(2 * n, dtype=lib._rec_dtype())
    recs["tmpl"] = np.arange(2 * n) % 6
    recs["start"] = np.tile([0, 40], n)
    recs["end"] = np.tile([39, 99], n)
    recs["score"] = np.arange(2 * n) - 50
    off = np.arange(n + 1, dtype=np.int64) * 2
    mn = ["a", "b", "c"]
    exp = lib.assemble_tsv(names, lens, mn, recs, off, threads=1)
    assert exp.count(b"\n") == 2 * n

    def one(_):
        return lib.assemble_tsv(names, lens, mn, recs, off, threads=6)

    with ThreadPoolExecutor(4) as ex:
        assert all(x == exp for x in ex.map(one, range(40)))


def test_native_convert_raw_tsv_reproduces_reference_golden(tmp_path):
    """sd_convert_raw_tsv = convert_tsv (main.py:168-184) as one native call (host identities here): the
    reference's golden final TSV, the _alt file of the Python implementation, the light-mode sha of the
    unmodified reference CLI, -i filtering, and a monomer file with a repeated name."""
    c = load_case("td_default")
    td = os.path.join(GOLDEN, "test_data")
    rfa, mfa = os.path.join(td, "read.fa"), os.path.join(td, "DXZ1_star_monomers.fa")
    raw = str(tmp_path / "raw.tsv")
    with open(raw, "wb") as f:
        f.write(c["raw"])
    fin, alt = str(tmp_path / "n.tsv"), str(tmp_path / "n_alt.tsv")
    lib.convert_raw_tsv(raw, rfa, mfa, fin, alt, 0, True, device=-1, threads=4)
    with open(fin, "rb") as f, open(os.path.join(td, "final_decomposition_fc89af8.tsv"), "rb") as g:
        assert f.read() == g.read()
    reads = sdmain.load_fasta(rfa, "map")
    mons = sdmain.add_rc_monomers(sdmain.load_fasta(mfa))
    pout = str(tmp_path / "p.tsv")
    sdmain.convert_tsv(c["raw"].decode(), reads, mons, pout, 0, False, threads=4)
    with open(alt, "rb") as f, open(pout[:-4] + "_alt.tsv", "rb") as g:
        assert f.read() == g.read()
    lib.convert_raw_tsv(raw, rfa, mfa, fin, alt, 0, False, device=-1, threads=3)
    with open(fin, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == "de9d4cc554051d842022f0a75db18a7d4ce7b3aa97af9cadc6ef960c7ea8ee85"
    assert os.path.getsize(alt) == 0
    for th in (85, 95):
        lib.convert_raw_tsv(raw, rfa, mfa, fin, alt, th, True, device=-1, threads=2)
        sdmain.convert_tsv(c["raw"].decode(), reads, mons, pout, th, False, threads=2)
        for a, b in ((fin, pout), (alt, pout[:-4] + "_alt.tsv")):
            with open(a, "rb") as f, open(b, "rb") as g:
                assert f.read() == g.read()
    # repeated monomer name + several reads + a read that comes back later in the raw file
    st = synth.Stream(6, 66)
    base = synth._ACGT[st.below(80, 4)].tobytes()
    mseqs = [base, base[:40] + synth._ACGT[st.below(40, 4)].tobytes(), base[::-1], base[5:] + b"ACGTA"]
    mfa2, rfa2 = str(tmp_path / "m2.fa"), str(tmp_path / "r2.fa")
    synth.write_fasta(mfa2, ["a", "b", "a", "c"], mseqs)
    rseqs = [base * 3 + mseqs[1] * 2, mseqs[2] * 4, mseqs[3] + base]
    synth.write_fasta(rfa2, ["r0", "r1", "r2"], rseqs, width=50)
    lines = []
    for rn, n, mn in (("r0", 5, ["a", "a", "b'", "b", "c"]), ("r1", 4, ["a'", "a", "c'", "b"]), ("r0", 2, ["c", "a"]),
                      ("r2", 2, ["c", "a"])):
        for i in range(n):
            lines.append("%s\t%s\t%d\t%d\t10.000000\t0\t79\n" % (rn, mn[i], 80 * i, 80 * i + 79))
    raw2 = str(tmp_path / "raw2.tsv")
    with open(raw2, "w") as f:
        f.write("".join(lines))
    for sb in (True, False):
        lib.convert_raw_tsv(raw2, rfa2, mfa2, fin, alt, 0, sb, device=-1, threads=2)
        sdmain.convert_tsv("".join(lines), sdmain.load_fasta(rfa2, "map"), sdmain.add_rc_monomers(sdmain.load_fasta(mfa2)),
                           pout, 0, not sb, threads=2)
        for a, b in ((fin, pout), (alt, pout[:-4] + "_alt.tsv")):
            with open(a, "rb") as f, open(b, "rb") as g:
                assert f.read() == g.read(), (sb, a)
    with pytest.raises(lib.SdError):
        lib.convert_raw_tsv(raw2, rfa, mfa2, fin, alt, 0, True, device=-1)     # reads of the raw file missing


def test_fasta_file_mapped_loader_semantics(tmp_path):
    """sd::FastaFile (mmap + parallel index) behind sd_fasta_load: same records as the line-by-line loader
    semantics of main.cpp:314-346 for single-line, multi-line, blank-line and header-description input."""
    p = tmp_path / "a.fa"
    p.write_bytes(b"\n>r1 some description\nACGT\n\nNNAC\n>r2\tx\nGG\n>r3\nA\n>r4\nACGTACGT")
    names, seqs, has_n = lib.fasta_load(str(p))
    assert names == ["r1", "r2", "r3", "r4"] and seqs == [b"ACGTNNAC", b"GG", b"A", b"ACGTACGT"] and has_n
    big = tmp_path / "big.fa"
    mn, ms = synth.make_monomers(12, seed=2)
    rn, rs = synth.make_reads(ms, 40, read_len=30000, seed=2)
    synth.write_fasta(str(big), rn, rs, width=0)
    n2, s2, hn = lib.fasta_load(str(big))
    assert n2 == rn and s2 == rs and not hn
    synth.write_fasta(str(big), rn, rs, width=61)
    n2, s2, hn = lib.fasta_load(str(big))
    assert n2 == rn and s2 == rs
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b">x\nACGT\n>y\nACGu\nAC\n>z\nAXC\n")
    with pytest.raises(lib.SdError) as e:
        lib.fasta_load(str(bad))
    assert e.value.code == lib.SD_ERR_SYMBOL and e.value.msg == "ERROR: Sequence y contains undefined symbol (not ACGT): u"
    for content in (b"ACGT\n>x\nAC\n", b">\nACGT\n"):
        bad.write_bytes(content)
        with pytest.raises(lib.SdError) as e:
            lib.fasta_load(str(bad))
        assert e.value.code == lib.SD_ERR_FORMAT


def _host_rate_worker(k, threads, q):
    from stringdecomposer_amd import lib as L, synth as S
    mn, ms = S.make_monomers(12, seed=1)
    rn, rs = S.make_reads(ms, 40, read_len=50000, seed=100 + k)
    q.put((k, L.host_stage_rates(rs, iters=3, threads=threads)))


def test_eight_concurrent_host_pipelines_report_aggregate_rate(capsys):
    """the host side of an 8-GPU node must feed 8 pipelines at once.  Eight processes (one per
    would-be rank) run the host stages -- chunk table + 2-bit packing, per-read assembly + raw TSV text --
    concurrently, each with its share of the host threads; the aggregate bp/s is reported (on the 256-core
    GPU box: see profiles/; this container has 8 cores, so only sanity is asserted here)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    threads = max(1, (os.cpu_count() or 8) // 8)
    ps = [ctx.Process(target=_host_rate_worker, args=(k, threads, q)) for k in range(8)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300)[1] for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    pack = sum(r["pack_bp_per_s"] for r in res)
    asm = sum(r["assemble_format_bp_per_s"] for r in res)
    with capsys.disabled():
        print("\n[host stages, 8 processes x %d threads] packer %.2f Gbp/s, assembler+formatter %.2f Gbp/s aggregate"
              % (threads, pack / 1e9, asm / 1e9))
    assert all(r["text_bytes"] > 0 and r["rows_per_s"] > 0 for r in res)
    assert pack > 2e8 and asm > 2e8


FINAL = os.path.join(GOLDEN, "final")


def final_cases():
    return sorted(os.listdir(FINAL)) if os.path.isdir(FINAL) else []


def load_final_case(name):
    import json
    with open(os.path.join(FINAL, name, "params.json")) as f:
        meta = json.load(f)
    meta["reads"], meta["monomers"] = [os.path.join(GOLDEN, p) for p in meta["inputs"]]
    with open(os.path.join(FINAL, name, "final.tsv"), "rb") as f:
        meta["final"] = f.read()
    return meta


@pytest.mark.parametrize("name", final_cases())
def test_native_post_processing_vs_unmodified_reference_cli(name, tmp_path, oracle):
    """Final and _alt TSVs of the UNMODIFIED reference command line (tests/golden/make_final_golden.py: the
    reference's bin/stringdecomposer run here with its own dp binary and its vendored edlib): the raw rows of
    the oracle must hash to the reference's raw file, and sd_convert_raw_tsv (host identities) must reproduce
    the final TSV byte for byte and the _alt TSV by sha256 (the whole 2.67 MB file for the test data)."""
    c = load_final_case(name)
    a = c["args"]
    raw = oracle.decompose_files(c["reads"], c["monomers"], threads=8, part=int(a[a.index("-b") + 1]) if "-b" in a else 5000)
    assert hashlib.sha256(raw).hexdigest() == c["raw_sha256"]
    rawf, fin, alt = str(tmp_path / "raw.tsv"), str(tmp_path / "f.tsv"), str(tmp_path / "f_alt.tsv")
    with open(rawf, "wb") as f:
        f.write(raw)
    mi = int(a[a.index("-i") + 1]) if "-i" in a else 0
    lib.convert_raw_tsv(rawf, c["reads"], c["monomers"], fin, alt, mi, "--second-best" in a, device=-1, threads=8)
    with open(fin, "rb") as f:
        assert f.read() == c["final"]
    with open(alt, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == c["alt_sha256"]


def test_reference_import_path_is_an_alias():
    """north_star keeps the Python CLI `stringdecomposer.main`: the name resolves to this build's driver."""
    import stringdecomposer.main as ref_name
    assert ref_name.main is sdmain.main and ref_name.convert_tsv is sdmain.convert_tsv and ref_name.run is sdmain.run


def test_two_decimal_text_equals_python_format():
    """sd::put_fixed2 (csrc/sd_host.hpp) writes the identity columns of the final and _alt TSVs without printf; it
    has to be "{:.2f}".format(v) (main.py:157-165) for every double: exact ties, the nearest doubles of x.xx5,
    identities m / c * 100, random bit patterns, negative zero, subnormals, huge values."""
    import random
    import struct
    rnd = random.Random(5)
    vals = [i / 1000.0 for i in range(0, 30000)]
    vals += [i * 0.125 * j / 4 for i in range(1, 800) for j in (1, 3, 5, 7)]
    vals += [m / c * 100 for c in range(1, 300) for m in range(0, c + 1, 3)]
    vals += [struct.unpack("<d", struct.pack("<Q", rnd.getrandbits(64)))[0] for _ in range(60000)]
    vals += [rnd.uniform(-5, 105) for _ in range(30000)]
    vals += [0.0, -0.0, -1.0, 0.005, 0.015, 0.025, 0.035, 1e-300, 5e-324, -1e-9, 99.995, 100.0, 1e12 - 0.005,
             2.0 ** 39 + 0.125, 2.0 ** 40, 1e15, float("inf"), float("-inf")]
    v = np.array([x for x in vals if x == x], dtype=np.float64).reshape(-1, 1)
    n = v.shape[0]
    z = np.zeros(n, dtype=np.int64)
    txt = lib.format_alt_rows("r", ["k"], z, z, np.zeros(n, dtype=np.int32), v, threads=3)
    got = [line.split("\t")[4] for line in txt.split("\n")[:-1]]
    assert len(got) == n
    bad = [(x, g) for x, g in zip(v[:, 0].tolist(), got) if "{:.2f}".format(x) != g]
    assert not bad, bad[:5]


@pytest.mark.parametrize("sc,cells,rebase", [((-2, -1, -3, 6), "f16", 128), ((-2, -2, -3, 8), "f16", 128),
                                             # beyond the range at 128 rows between rebases: 64 rows, then integer cells
                                             ((-2, -2, -3, 9), "f16", 64), ((-2, -2, -3, 18), "f16", 64), ((-2, -2, -3, 33), "int16", 128),
                                             ((-2, -2, -3, 40), "int16", 128),
                                             ((-1, -1, -1, 1), "f16", 128), ((0, 0, -1, 14), "f16", 128), ((0, 0, -1, 15), "f16", 64),
                                             ((0, 0, -1, 31), "int16", 128), ((0, 0, -1, 50), "int16", 128),
                                             # a large |del| with non-positive scores: the range is (Lmax-1)*|del| wide
                                             ((0, -6, -4, -1), "f16", 128), ((0, -7, -4, -1), "f16", 64), ((0, -9, -4, -1), "int16", 128),
                                             ((0, -10, -4, -1), "int16", 128), ((-1, -5, -2, 1), "f16", 128),
                                             # a common factor is divided out on the device and multiplied back
                                             ((-10, -10, -10, 10), "f16", 128), ((-4, -6, -8, 4), "f16", 128), ((0, 0, -3, 45), "f16", 64),
                                             ((0, 0, -3, 150), "int16", 128)])
def test_plan_cell_format_switch(sc, cells, rebase):
    """sd_plan_info (host only): the fp16 / int16 decision of fast_plan_build for scorings on both sides of the
    exact-integer range of fp16 -- the same cases the GPU test runs against the oracle.  Since round 6 that decision is
    what FLAG_NO_U16 leaves (and what a set beyond the narrow layout gets); the narrow layout itself takes the biased-u16
    cells for all of them, at the full 128-row rebase period."""
    mn, ms = synth.make_monomers(12, seed=3)
    info = lib.plan_info(ms, scoring=sc, flags=lib.FLAG_NO_U16)
    assert info["family"] == "fast" and info["cells"] == cells and info["rebase"] == rebase, info
    info = lib.plan_info(ms, scoring=sc)
    assert info["family"] == "fast" and info["cells"] == "u16" and info["rebase"] == 128, info


@pytest.mark.parametrize("sc,cells", [((0, 0, -1, 90), "u16"),          # 129 * 91 + ... = 12 600: beyond saturating int16's 12 000
                                      ((0, 0, -1, 100), "u16"),         # 13 900 of the 15 000 the window leaves
                                      ((0, 0, -1, 115), "int32"),       # beyond it: generic family
                                      ((-3, -60, -4, 5), "int32"),      # (Lmax + 1) * |del| alone takes 10 500 of the window
                                      ((-30, -30, -40, 50), "u16")])
def test_plan_biased_u16_window(sc, cells):
    """The biased-u16 cells (CellOps<CF_U16>): the stored-cell bound of the 128-row period must fit what +-15 800 leaves after
    the end offsets ((Lmax + 1) * |del|) and a few single scores (FastPlan::u16_lim)."""
    mn, ms = synth.make_monomers(12, seed=3)
    info = lib.plan_info(ms, scoring=sc)
    assert info["cells"] == cells, info
    if cells == "u16":
        lmax = max(len(m) for m in ms)
        g = 1
        import math
        g = math.gcd(math.gcd(abs(sc[0]), abs(sc[1])), math.gcd(abs(sc[2]), abs(sc[3]))) or 1
        room = 15800 - (lmax + 1) * abs(sc[1] // g) - 8 * max(abs(x // g) for x in sc)
        assert info["range_bound"] <= room, (info, room)


def test_plan_layouts_of_the_baseline_sets_and_fuzz_fixtures():
    mn, ms = synth.make_monomers(12, seed=1)
    c2 = lib.plan_info(ms)
    assert (c2["cells_per_lane"], c2["cells"], c2["waves"]) == (35, "u16", 1) and c2["floor_slots"] <= 16, c2
    c2f = lib.plan_info(ms, flags=lib.FLAG_NO_U16)
    assert (c2f["cells_per_lane"], c2f["cells"], c2f["waves"]) == (35, "f16", 1), c2f
    mn, ms = synth.make_monomers(64, seed=1)
    c4 = lib.plan_info(ms)
    assert c4["cells"] == "f16/bf8-table" and c4["cells_per_lane"] == 176 and c4["floor_slots"] <= 32, c4
    mn, ms = synth.make_monomers(150, seed=1)
    big = lib.plan_info(ms)
    assert big["cells"] == "f16/bf8-codes x waves" and big["waves"] == 3, big
    one = lib.plan_info([b"A", b"ACGTACGT"])                                   # a 1-bp monomer: a lane of its own, ended at slot 0
    assert one["family"] == "fast" and one["trace_regs"] == 1, one
    mn, ms = synth.make_monomers(70, seed=1)
    w1 = lib.plan_info(list(ms) + [b"G"])                                      # in a set beyond one wave: the tiled layout knows the form too
    assert (w1["family"], w1["cells"]) == ("fast", "f16/bf8-codes tiled x waves"), w1
    assert lib.plan_info([b"ACGTACGT"], scoring=(-1, 1, -1, 1))["family"] == "generic"   # positive DELETION score: a deleted template gains
    assert lib.plan_info([b"ACGTACGT"], scoring=(1, -1, -1, 1))["family"] == "fast"      # positive insertion score: taken since round 6
    mn12, ms12 = synth.make_monomers(12, seed=1)
    pi = lib.plan_info(ms12, scoring=(2, -3, -4, 2))
    assert (pi["family"], pi["cells"]) == ("fast", "u16") and pi["range_bound"] >= 129 * 5, pi   # B may grow by smax - del = 5 a row
    # templates longer than the widest lane (224 slots) in a set beyond one wave of the narrow layout: tiled over the
    # virtual lanes of up to eight waves (sd_fast_wt.hip); the slot count with the least SIMD time per row at the occupancy it gets
    st = synth.Stream(5, 5)
    rnd = lambda n: synth._to_ascii(st.below(n, 4))
    t1 = lib.plan_info([rnd(330 + j % 20) for j in range(30)])                 # 60 templates of four 96-slot lanes: two waves, four
    assert (t1["family"], t1["cells"], t1["cells_per_lane"], t1["waves"]) == ("fast", "f16/bf8-codes tiled x waves", 96, 2), t1   # chunks per CU (one wave of 192 slots: six)
    t2 = lib.plan_info([rnd(400 + j) for j in range(100)])                     # 200 templates of two or three 224-slot lanes
    assert (t2["cells"], t2["cells_per_lane"], t2["waves"]) == ("f16/bf8-codes tiled x waves", 224, 4), t2
    t3 = lib.plan_info([rnd(1000)] * 5)                                        # ten templates of eleven lanes: one wave of 96 slots
    assert (t3["cells"], t3["cells_per_lane"], t3["waves"]) == ("f16/bf8-codes tiled x waves", 96, 1), t3
    t4 = lib.plan_info([rnd(1000)] * 5, scoring=(-1, -2, -1, 1))               # beyond the fp16 range: the tiled form with integer cells (round 5)
    assert (t4["family"], t4["cells"], t4["waves"]) == ("fast", "int16/int8-codes tiled x waves", 1), t4
    t5 = lib.plan_info([rnd(150 + j % 30) for j in range(150)], scoring=(-2, -2, -3, 40))   # 44 is not exact in bf8, B grows by 42 a row
    assert (t5["family"], t5["cells"], t5["waves"]) == ("fast", "int16/int8-codes x waves", 3), t5
    assert lib.plan_info([rnd(150 + j % 30) for j in range(150)], scoring=(-2, -2, -3, 90))["family"] == "generic"   # beyond int16 cells
    huge = [rnd(100 + j % 150) for j in range(600)]
    assert lib.plan_info(huge)["family"] == "generic"                           # more lanes than eight waves hold ...
    h2 = lib.plan_info(huge, ed_thr=30)                                         # ... but with --ed_thr a chunk's kept templates mostly fit:
    assert (h2["family"], h2["cells"], h2["waves"]) == ("fast", "f16/bf8-codes tiled x waves", 8), h2   # the filter-only form
    assert lib.plan_info(huge, ed_thr=30, flags=lib.FLAG_NO_EDTHR_COMPACT)["family"] == "generic"
    fz = os.path.join(GOLDEN, "fuzz")
    # the scoring that overran fp16 before the range bound charged B's growth (seed 906)
    mn, ms, _ = lib.fasta_load(os.path.join(fz, "fuzz_fail_906_791", "m.fa"))
    assert lib.plan_info(ms, scoring=(0, -4, -4, -1), flags=lib.FLAG_NO_U16)["cells"] == "int16"
    assert lib.plan_info(ms, scoring=(0, -4, -4, -1))["cells"] == "u16"
    # the 5-bp template whose first lane must keep two cells (seed 1301)
    mn, ms, _ = lib.fasta_load(os.path.join(fz, "fuzz_fail_1301_1135", "m.fa"))
    assert lib.plan_info(ms, scoring=(-9, -7, -8, 9))["min_first_lane_cells"] >= 2


def test_plan_lane_boundaries_properties():
    """The plan may give a lane fewer than P cells to keep the start-term maxima in few slots: lanes never exceed P,
    the first lane of a template keeps at least two cells, and the chosen layout never costs more cell ops per row
    (2 P + floor_slots) than the uniform one."""
    st = synth.Stream(11, 4)
    for trial in range(40):
        n = 1 + int(st.below(1, 22)[0])
        lo = [8, 40, 120, 160, 300][int(st.below(1, 5)[0])]
        ms = [synth._ACGT[st.below(lo + int(st.below(1, 40)[0]), [4, 2, 3][int(st.below(1, 3)[0])])].tobytes() for _ in range(n)]
        info = lib.plan_info(ms)
        if info["family"] != "fast" or info["cells_per_lane"] > 64:
            continue
        P = info["cells_per_lane"]
        assert 2 <= info["min_first_lane_cells"] and info["max_lane_cells"] <= P and 1 <= info["floor_slots"] < max(P, 2), info
        os.environ["SD_PLAN_UNIFORM_LANES"] = "1"
        try:
            uni = lib.plan_info(ms)
        finally:
            del os.environ["SD_PLAN_UNIFORM_LANES"]
        assert 2 * P + info["floor_slots"] <= 2 * uni["cells_per_lane"] + uni["floor_slots"] + 16, (info, uni)
        assert uni["max_lane_cells"] <= uni["cells_per_lane"]


def _stored_extrema(tmpls, read, sc, rebase=128):
    """Largest |stored cell| of the fast fills on one chunk, from the reference recurrence itself
    (main.cpp:171-207): S = E - base - tp*ins with E = D - k*del, base = the between-monomers score at the last
    rebase (every `rebase` rows; 0 before the first), tp = rows since then (sd_fast_fill.hpp)."""
    ins, dele, mis, mat = sc
    n = len(read)
    D = []
    for t in tmpls:
        row = np.empty(len(t), dtype=np.int64)
        for k in range(len(t)):
            s_ = mat if t[k] == read[0] else mis
            row[k] = s_ if k == 0 else max(row[k - 1] + dele, dele * (k - 1) + s_)
        D.append(row)
    kd = [np.arange(len(t), dtype=np.int64) * dele for t in tmpls]
    worst = max(int(np.abs(D[j] - kd[j]).max()) for j in range(len(tmpls)))     # row 0: base 0, tp 0
    base, r0 = 0, 0
    for i in range(1, n):
        B = max(int(d[-1]) for d in D)
        if i % rebase == 0:
            # state of row i-1 under the new base, tp = 0 (the guard's check_low) -- and under the old one (check_high)
            worst = max(worst, max(int(np.abs(D[j] - kd[j] - B).max()) for j in range(len(tmpls))))
            base, r0 = B, i
        nxt = []
        for j, t in enumerate(tmpls):
            s_ = np.where(np.frombuffer(t, dtype=np.uint8) == read[i], mat, mis).astype(np.int64)
            cand = B + s_ + kd[j]
            cand[1:] = np.maximum(cand[1:], np.maximum(D[j][:-1] + s_[1:], D[j][1:] + ins))
            # in-row deletion chain: prefix maximum of (cand - k*del) shifted back
            e = np.maximum.accumulate(cand - kd[j])
            nxt.append(e + kd[j])
        D = nxt
        tp = i - r0 + 1 if r0 else i
        worst = max(worst, max(int(np.abs(D[j] - kd[j] - base - tp * ins).max()) for j in range(len(tmpls))))
    return worst


@pytest.mark.parametrize("sc", [(-1, -1, -1, 1), (-2, -3, -4, 2), (0, -4, -4, -1), (-1, -5, -2, 3), (-3, -1, -6, 2),
                                (0, 0, -1, 15), (-2, -2, -3, 33), (0, -40, -4, -1)])   # the last three: 64 rows between rebases, or integer cells
def test_plan_range_bound_covers_the_recurrence(sc):
    """VERDICT r02 (weak 6): the fp16 fills rest on fast_plan_build's bound |stored cell| <= range_bound.  Here the
    stored values are computed from the reference recurrence in numpy for reads built to push them -- repeats of one
    template (B grows fastest), runs of a base absent from the templates (B falls), random reads -- over two rebase
    periods, and must stay inside the bound the plan reports (round 1's formula failed exactly this for 0,-4,-4,-1)."""
    st = synth.Stream(2024, sc[1] * 7 + sc[3])
    for trial in range(4):
        nm = 2 + trial
        ms = [synth._to_ascii(st.below(int(st.below(1, 40)[0]) + 12, 4)) for _ in range(nm)]
        info = lib.plan_info(ms, scoring=sc)
        if info["family"] != "fast":
            continue
        tm = ms + [synth.revcomp_bytes(m) for m in ms]
        reads = [(ms[0] * 40)[:300], (b"A" * 150 + ms[-1] * 20)[:300],
                 synth._to_ascii(st.below(300, 4)), (ms[0][: len(ms[0]) // 2] * 60)[:300]]
        for rd in reads:
            assert _stored_extrema(tm, rd, sc, rebase=info["rebase"]) <= info["range_bound"], (sc, trial, info)


def _trace_extrema(tmpls, read, sc, block=32):
    """Largest |X| of the packed two-block traceback (sd_fast_trace2.hip) on one chunk, from the reference recurrence:
    X = E - r*ins - base with E = D - k*del and base = the start term of the block's first computed row,
    B(rs) + del - (rs - 1)*ins (rs = max(block start, 1)); taken over the checkpoint row above a block and the block's 32
    rows, and over the start terms of those rows."""
    ins, dele, mis, mat = sc
    n = len(read)
    rows, Bs = [], [0]
    D = []
    for t in tmpls:
        row = np.empty(len(t), dtype=np.int64)
        for k in range(len(t)):
            s_ = mat if t[k] == read[0] else mis
            row[k] = s_ if k == 0 else max(row[k - 1] + dele, dele * (k - 1) + s_)
        D.append(row)
    kd = [np.arange(len(t), dtype=np.int64) * dele for t in tmpls]
    rows.append([D[j] - kd[j] for j in range(len(tmpls))])
    for i in range(1, n):
        B = max(int(d[-1]) for d in D)
        Bs.append(B)
        nxt = []
        for j, t in enumerate(tmpls):
            s_ = np.where(np.frombuffer(t, dtype=np.uint8) == read[i], mat, mis).astype(np.int64)
            cand = B + s_ + kd[j]
            cand[1:] = np.maximum(cand[1:], np.maximum(D[j][:-1] + s_[1:], D[j][1:] + ins))
            nxt.append(np.maximum.accumulate(cand - kd[j]) + kd[j])
        D = nxt
        rows.append([D[j] - kd[j] for j in range(len(tmpls))])
    worst = 0
    for ab in range(0, n, block):
        rs = max(ab, 1)
        if rs >= n:
            break
        base = Bs[rs] + dele - (rs - 1) * ins
        for r in range(max(ab - 1, 0), min(ab + block, n)):
            worst = max(worst, max(int(np.abs(e - r * ins - base).max()) for e in rows[r]))
            if r >= 1:
                worst = max(worst, abs(Bs[r] + dele - (r - 1) * ins - base))
    return worst


@pytest.mark.parametrize("sc", [(-1, -1, -1, 1), (-2, -3, -4, 2), (0, -4, -4, -1), (-1, -5, -2, 3), (-3, -1, -6, 2), (-9, -7, -8, 9)])
def test_plan_traceback_word_range_covers_the_recurrence(sc):
    """The packed two-block traceback keeps cells as 16-bit words 4*(E' - base) + tag + 0x4000, exact while
    |E' - base| stays below the bound the plan proves per template set and scoring (sd_plan_info: trace_bound; the
    kernel also checks every checkpoint cell and start term at run time).  Same reads as the fill's range test --
    repeats of one template, runs of an absent base, random sequence -- through the reference recurrence in numpy."""
    st = synth.Stream(4242, sc[1] * 5 + sc[3])
    seen = 0
    for trial in range(4):
        nm = 2 + trial
        ms = [synth._to_ascii(st.below(int(st.below(1, 60)[0]) + 12, 4)) for _ in range(nm)]
        info = lib.plan_info(ms, scoring=sc)
        if info["family"] != "fast" or info["trace_regs"] == 0:
            continue
        seen += 1
        assert 4 * info["trace_bound"] + 16 <= 15000
        tm = ms + [synth.revcomp_bytes(m) for m in ms]
        reads = [(ms[0] * 40)[:200], (b"A" * 90 + ms[-1] * 20)[:200],
                 synth._to_ascii(st.below(200, 4)), (ms[0][: len(ms[0]) // 2] * 60)[:200]]
        for rd in reads:
            g = info["score_factor"]
            scr = tuple(x // g for x in sc)   # the device computes with the scores divided by their common factor
            assert _trace_extrema(tm, rd, scr) <= info["trace_bound"], (sc, trial, info)
    assert seen or abs(sc[0]) >= 9   # (-9,-7,-8,9 leaves the 16-bit range on longer templates: the one-block form runs)


@pytest.mark.parametrize("fail_reserve", [False, True])
def test_file_writer_on_tmpfs_with_and_without_page_reservation(fail_reserve, tmp_path):
    """sd::write_parts (the TSV writer of sd_run_files): on tmpfs large texts go through a shared mapping, but only
    after fallocate has reserved the pages -- when that fails (a full /dev/shm) the text must take the pwritev
    loop (which reports ENOSPC as SD_ERR_IO) instead of a store into a sparse mapping (SIGBUS).  Both ways give
    the same bytes; a disk directory always takes pwritev."""
    shm = "/dev/shm"
    dirs = [str(tmp_path)]
    if os.path.isdir(shm) and os.access(shm, os.W_OK):
        free = os.statvfs(shm).f_bavail * os.statvfs(shm).f_frsize
        if free > (64 << 20):
            dirs.append(shm)
    for d in dirs:
        path = os.path.join(d, "sd_wp_selftest_%d_%d.bin" % (os.getpid(), int(fail_reserve)))
        try:
            # 2 x 12 parts x 512 KiB = 12 MiB: above the 4-MiB threshold of the mapped path
            n, ram = lib.write_parts_selftest(path, 12, 512 << 10, threads=4, fail_reserve=fail_reserve)
            assert n == 2 * 12 * (512 << 10)
            assert os.path.getsize(path) == n
            if d == shm:
                assert ram
            # small texts (below the threshold) and empty parts
            n2, _ = lib.write_parts_selftest(path, 3, 0, threads=2, fail_reserve=fail_reserve)
            assert n2 == 0 and os.path.getsize(path) == 0
        finally:
            if os.path.exists(path):
                os.unlink(path)


def test_alphabet_check_vector_path_names_the_first_bad_byte(tmp_path):
    """The alphabet check (main.cpp:329-341: the first symbol outside ACGTN of the first offending sequence, exit -1)
    runs 32 bytes per step where AVX2 is available and hands the offending stretch to the scalar loop: the message must
    name the same byte wherever it sits -- in a full vector, in the scalar tail, right behind a vector boundary -- for
    lower case, CR, NUL-free control bytes and bytes with the top bit set; N alone is not an error and is reported."""
    st = synth.Stream(77, 5)
    for n in (63, 64, 65, 96, 100, 333, 5000):
        base = bytearray(synth._ACGT[st.below(n, 4)].tobytes())
        ok = tmp_path / ("ok%d.fa" % n)
        ok.write_bytes(b">r\n" + bytes(base) + b"\n")
        assert lib.fasta_load(str(ok))[2] is False
        for pos in sorted({0, 1, 31, 32, 33, 62, 63, n // 2, n - 2, n - 1} & set(range(n))):
            for bad in (b"a", b"\r", b"\x01", b"\xc1", b"M", b"U"):
                b = bytearray(base)
                b[pos:pos + 1] = bad
                if pos + 40 < n:
                    b[pos + 40] = ord("z")          # a later offender must not be the one reported
                p = tmp_path / "bad.fa"
                p.write_bytes(b">r1\n" + bytes(b) + b"\n")
                with pytest.raises(lib.SdError) as ei:
                    lib.fasta_load(str(p))
                assert ei.value.code == lib.SD_ERR_SYMBOL
                assert ei.value.msg.encode(errors="surrogateescape").endswith(b"(not ACGT): " + bad) or \
                    ei.value.msg.endswith("(not ACGT): " + bad.decode(errors="replace")), (n, pos, bad, ei.value.msg)
            b = bytearray(base)
            b[pos] = ord("N")
            p = tmp_path / "n.fa"
            p.write_bytes(b">r1\n" + bytes(b) + b"\n")
            assert lib.fasta_load(str(p))[2] is True


def test_committed_pmc_profile_belongs_to_these_kernel_sources():
    """bench.py derives instructions per row and HBM traffic of the fill from the committed rocprofv3 counters
    (profiles/fill_traffic.json) and withholds them when the kernel sources have changed since they were taken: a kernel
    edit without a fresh profile fails HERE, before a bench line goes out without its roofline."""
    import json
    import bench
    with open(os.path.join(ROOT, "profiles", "fill_traffic.json")) as f:
        tj = json.load(f)
    assert tj.get("kernel_sources_sha256") == bench.kernel_source_hashes(), \
        "kernel sources changed: rerun tools/profile_r02.sh + tools/reduce_pmc.py on the GPU box"


def test_packed_traceback_applies_up_to_256_bp_templates():
    """sd_plan_info's public trace_regs: the packed two-block traceback takes templates of up to 64 * 4 = 256 bp
    (ceil(L / 64) registers per lane); one base more and the one-block int32 form runs (trace_regs == 0)."""
    import random
    rnd = random.Random(5)
    for L, regs in ((255, 4), (256, 4), (257, 0), (128, 2), (129, 3)):
        ms = ["".join(rnd.choice("ACGT") for _ in range(L)) for _ in range(3)]
        info = lib.plan_info(ms)
        assert info["family"] == "fast" and info["trace_regs"] == regs, (L, info)


def test_streamed_seam_merge_of_a_huge_read_equals_the_literal_scan():
    """A read of many chunks is merged and formatted as its chunks arrive (ReadAssembler::stream_advance: the scan of
    main.cpp:287-302 stopped eight records before the end of what has arrived); rows must equal the literal scan over the
    whole read -- records built so that every branch of the scan fires (overlaps of more than half a record at distances
    1..6, runs of them, the unchecked record behind a drop), small reads before and after the huge one."""
    import random
    import numpy as np
    rnd = random.Random(11)
    part, ov = 100, 30
    read_lens = [900, 700_000, 1300, 950_000, 40]          # 9 + 7 000 + 13 + 9 500 + 1 chunks: the huge reads arrive in several calls
    rec_dt = lib._rec_dtype()
    recs, off, per_read = [], [0], []
    for rl in read_lens:
        rows = []
        for (o, ln) in lib.chunk_plan(rl, part, ov):
            pos, k = 0, 0
            chunk = []
            while pos < ln - 5:
                w = rnd.randint(3, 40)
                e = min(ln - 1, pos + w)
                chunk.append((rnd.randrange(6), pos, e, rnd.randint(-5, 30)))
                # the next record often starts INSIDE this one (a seam-like overlap), sometimes far behind it
                pos = max(0, e - rnd.randint(0, w)) if rnd.random() < 0.45 else e + 1 + rnd.randint(0, 3)
                k += 1
                if k > 60:
                    break
            recs.extend(chunk)
            off.append(len(recs))
            rows.extend((t, s + o, e + o, sc) for t, s, e, sc in chunk)
        per_read.append(rows)

    def literal(b):   # main.cpp:287-302
        res, i = [], 0
        while i < len(b):
            for j in range(i + 1, min(i + 7, len(b))):
                if (b[i][2] - b[j][1]) * 2 > (b[j][2] - b[j][1]):
                    res.append(b[i])
                    i = j + 1
                    break
            if i < len(b):
                res.append(b[i])
            i += 1
        return res
    names = ["r%d" % i for i in range(len(read_lens))]
    mono = ["m0", "m1", "m2"]
    tn = mono + [m + "'" for m in mono]
    want = []
    for nm, rows in zip(names, per_read):
        prev = 0
        for t, s, e, sc in literal(rows):
            want.append("%s\t%s\t%d\t%d\t%d.000000\t%d\t%d\n" % (nm, tn[t], s, e, sc, s - prev, e - s))
            prev = e
    arr = np.array(recs, dtype=rec_dt)
    for threads in (1, 4):
        got = lib.assemble_tsv(names, read_lens, mono, arr, np.array(off, dtype=np.int64), part_size=part, overlap=ov, threads=threads)
        assert got.decode() == "".join(want)


def _seam_records(rnd, read_lens, part, ov, p_overlap, wmax=40):
    """Per-chunk records whose neighbours overlap with probability p_overlap (every branch of main.cpp:287-302 fires)."""
    recs, off = [], [0]
    for rl in read_lens:
        for (_o, ln) in lib.chunk_plan(rl, part, ov):
            pos, k = 0, 0
            while pos < ln - 5 and k <= 60:
                w = rnd.randint(3, wmax)
                e = min(ln - 1, pos + w)
                recs.append((rnd.randrange(6), pos, e, rnd.randint(-5, 30)))
                pos = max(0, e - rnd.randint(0, w)) if rnd.random() < p_overlap else e + 1 + rnd.randint(0, 3)
                k += 1
            off.append(len(recs))
    return recs, off


@pytest.mark.parametrize("p_overlap", [0.05, 0.45, 0.9])
def test_every_rank_assembles_its_own_chunk_range(p_overlap):
    """sd_range_assemble_* (csrc/sd_seam.hpp): the seam merge of a read whose chunks are spread over several ranks, made
    by every rank on its own records from the exchanged edges -- the texts of the ranks in rank order must be the bytes
    sd_assemble_tsv makes of all records.  Shares inside one read, shares that end / begin inside a read, reads that lie
    completely inside a share; records that overlap seldom (the real case: the scans join at once), often, and nearly
    always (the assumed scan and the real one stay apart for long: the piece is formatted again)."""
    import random
    import numpy as np
    from stringdecomposer_amd import shard
    rnd = random.Random(int(p_overlap * 100) + 5)
    part, ov = 100, 30
    rec_dt = lib._rec_dtype()
    mono = ["m0", "m1", "m2"]
    again = 0
    for case in range(12):
        read_lens = [rnd.choice([40, 900, 2500, 30_000, 120_000]) for _ in range(rnd.randint(1, 6))]
        if case % 3 == 0:
            read_lens = [rnd.randint(150_000, 400_000)]          # one chromosome: every share lies inside it
        names = ["r%d" % i for i in range(len(read_lens))]
        recs, off = _seam_records(rnd, read_lens, part, ov, p_overlap)
        arr = np.array(recs, dtype=rec_dt)
        off = np.array(off, dtype=np.int64)
        n_chunks = len(off) - 1
        want = lib.assemble_tsv(names, read_lens, mono, arr, off, part_size=part, overlap=ov, threads=2)
        for world in (1, 2, 3, 5, 8):
            asm = []
            for rank in range(world):
                lo, hi = shard.block_range(n_chunks, rank, world)
                asm.append(lib.RangeAssembler.from_lists(names, read_lens, mono, lo, hi, arr[off[lo]:off[hi]], off[lo:hi + 1] - off[lo],
                                                         part_size=part, overlap=ov, threads=1 + rank % 3))
            edges = [a.edge for a in asm]
            shareable = all(lib.SeamEdge.from_buffer_copy(e).ok for e in edges)
            if not shareable:
                with pytest.raises(lib.SdError) as ei:
                    asm[0].text(edges, 0)
                assert ei.value.code == lib.SD_ERR_UNSUPPORTED
                # refused only when a share is empty or a crossing piece has fewer than 32 records
                cs = np.cumsum([0] + [len(lib.chunk_plan(rl, part, ov)) for rl in read_lens])
                short = False
                for rank in range(world):
                    lo, hi = shard.block_range(n_chunks, rank, world)
                    if hi == lo:
                        short = True
                        continue
                    ra = int(np.searchsorted(cs, lo, side="right")) - 1
                    rb = int(np.searchsorted(cs, hi - 1, side="right")) - 1
                    of, ob = lo > cs[ra], hi < cs[rb + 1]
                    if of and ob and ra == rb:
                        short |= off[hi] - off[lo] < 32
                    else:
                        short |= of and off[min(hi, cs[ra + 1])] - off[lo] < 32
                        short |= ob and off[hi] - off[max(lo, cs[rb])] < 32
                assert short, (read_lens, world)
                continue
            got = b""
            for rank, a in enumerate(asm):
                n = a.text(edges, rank)
                t = a.bytes()
                assert len(t) == n
                got += t
                again += a.stats()["formatted_again"]
            assert got == want, (case, world, read_lens)
            for a in asm:
                a.close()
    if p_overlap <= 0.05:
        assert again == 0


def test_rank_local_assembly_when_the_real_scan_never_meets_the_assumed_one():
    """Records built so that every position of the merge scan jumps three ahead (each record covers the next): scans
    that enter at positions of different residues never share a position, the text a rank made ahead from the assumed
    entry is wrong and sd_range_assemble_text formats the piece again -- same bytes as the serial merge."""
    import numpy as np
    from stringdecomposer_amd import shard
    part, ov = 100, 30
    rec_dt = lib._rec_dtype()
    mono = ["m0", "m1", "m2"]
    read_lens = [3000]
    names = ["chr"]
    again = 0
    for per_chunk in (299, 300, 301, 302):
        recs, off = [], [0]
        for k, (_o, ln) in enumerate(lib.chunk_plan(read_lens[0], part, ov)):
            recs.extend([(k % 6, 5, min(95, ln - 1), 7)] * (per_chunk + k % 2))
            off.append(len(recs))
        arr = np.array(recs, dtype=rec_dt)
        off = np.array(off, dtype=np.int64)
        n_chunks = len(off) - 1
        want = lib.assemble_tsv(names, read_lens, mono, arr, off, part_size=part, overlap=ov, threads=2)
        for world in (2, 3, 4, 7):
            asm = []
            for rank in range(world):
                lo, hi = shard.block_range(n_chunks, rank, world)
                asm.append(lib.RangeAssembler.from_lists(names, read_lens, mono, lo, hi, arr[off[lo]:off[hi]], off[lo:hi + 1] - off[lo],
                                                         part_size=part, overlap=ov, threads=2))
            edges = [a.edge for a in asm]
            got = b""
            for rank, a in enumerate(asm):
                a.text(edges, rank)
                got += a.bytes()
                again += a.stats()["formatted_again"]
            assert got == want, (per_chunk, world)
    assert again > 0
    # misuse is refused, not answered with wrong text: the text of a handle is made once, and edges[rank] must be its own
    lo, hi = shard.block_range(n_chunks, 1, 3)
    a = lib.RangeAssembler.from_lists(names, read_lens, mono, lo, hi, arr[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], part_size=part, overlap=ov)
    others = [lib.RangeAssembler.from_lists(names, read_lens, mono, *shard.block_range(n_chunks, g, 3),
                                            arr[off[shard.block_range(n_chunks, g, 3)[0]]:off[shard.block_range(n_chunks, g, 3)[1]]],
                                            off[shard.block_range(n_chunks, g, 3)[0]:shard.block_range(n_chunks, g, 3)[1] + 1] - off[shard.block_range(n_chunks, g, 3)[0]],
                                            part_size=part, overlap=ov).edge for g in (0, 2)]
    edges = [others[0], a.edge, others[1]]
    with pytest.raises(lib.SdError) as ei:
        a.text(edges, 0)            # rank 0's edge has no front piece: not this handle's
    assert ei.value.code == lib.SD_ERR_PARAM
    a.text(edges, 1)
    with pytest.raises(lib.SdError) as ei:
        a.text(edges, 1)
    assert ei.value.code == lib.SD_ERR_PARAM


def test_pipeline_cache_key_and_batch_planner_contracts():
    """VERDICT r05 (weak 10): the invariants of the pipeline cache key and of the batch planner were only testable through
    whole jobs on a GPU.  sd_pipeline_logic_selftest (csrc/sd_engine.hip) checks them on the host: the key covers every
    engine-shaping parameter and the monomer set (boundaries, order) but not the host-thread count; batches are consecutive,
    complete, within the budget, at least min_batches where the chunks allow, never a small remainder."""
    import ctypes as C
    L = lib.load()
    err = C.create_string_buffer(1024)
    rc = L.sd_pipeline_logic_selftest(err, C.c_size_t(1024))
    assert rc == lib.SD_OK, err.value.decode()


def test_fast_exit_of_the_launcher_loses_no_output(tmp_path):
    """VERDICT r05 (weak 12): bin/stringdecomposer leaves without the runtime's tear-down.  What it leaves through
    (stringdecomposer_amd.leave_without_teardown) must not depend on every output having been closed: exit handlers run, open
    Python files (text and binary, never closed here) and the C stdio buffers are flushed, the exit code is the caller's."""
    import subprocess
    import sys
    out = tmp_path / "o"
    code = (
        "import sys, atexit, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "import stringdecomposer_amd as s\n"
        "f = open(sys.argv[1] + '.txt', 'w'); f.write('x' * 1000)\n"
        "g = open(sys.argv[1] + '.bin', 'wb'); g.write(b'y' * 10)\n"
        "atexit.register(lambda: open(sys.argv[1] + '.atexit', 'w').write('ran'))\n"
        "libc = ctypes.CDLL(None); libc.fopen.restype = ctypes.c_void_p\n"
        "h = libc.fopen((sys.argv[1] + '.c').encode(), b'w'); libc.fputs(b'stdio', ctypes.c_void_p(h))\n"
        "print('tail of stdout', end='')\n"
        "s.leave_without_teardown(7)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code, str(out)], capture_output=True, timeout=120)
    assert r.returncode == 7, r.stderr.decode()[-500:]
    assert r.stdout == b"tail of stdout"
    assert open(str(out) + ".txt").read() == "x" * 1000
    assert open(str(out) + ".bin", "rb").read() == b"y" * 10
    assert open(str(out) + ".atexit").read() == "ran"
    assert open(str(out) + ".c").read() == "stdio"
