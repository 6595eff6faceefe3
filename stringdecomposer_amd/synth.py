"""Deterministic synthetic monomer / read generator.

Version-stable: every random number comes from a counter-based splitmix64 implemented here with
numpy uint64 arithmetic, so the same (seed, stream) gives the same bytes on any numpy version.

  ancestor        171 random ACGT
  12-monomer set  ancestor mutated (sub 20 %, ins 1 %, del 1 %)
  64-monomer set  5 families (ancestor mutated 25/1/1 %) x per-monomer 10/0.5/0.5 %
  reads           cyclic concatenation of the monomers from a random phase, every copy mutated
                  (sub 4 %, ins 3 %, del 3 %), truncated to exactly read_len, 50 % reverse-complemented
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def _mix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream_key(seed, stream):
    with np.errstate(over="ignore"):
        s = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        k = _mix(np.asarray([s + _GOLD], dtype=np.uint64))[0]
        return _mix(np.asarray([k ^ (np.uint64(stream & 0xFFFFFFFFFFFFFFFF) * _M2 + _GOLD)],
                               dtype=np.uint64))[0]


class Stream:
    """Counter-based splitmix64 stream: u64(n) returns the next n values."""

    def __init__(self, seed, stream):
        self.key = _stream_key(seed, stream)
        self.ctr = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            self.ctr += n
            return _mix(self.key + idx * _GOLD)

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def below(self, n, k):
        return ((self.u64(n) >> np.uint64(33)) % np.uint64(k)).astype(np.int64)


def mutate(seq, st, psub, pins, pdel):
    """seq: uint8 codes 0..3.  Per source base: delete (pdel) / substitute (psub) / keep, and
    independently insert one random base after it (pins)."""
    n = len(seq)
    r = st.uniform(n)
    ri = st.uniform(n)
    alt = st.below(n, 3)
    insb = st.below(n, 4)
    keep = r >= pdel
    sub = keep & (r < pdel + psub)
    base = np.where(sub, (seq + 1 + alt) % 4, seq)
    ins = ri < pins
    counts = keep.astype(np.int64) + ins.astype(np.int64)
    pos = np.cumsum(counts) - counts
    out = np.empty(int(counts.sum()), dtype=np.int64)
    out[pos[keep]] = base[keep]
    out[(pos + keep)[ins]] = insb[ins]
    return out


def _to_ascii(codes):
    return _ACGT[codes].tobytes()


def revcomp_bytes(b):
    return _COMP[np.frombuffer(b, dtype=np.uint8)][::-1].tobytes()


def make_monomers(n_monomers=12, seed=1, length=171):
    """Returns (names, seqs) with seqs as ASCII bytes."""
    anc = Stream(seed, 1).below(length, 4)
    seqs = []
    if n_monomers <= 16:
        for j in range(n_monomers):
            seqs.append(_to_ascii(mutate(anc, Stream(seed, 100 + j), 0.20, 0.01, 0.01)))
    else:
        fams = [mutate(anc, Stream(seed, 50 + f), 0.25, 0.01, 0.01) for f in range(5)]
        for j in range(n_monomers):
            seqs.append(_to_ascii(mutate(fams[j % 5], Stream(seed, 100 + j), 0.10, 0.005, 0.005)))
    names = ["M%d" % j for j in range(n_monomers)]
    return names, seqs


def make_reads(monomer_seqs, n_reads, read_len=50000, seed=1, first_index=0,
               psub=0.04, pins=0.03, pdel=0.03):
    """Returns (names, seqs).  Read i depends only on (seed, first_index + i), so any shard of the
    read set can be generated independently (used for multi-GPU sharding)."""
    code = np.zeros(256, dtype=np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    mons = [code[np.frombuffer(m, dtype=np.uint8)] for m in monomer_seqs]
    M = len(mons)
    mean_len = sum(len(m) for m in mons) / M
    names, seqs = [], []
    for r in range(first_index, first_index + n_reads):
        st = Stream(seed, 1000000 + r)
        hdr = st.u64(3)
        m0 = int(hdr[0] % np.uint64(M))
        phase = int(hdr[1] % np.uint64(len(mons[m0])))
        rc = bool(hdr[2] & np.uint64(1))
        pieces, total, j = [], 0, m0
        need = read_len * 1.08 + 2 * mean_len
        first = True
        while total < need:
            m = mons[j % M]
            if first:
                m = m[phase:]
                first = False
            pieces.append(m)
            total += len(m)
            j += 1
        src = np.concatenate(pieces)
        out = mutate(src, st, psub, pins, pdel)
        while len(out) < read_len:  # practically never; keep deterministic anyway
            out = np.concatenate([out, mutate(src, st, psub, pins, pdel)])
        b = _to_ascii(out[:read_len])
        if rc:
            b = revcomp_bytes(b)
        names.append("r%d" % r)
        seqs.append(b)
    return names, seqs


def write_fasta(path, names, seqs, width=0):
    with open(path, "wb") as f:
        for n, s in zip(names, seqs):
            f.write(b">" + (n.encode() if isinstance(n, str) else n) + b"\n")
            if width and width > 0:
                for i in range(0, len(s), width):
                    f.write(s[i:i + width] + b"\n")
            else:
                f.write(s + b"\n")
