"""Run the per-lane device primitives (csrc/grm_device_fns.h) on the CPU and compare them
with the oracle.  No GPU needed: tests/host/host_emul.cpp includes the very header the
gfx950 kernels include; block-level cooperation is replaced by its sequential meaning."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_ctypes as orc
from tests import cases

HERE = os.path.dirname(os.path.abspath(__file__))
TILE = 16384


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "host", "host_emul.cpp")
    hdr = os.path.join(HERE, "..", "genomic-resistance-mapping-grm-_amd", "csrc", "grm_device_fns.h")
    # tests/test_sanitizers.py runs this module once more in a child process with AddressSanitizer + UBSan preloaded: the per-lane device
    # code is the one place where an out-of-range shift or index of the kernels can be caught without a GPU
    sanitized = os.environ.get("GRM_HOST_EMUL_SANITIZED") == "1"
    so = os.path.join(HERE, "host", "libhost_emul_asan.so" if sanitized else "libhost_emul.so")
    flags = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if sanitized else []
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared"] + flags + ["-o", so, src])
    L = C.CDLL(so)
    L.emul_parse.restype = C.c_uint64
    L.emul_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.emul_kmers.restype = C.c_uint64
    L.emul_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    L.emul_parse2.restype = C.c_uint64
    L.emul_parse2.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.emul_parse3.restype = C.c_uint64
    L.emul_parse3.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.emul_parse_fastq.restype = C.c_uint64
    L.emul_parse_fastq.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.emul_kmers_wide.restype = C.c_uint64
    L.emul_kmers_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    L.emul_kmers16.restype = C.c_uint64
    L.emul_kmers16.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    L.emul_kmers32.restype = C.c_uint64
    L.emul_kmers32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    L.emul_summarize.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64] + [C.POINTER(C.c_uint32)] * 3
    L.emul_valid_starts.restype = C.c_uint64
    L.emul_valid_starts.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    L.emul_revcomp.restype = C.c_uint64
    L.emul_revcomp.argtypes = [C.c_uint64, C.c_int]
    L.emul_mix64.restype = C.c_uint64
    L.emul_mix64.argtypes = [C.c_uint64]
    L.emul_runs_wide.restype = C.c_uint64
    L.emul_runs_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.c_void_p, C.c_uint64, C.c_void_p]
    L.emul_runs.restype = C.c_uint64
    L.emul_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.emul_minimizer_bucket_of_kmer.restype = C.c_uint32
    L.emul_minimizer_bucket_of_kmer.argtypes = [C.c_uint64, C.c_int, C.c_int]
    return L


def layout(files):
    """the raw image grm_batch_upload builds: '>\\n' + bytes + '\\n', '\\n'-padded to a tile"""
    out = bytearray()
    for f in files:
        img = b">\n" + f + b"\n"
        out += img
        out += b"\n" * ((-len(out)) % TILE)
    if not out:
        out += b"\n" * TILE
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def extract(emul, files, k):
    raw = layout(files)
    ng = len(raw) // 64 + 8
    sym2 = np.zeros(2 * ng, dtype=np.uint64)
    inv = np.zeros(ng, dtype=np.uint64)
    nsym = emul.emul_parse(raw.ctypes.data, len(raw), sym2.ctypes.data, inv.ctypes.data, ng)
    # restructured parse path (associative tile scan + bit-string insertion): identical stream
    for tile_bytes in (TILE, 256):
        s2 = np.zeros(2 * ng, dtype=np.uint64)
        i2 = np.zeros(ng, dtype=np.uint64)
        n2 = emul.emul_parse2(raw.ctypes.data, len(raw), tile_bytes, s2.ctypes.data, i2.ctypes.data, ng)
        assert n2 == nsym and (s2 == sym2).all() and (i2 == inv).all()
        # and with the chunks of letters and newlines only taking the kernels' short route
        nc = C.c_uint64(0)
        n3 = emul.emul_parse3(raw.ctypes.data, len(raw), tile_bytes, s2.ctypes.data, i2.ctypes.data, ng, C.byref(nc))
        assert n3 == nsym and (s2 == sym2).all() and (i2 == inv).all()
    cap = max(1, int(nsym))
    out = np.zeros(cap, dtype=np.uint64)
    n = emul.emul_kmers(sym2.ctypes.data, inv.ctypes.data, nsym, k, out.ctypes.data, cap)
    assert n <= cap
    out32 = np.zeros(cap, dtype=np.uint64)
    n32 = emul.emul_kmers32(sym2.ctypes.data, inv.ctypes.data, nsym, k, out32.ctypes.data, cap)
    assert n32 == n and (out32[:n] == out[:n]).all()      # 32-position iterator == 64-position iterator
    n16 = emul.emul_kmers16(sym2.ctypes.data, inv.ctypes.data, nsym, k, out32.ctypes.data, cap)
    assert n16 == n and (out32[:n] == out[:n]).all()      # 16-position iterator too
    return raw, int(nsym), out[:n]


@pytest.mark.parametrize("name,k,genomes", [c for c in cases.micro_cases() if c[1] <= 32], ids=lambda x: x if isinstance(x, str) else None)
def test_extraction_matches_oracle(emul, name, k, genomes):
    for texts in genomes:
        files = [t.encode() for t in texts]
        _, _, got = extract(emul, files, k)
        km, ct, nocc = orc.count_genome(files, k)
        assert len(got) == nocc
        vals, counts = np.unique(got, return_counts=True)
        assert (vals == km[:, 0]).all()
        assert (counts == ct).all()


def test_long_lines_and_tile_straddling(emul):
    rng = np.random.RandomState(7)
    # single-line 40 kb sequence (no newline inside several tiles), a header that straddles a
    # tile boundary, and a very long header
    seq = cases.rand_seq(rng, 40000)
    f1 = (">one\n" + seq + "\n").encode()
    pad = TILE - 2 - 5          # puts the next '>' header start near the end of tile 0
    f2 = (">a\n" + cases.rand_seq(rng, pad - 3 - 1) + "\n>" + "h" * 40000 + "\n" + cases.rand_seq(rng, 500) + "\n").encode()
    f3 = cases.fasta([("x", cases.rand_seq(rng, 70000))], width=61).encode()
    for files in ([f1], [f2], [f3], [f1, f2, f3]):
        for k in (31, 8):
            raw, nsym, got = extract(emul, files, k)
            km, ct, nocc = orc.count_genome(files, k)
            assert len(got) == nocc
            vals, counts = np.unique(got, return_counts=True)
            assert (vals == km[:, 0]).all() and (counts == ct).all()
            # tile-summary algebra (parse_summarize + parse_scan): resolving the incoming
            # line type tile by tile must reproduce the symbol count
            state, total = 0, 0
            for t0 in range(0, len(raw), TILE):
                kn, un, le = C.c_uint32(), C.c_uint32(), C.c_uint32()
                emul.emul_summarize(raw.ctypes.data, t0, t0 + TILE, C.byref(kn), C.byref(un), C.byref(le))
                total += kn.value + (un.value if state != 2 else 0)
                if le.value:
                    state = le.value
            assert total == nsym


def test_clean_chunks_take_the_short_route_to_the_same_stream(emul):
    """letters-and-newlines chunks (what nearly every wave of a FASTA holds) through clean_elem / clean_chunk_insert: blank lines, lines
    shorter than a chunk, a newline as the chunk's first or last byte, N runs, lower case, high bytes -- same stream as the masks give;
    chunks with '>' or CR or a digit keep to the general route"""
    rng = np.random.RandomState(77)
    files = []
    for width in (1, 2, 7, 15, 16, 17, 31, 60, 80, 333):
        seq = "".join(rng.choice(list("ACGTNacgtnRYKM"), p=[.2, .2, .2, .2, .03, .03, .03, .03, .03, .01, .01, .01, .01, .01]) for _ in range(3000))
        body = "\n".join(seq[i:i + width] for i in range(0, len(seq), width))
        files.append((">c1 x\n" + body + "\n\n\n>c2\n" + body[:500] + "\n").encode())
    files.append(b">h\nACGT\r\nAC9GT\nAC>GT\n" + bytes([65, 200, 67, 0xff, 71, 10]) * 40)
    files.append(b"ACGT" * 5000)                                    # no header of its own, no newline for 20 000 bytes
    for f in files:
        raw = layout([f])
        ng = len(raw) // 64 + 8
        sym2 = np.zeros(2 * ng, dtype=np.uint64)
        inv = np.zeros(ng, dtype=np.uint64)
        nsym = emul.emul_parse(raw.ctypes.data, len(raw), sym2.ctypes.data, inv.ctypes.data, ng)
        for tile_bytes in (TILE, 64):
            s2 = np.zeros(2 * ng, dtype=np.uint64)
            i2 = np.zeros(ng, dtype=np.uint64)
            nc = C.c_uint64(0)
            n3 = emul.emul_parse3(raw.ctypes.data, len(raw), tile_bytes, s2.ctypes.data, i2.ctypes.data, ng, C.byref(nc))
            assert n3 == nsym and (s2 == sym2).all() and (i2 == inv).all()
            assert nc.value > 0


def test_valid_starts_bruteforce(emul):
    rng = np.random.RandomState(11)
    for _ in range(300):
        i0 = int(rng.randint(0, 2**31)) << 33 | int(rng.randint(0, 2**31)) if rng.rand() < 0.5 else (1 << int(rng.randint(0, 64)))
        i1 = (1 << int(rng.randint(0, 64))) if rng.rand() < 0.7 else 0
        k = int(rng.randint(1, 65))
        got = emul.emul_valid_starts(i0 & (2**64 - 1), i1, k)
        x = (i1 << 64) | (i0 & (2**64 - 1))
        want = 0
        for p in range(64):
            if (x >> p) & ((1 << k) - 1) == 0:
                want |= 1 << p
        assert got == want, (hex(i0), hex(i1), k)


def test_revcomp_and_mix_bijective(emul):
    rng = np.random.RandomState(12)
    for m in (1, 2, 15, 30, 31, 32):
        for _ in range(50):
            s = cases.rand_seq(rng, m)
            v = 0
            for ch in s:
                v = (v << 2) | ((ord(ch) >> 1) & 3)
            r = cases.revcomp(s)
            w = 0
            for ch in r:
                w = (w << 2) | ((ord(ch) >> 1) & 3)
            assert emul.emul_revcomp(v, m) == w


def _hashes(emul, keys):
    return np.array([emul.emul_mix64(int(x)) for x in keys], dtype=np.uint64)


def test_hash_balance(emul):
    """bucket (top 13 bits) and LDS slot (low 12 bits) spread for random keys AND for the
    canonical k-mers of a genome-like sequence (consecutive, overlapping windows)"""
    rng = np.random.RandomState(12)
    rand = rng.randint(0, 2**62, size=60000, dtype=np.int64).astype(np.uint64)
    seq = cases.rand_seq(rng, 60000)
    _, _, walk = extract(emul, [(">g\n" + seq + "\n").encode()], 31)
    lowc = ("ACGT" * 20 + "A" * 40 + cases.rand_seq(rng, 200)) * 150          # repeats / low complexity
    _, _, rep = extract(emul, [(">g\n" + lowc + "\n").encode()], 31)
    for name, keys in (("random", rand), ("genome walk", np.unique(walk)), ("repeats", np.unique(rep))):
        hs = _hashes(emul, keys)
        nb = 256 if len(keys) > 20000 else 16
        b = (hs >> np.uint64(64 - int(np.log2(nb)))).astype(np.int64)
        cnt = np.bincount(b, minlength=nb)
        tol = 5 * np.sqrt(cnt.mean())                     # Poisson spread of an ideal hash
        assert cnt.max() < cnt.mean() + tol and cnt.min() > cnt.mean() - tol, (name, cnt.min(), cnt.mean(), cnt.max())
        # inside one bucket the slot bits must not cluster: expected distinct slots ~ m(1-exp(-n/m))
        top = (hs >> np.uint64(64 - 4)).astype(np.int64)
        sel = hs[top == 3]
        slots = (sel & np.uint64(4095)).astype(np.int64)
        n, m = len(sel), 4096
        expect = m * (1 - np.exp(-n / m))
        assert len(np.unique(slots)) > 0.9 * expect, (name, len(np.unique(slots)), expect)


def test_fastq_primitives_match_oracle(emul):
    """FASTQ classification (phase masks, associative elements) on the CPU vs the oracle"""
    for name, k, genomes in cases.fastq_cases():
        for texts in genomes:
            for t in texts:
                f = t.encode()
                img = f + b"\n"
                img += b"\n" * ((-len(img)) % TILE)
                raw = np.frombuffer(img, dtype=np.uint8).copy()
                for tile_bytes in (TILE, 64):
                    ng = len(raw) // 64 + 8
                    sym2 = np.zeros(2 * ng, dtype=np.uint64)
                    inv = np.zeros(ng, dtype=np.uint64)
                    nsym = emul.emul_parse_fastq(raw.ctypes.data, len(raw), tile_bytes, sym2.ctypes.data, inv.ctypes.data, ng)
                    out = np.zeros(max(1, int(nsym)), dtype=np.uint64)
                    n = emul.emul_kmers(sym2.ctypes.data, inv.ctypes.data, nsym, k, out.ctypes.data, len(out))
                    km, ct, nocc = orc.count_genome([f], k)
                    assert n == nocc
                    vals, counts = np.unique(out[:n], return_counts=True)
                    assert (vals == km[:, 0]).all() and (counts == ct).all()


@pytest.mark.parametrize("k", [33, 47, 63, 64])
def test_wide_kmers_match_oracle(emul, k):
    rng = np.random.RandomState(k)
    a = cases.rand_seq(rng, 700)
    texts = [cases.fasta([("a", a), ("b", cases.revcomp(a)[:300]), ("c", a[:90] + "N" + a[91:200]), ("short", a[:k - 1])], width=70)]
    files = [t.encode() for t in texts]
    raw = layout(files)
    ng = len(raw) // 64 + 8
    sym2 = np.zeros(2 * ng, dtype=np.uint64)
    inv = np.zeros(ng, dtype=np.uint64)
    nsym = emul.emul_parse(raw.ctypes.data, len(raw), sym2.ctypes.data, inv.ctypes.data, ng)
    out = np.zeros(2 * max(1, int(nsym)), dtype=np.uint64)
    n = emul.emul_kmers_wide(sym2.ctypes.data, inv.ctypes.data, nsym, k, out.ctypes.data, len(out) // 2)
    km, ct, nocc = orc.count_genome(files, k)
    assert n == nocc
    got = out[:2 * n].reshape(n, 2)
    order = np.lexsort((got[:, 1], got[:, 0]))
    got = got[order]
    uniq, counts = np.unique(got, axis=0, return_counts=True)
    assert uniq.shape == km.shape and (uniq == km).all() and (counts == ct).all()


def _parse(emul, files):
    raw = layout(files)
    ng = len(raw) // 64 + 8
    sym2 = np.zeros(2 * ng, dtype=np.uint64)
    inv = np.zeros(ng, dtype=np.uint64)
    nsym = int(emul.emul_parse(raw.ctypes.data, len(raw), sym2.ctypes.data, inv.ctypes.data, ng))
    return sym2, inv, nsym


def _runs(emul, sym2, inv, nsym, k, lo=0, hi=None, coarse_bits=9):
    """-> (keys, buckets, records) of the genome [lo, hi) of the stream; records: (n, 6) uint32 {len, flipped, x lo, x hi, y lo, y hi}"""
    hi = nsym if hi is None else hi
    cap = max(1, nsym)
    keys = np.zeros(cap, dtype=np.uint64)
    buckets = np.zeros(cap, dtype=np.uint32)
    recs = np.zeros((cap, 6), dtype=np.uint32)
    n_rec = C.c_uint64(0)
    n = emul.emul_runs(sym2.ctypes.data, inv.ctypes.data, nsym, lo, hi, k, coarse_bits, keys.ctypes.data, buckets.ctypes.data, cap,
                       recs.ctypes.data, cap, C.byref(n_rec))
    assert n <= cap, "emul_runs found an inconsistency: code %d" % ((1 << 64) - 1 - n)
    return keys[:n], buckets[:n], recs[:n_rec.value]


def _record_words(recs):
    x = recs[:, 2].astype(np.uint64) | (recs[:, 3].astype(np.uint64) << np.uint64(32))
    y = recs[:, 4].astype(np.uint64) | (recs[:, 5].astype(np.uint64) << np.uint64(32))
    return x, y


@pytest.mark.parametrize("k", [11, 12, 15, 21, 27, 31, 32])
def test_run_records_decode_to_the_oracles_kmers(emul, k):
    """record form of the partition, on the CPU with the kernels' own per-lane functions: the runs of one minimizer occurrence,
    as 16-byte records, decode to exactly the canonical k-mers the oracle counts -- record after record the k-mers of the stream
    in order, or in reverse order where the record is stored on the other strand; every k-mer of a record has the record's
    bucket when that is re-derived from the k-mer alone (what the probing fill and the ranks rely on); a record holds
    1 .. k - 10 k-mers, and a random text makes runs of every length"""
    rng = np.random.RandomState(100 + k)
    files = [cases.fasta([("a", cases.rand_seq(rng, 20_000)), ("b", "ACGT" * 300 + "A" * 200 + cases.rand_seq(rng, 3000) + "N" * 7 + "GATTACA" * 90)], width=70).encode(),
             (">c\n" + cases.rand_seq(rng, 5000).lower() + "\n").encode()]
    raw, nsym, direct = extract(emul, files, k)
    sym2, inv, nsym2 = _parse(emul, files)
    assert nsym2 == nsym
    keys, buckets, recs = _runs(emul, sym2, inv, nsym, k)
    n = len(keys)
    assert n == len(direct)
    lens, flipped = recs[:, 0].astype(np.int64), recs[:, 1]
    assert lens.sum() == n and lens.min() >= 1 and lens.max() == k - 10
    at = 0
    for ln, fl in zip(lens, flipped):
        want = direct[at:at + ln]
        assert (keys[at:at + ln] == (want[::-1] if fl else want)).all()
        at += ln
    assert 0.25 < flipped.mean() < 0.75                          # both orientations occur
    km, ct, nocc = orc.count_genome(files, k)
    vals, counts = np.unique(keys, return_counts=True)
    assert nocc == n and (vals == km[:, 0]).all() and (counts == ct).all()
    nbits = 9 + 7
    for i in rng.randint(0, n, size=400):
        assert emul.emul_minimizer_bucket_of_kmer(int(keys[i]), k, nbits) == int(buckets[i])
        # fewer buckets = the top bits (what level 2 and the union of ranks with different bucket counts rely on)
        assert emul.emul_minimizer_bucket_of_kmer(int(keys[i]), k, nbits - 5) == int(buckets[i]) >> 5
    if k >= 19:
        assert 0.8 * 2 / (k - 9) < len(recs) / n < 1.3 * 2 / (k - 9)      # ~2 / (W + 1) runs per k-mer (low-complexity stretches: more)


@pytest.mark.parametrize("k", [33, 34, 47, 63, 64])
def test_wide_run_records_decode_to_the_oracles_kmers(emul, k):
    """two-word k-mers through the record form (24-byte records; the minimizer among the 21 / 22 m-mers in the middle of the k-mer):
    record after record the k-mers of the stream in order, or in reverse where the record is stored on the other strand; together
    exactly the oracle's counted k-mers; a k-mer and its reverse complement get the same bucket"""
    rng = np.random.RandomState(200 + k)
    a = cases.rand_seq(rng, 6000)
    files = [cases.fasta([("a", a), ("b", "ACGT" * 100 + "A" * 150 + cases.rand_seq(rng, 2000) + "N" * 3 + "GATTACA" * 60), ("short", a[:k - 1])], width=70).encode(),
             (">c\n" + cases.revcomp(a)[100:4000].lower() + "\n").encode()]
    sym2, inv, nsym = _parse(emul, files)
    cap = max(1, nsym)
    direct = np.zeros(2 * cap, dtype=np.uint64)
    nd = emul.emul_kmers_wide(sym2.ctypes.data, inv.ctypes.data, nsym, k, direct.ctypes.data, cap)
    direct = direct[:2 * nd].reshape(nd, 2)
    keys = np.zeros(2 * cap, dtype=np.uint64)
    buckets = np.zeros(cap, dtype=np.uint32)
    recs = np.zeros((cap, 2), dtype=np.uint32)
    n_rec = C.c_uint64(0)
    n = emul.emul_runs_wide(sym2.ctypes.data, inv.ctypes.data, nsym, 0, nsym, k, 9, keys.ctypes.data, buckets.ctypes.data, cap,
                            recs.ctypes.data, cap, C.byref(n_rec))
    assert n <= cap, "emul_runs_wide found an inconsistency: code %d" % ((1 << 64) - 1 - n)
    assert n == nd
    keys = keys[:2 * n].reshape(n, 2)
    recs = recs[:n_rec.value]
    lens, flipped = recs[:, 0].astype(np.int64), recs[:, 1]
    w = 21 if (k - 10) % 2 else 22
    assert lens.sum() == n and lens.min() >= 1 and lens.max() == w
    at = 0
    for ln, fl in zip(lens, flipped):
        want = direct[at:at + ln]
        assert (keys[at:at + ln] == (want[::-1] if fl else want)).all()
        at += ln
    assert 0.25 < flipped.mean() < 0.75
    km, ct, nocc = orc.count_genome(files, k)
    uniq, counts = np.unique(keys, axis=0, return_counts=True)
    assert nocc == n and uniq.shape == km.shape and (uniq == km).all() and (counts == ct).all()
    # one bucket per k-mer, whatever the strand it was read on (the reverse-complemented copy of `a`)
    seen = {}
    for (h, l), b in zip(keys.tolist(), buckets[:n].tolist()):
        assert seen.setdefault((h, l), b) == b


def _record_multiset(recs):
    x, y = _record_words(recs)
    return sorted(zip(x.tolist(), y.tolist()))


@pytest.mark.parametrize("k", [19, 31, 32])
def test_run_records_do_not_depend_on_frame_or_strand(emul, k):
    """what dict_build's record memo lives on: the same sequence gives the same 16-byte records wherever it starts in the packed
    stream (an insertion upstream, another contig order), on whichever strand it is given, and however the genome is cut into
    genomes / parts around it -- apart from the runs that touch the changed place itself"""
    rng = np.random.RandomState(7 + k)
    seq = cases.rand_seq(rng, 6000)

    def records_of(text_files, lo_hi=None):
        sym2, inv, nsym = _parse(emul, text_files)
        if lo_hi is None:
            return _record_multiset(_runs(emul, sym2, inv, nsym, k)[2])
        out = []
        for lo, hi in lo_hi(nsym):
            out += _record_multiset(_runs(emul, sym2, inv, nsym, k, lo, hi)[2])
        return sorted(out)

    base = records_of([(">s\n" + seq + "\n").encode()])
    assert len(base) > 300
    # every phase of the 32-position windows: 1 .. 40 bases of another contig in front
    for shift in (1, 2, 5, 13, 31, 32, 33, 40):
        pre = cases.rand_seq(rng, shift)
        got = records_of([(">p\n" + pre + "\n>s\n" + seq + "\n").encode()])
        assert got == sorted(base + records_of([(">p\n" + pre + "\n").encode()]))
    # the other strand, alone and behind another contig.  (Two equal order values inside one k-mer -- the same canonical m-mer
    # twice, or a 24-bit collision -- resolve to the LEFTMOST occurrence, which is the other one on the other strand: the two
    # runs around such a tie are cut one position apart.  Rare, and only a missed memo hit.)
    def differing(a, b):
        from collections import Counter
        ca, cb = Counter(a), Counter(b)
        return sum(((ca - cb) + (cb - ca)).values())
    rc = cases.revcomp(seq)
    assert differing(records_of([(">r\n" + rc + "\n").encode()]), base) <= len(base) // 100
    other = cases.rand_seq(rng, 777)
    both = records_of([(">o\n" + other + "\n>r\n" + rc + "\n").encode()])
    alone = records_of([(">o\n" + other + "\n").encode()])
    assert differing(both, base + alone) <= len(base) // 100
    # an insertion of 3 bases in the middle: only the runs that contain the junction change
    ins = seq[:3000] + "GAT" + seq[3000:]
    got = records_of([(">i\n" + ins + "\n").encode()])
    common = set(got) & set(base)
    assert len(base) - len(common) <= 2 * (k - 10) and len(got) - len(common) <= 2 * (k - 10)
    # the stream cut into two "genomes" at any symbol: the union of their records is the whole (a run never spans the cut only
    # when the cut is a separator; inside a contig both sides keep their own k-mers and the k-mers across the cut belong to neither)
    text = [(">a\n" + seq[:2500] + "\n>b\n" + seq[2500:] + "\n").encode()]
    sym2, inv, nsym = _parse(emul, text)
    whole = _record_multiset(_runs(emul, sym2, inv, nsym, k)[2])
    cut = 1 + 2500 + 1                                          # the separator of contig b: symbols = sep a, 2500 bases, sep b, ...
    parts = _record_multiset(_runs(emul, sym2, inv, nsym, k, 0, cut)[2]) + _record_multiset(_runs(emul, sym2, inv, nsym, k, cut, nsym)[2])
    assert sorted(parts) == whole
