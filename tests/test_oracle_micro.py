"""C oracle vs the independent pure-Python mirror on hand-built micro genomes
(SURVEY 8(c) golden-vector item (iv)): N / lowercase / IUPAC / CRLF / multi-contig /
contig shorter than k / reverse-complement duplicates / multi-file genomes / FASTQ."""
import numpy as np
import pytest

from oracle import oracle_ctypes as orc
from oracle import pyoracle
from tests import cases


def _check_case(k, genomes, abundance_min, filter_singleton):
    bufs = [[t.encode() for t in g] for g in genomes]
    m = orc.build_matrix(bufs, k, abundance_min, filter_singleton)
    py_sets = []
    for g, texts in enumerate(genomes):
        want, n_occ = pyoracle.count_genome(texts, k, abundance_min)
        km, ct = m["per_genome"][g]
        assert orc.decode_kmers(km, k) == [w[0] for w in want]
        assert ct.tolist() == [w[1] for w in want]
        py_sets.append(want)
    kmers, rows = pyoracle.build_matrix(py_sets, filter_singleton)
    assert orc.decode_kmers(m["kmers"], k) == kmers
    got = m["matrix"]
    assert got.shape == (len(rows), len(kmers))
    for r in range(len(rows)):
        assert [int(x) for x in got[r]] == rows[r]
    # dictionary strictly ascending as integers
    if m["words"] == 1 and len(kmers) > 1:
        assert (np.diff(m["kmers"][:, 0].astype(object)) > 0).all()


@pytest.mark.parametrize("name,k,genomes", cases.micro_cases(), ids=lambda x: x if isinstance(x, str) else None)
@pytest.mark.parametrize("amin,filt", [(1, False), (1, True), (2, False)])
def test_micro(name, k, genomes, amin, filt):
    _check_case(k, genomes, amin, filt)


@pytest.mark.parametrize("name,k,genomes", cases.fastq_cases(), ids=lambda x: x if isinstance(x, str) else None)
@pytest.mark.parametrize("amin,filt", [(1, False), (2, True)])
def test_fastq(name, k, genomes, amin, filt):
    _check_case(k, genomes, amin, filt)


def test_two_word_k63():
    rng = np.random.RandomState(3)
    a = cases.rand_seq(rng, 300)
    genomes = [[cases.fasta([("a", a)])], [cases.fasta([("b", cases.revcomp(a)[:200])])], [cases.fasta([("c", a[:100] + "N" + a[101:])])]]
    for k in (33, 63, 64):
        _check_case(k, genomes, 1, False)
        _check_case(k, genomes, 1, True)


def test_three_and_four_word_kmers():
    """k = 65 .. 128 (the reference's --kmer-size range ends at 128, bin/kover/kover:114): 256-bit keys in the C oracle vs
    the string mirror, incl. N, lowercase, the reverse strand, a contig shorter than k and duplicates (abundance-min 2)"""
    rng = np.random.RandomState(8)
    a = cases.rand_seq(rng, 500)
    b = cases.rand_seq(rng, 140)
    genomes = [[cases.fasta([("a", a), ("short", b[:90])])],
               [cases.fasta([("rc", cases.revcomp(a)[:400]), ("again", a[50:300])])],
               [cases.fasta([("n", a[:200] + "N" + a[201:]), ("low", b.lower())], width=61)]]
    for k in (65, 95, 96, 97, 127, 128):
        _check_case(k, genomes, 1, False)
        _check_case(k, genomes, 1, True)
        _check_case(k, genomes, 2, False)
    # words: most significant first, ceil(k / 32) of them
    km = "ACGT" * 32
    can, words = orc.canonical_ascii(km)
    assert len(words) == 4 and can == pyoracle.canonical(km)
    v = pyoracle.kmer_value(can)
    assert [int(w) for w in words] == [(v >> (64 * (3 - j))) & (2**64 - 1) for j in range(4)]


def test_hand_computed():
    # k=3 over ACGTT: windows ACG,CGT,GTT ; rc: CGT,ACG,AAC -> canon ACG,ACG,AAC
    km, ct, nocc = orc.count_genome([b">s\nACGTT\n"], 3)
    assert nocc == 3
    assert orc.decode_kmers(km, 3) == ["AAC", "ACG"]
    assert ct.tolist() == [1, 2]
    # N splits: ACNGT with k=2 -> AC, GT ; canon(AC)=AC (rc GT: G>A) ; canon(GT)=AC
    km, ct, nocc = orc.count_genome([b">s\nACNGT\n"], 2)
    assert nocc == 2 and orc.decode_kmers(km, 2) == ["AC"] and ct.tolist() == [2]
    # order A<C<T<G: canonical of 'TG' vs rc 'CA' -> 'CA'
    assert orc.canonical_ascii("TG")[0] == "CA"
    # 'AT' vs rc 'AT' (palindrome, even k) ; 'GC' rc 'GC'
    assert orc.canonical_ascii("AT")[0] == "AT"
    # T<G: 'AG' vs rc 'CT' -> 'AG' ; 'TA' rc 'TA'
    assert orc.canonical_ascii("AG")[0] == "AG"
    assert orc.canonical_ascii("GA")[0] == "TC"   # rc(GA)=TC ; T(2)<G(3)
