"""Pure-Python mirror of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see grm_oracle.h).

Written independently of grm_oracle.c on purpose: it works on *strings* (reverse
complement by translation, canonical choice by comparing strings under the GATB
nucleotide order A<C<T<G) so that an arithmetic slip in the 2-bit C code cannot be
mirrored here.  Pure loops: only for small cases.

Semantics follow SURVEY.md 8(c) ([EXT] GATB-core 1.4.2 / Kover kmer_tools), call sites:
bin/kover/core/kover/dataset/tools/kmer_count.py:28-53, tools/kmer_pack.py:28-36,
bit layout bin/kover/core/kover/utils.py:133-156.  "parity unpinned" for [EXT] rules.
"""
from collections import Counter

_COMP = str.maketrans("ACGT", "TGCA")
_ORDER = str.maketrans("ACTG", "0123")        # GATB: A<C<T<G
_FROM_CODE = "ACTG"


def base_code(ch):
    return (ord(ch) >> 1) & 3


def base_bad(ch):
    return (ord(ch) >> 3) & 1


def normalise(ch):
    """the letter a (non-bad) byte aliases to under the GATB code"""
    return _FROM_CODE[base_code(ch)]


def revcomp(s):
    return s.translate(_COMP)[::-1]


def order_key(s):
    return s.translate(_ORDER)


def canonical(s):
    r = revcomp(s)
    return s if order_key(s) <= order_key(r) else r


def kmer_value(s):
    v = 0
    for ch in s:
        v = (v << 2) | base_code(ch)
    return v


def records(text):
    """yield the sequence string of every record of a FASTA / 4-line FASTQ image"""
    stripped = text.lstrip("\n\r \t")
    fastq = stripped.startswith("@")
    lines = text.split("\n")
    if text.endswith("\n"):
        lines = lines[:-1]
    if fastq:
        cur = None
        for i, ln in enumerate(lines):
            if i % 4 == 0:
                if cur is not None:
                    yield cur
                cur = ""
            elif i % 4 == 1:
                cur += ln.replace("\r", "")
        if cur is not None:
            yield cur
    else:
        cur = ""
        for ln in lines:
            if ln.startswith(">"):
                yield cur
                cur = ""
            else:
                cur += ln.replace("\r", "")
        yield cur


def count_genome(texts, k, abundance_min=1):
    """texts: list of file images (str) of ONE genome -> sorted [(kmer_str, count)]"""
    cnt = Counter()
    n_occ = 0
    for text in texts:
        for seq in records(text):
            for i in range(len(seq) - k + 1):
                w = seq[i:i + k]
                if any(base_bad(ch) for ch in w):
                    continue
                w = "".join(normalise(ch) for ch in w)
                cnt[canonical(w)] += 1
                n_occ += 1
    out = [(km, c) for km, c in cnt.items() if c >= abundance_min]
    out.sort(key=lambda kc: order_key(kc[0]))
    return out, n_occ


def build_matrix(sets, filter_singleton):
    """sets: list (per genome) of [(kmer, count)] -> (kmers sorted, rows of python ints)

    matrix[r][c] has bit 63-(g%64) set iff genome g=64r+... carries kmers[c]
    (utils.py:133-156)."""
    n = len(sets)
    carriers = {}
    for g, s in enumerate(sets):
        for km, _ in s:
            carriers.setdefault(km, []).append(g)
    kmers = [km for km, gs in carriers.items() if not (filter_singleton and len(gs) == 1)]
    kmers.sort(key=order_key)
    n_rows = (n + 63) // 64
    matrix = [[0] * len(kmers) for _ in range(n_rows)]
    for c, km in enumerate(kmers):
        for g in carriers[km]:
            matrix[g // 64][c] |= 1 << (63 - (g % 64))
    return kmers, matrix
