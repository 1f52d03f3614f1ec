"""Shared seeded micro-genome cases (inputs only) used by oracle and GPU parity tests."""
import numpy as np


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), size=n))


def fasta(records, width=80, crlf=False):
    nl = "\r\n" if crlf else "\n"
    out = []
    for name, seq in records:
        out.append(">" + name + nl)
        for i in range(0, len(seq), width):
            out.append(seq[i:i + width] + nl)
    return "".join(out)


_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def revcomp(s):
    return s.translate(_COMP)[::-1]


def micro_cases():
    """-> list of (name, k, genomes) ; genomes = list of list[str] file images"""
    rng = np.random.RandomState(1234)
    cases = []
    a = rand_seq(rng, 150)
    b = rand_seq(rng, 90)
    # plain, multi-contig, a contig shorter than k, reverse-complement duplicate contig
    g0 = fasta([("c1", a), ("c2", b), ("short", "ACGTAC"), ("rc", revcomp(a))], width=60)
    # N runs, lowercase, IUPAC letters (R,Y: alias; K,M: bad), CRLF
    g1 = fasta([("n", a[:40] + "NNN" + a[40:80] + "n" + a[80:]),
                ("lower", b.lower()),
                ("iupac", a[:30] + "R" + a[31:60] + "K" + a[61:100] + "Y" + a[101:])], width=70, crlf=True)
    # no trailing newline, single-line sequence, blank lines
    g2 = ">x\n" + a + "\n\n>y\n\n" + b[:50] + "\n" + b[50:]
    # a shared core with SNPs
    core = rand_seq(rng, 200)
    mut = list(core)
    for p in (17, 77, 140):
        mut[p] = "ACGT"[("ACGT".index(mut[p]) + 1) % 4]
    g3 = fasta([("core", core)], width=80)
    g4 = fasta([("core_mut", "".join(mut))], width=80)
    # header-only / empty
    g5 = ">empty\n"
    g6 = ""
    # two files for one genome (k-mers must not span files)
    g7 = [fasta([("p1", a[:75])], width=80), fasta([("p2", a[75:])], width=80)]
    genomes = [[g0], [g1], [g2], [g3], [g4], [g5], [g6], g7]
    for k in (1, 2, 3, 5, 11, 16, 17, 21, 31, 32):
        cases.append(("micro_k%d" % k, k, genomes))
    # low-complexity: heavy duplicates
    poly = fasta([("polyA", "A" * 300), ("at", "AT" * 150), ("mix", "ACG" * 100)], width=80)
    cases.append(("lowcomplex_k15", 15, [[poly], [g3], [poly]]))
    return cases


def fastq(reads):
    out = []
    for i, r in enumerate(reads):
        out.append("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
    return "".join(out)


def fastq_cases():
    rng = np.random.RandomState(99)
    ref = rand_seq(rng, 400)
    reads1 = [ref[s:s + 60] for s in rng.randint(0, 340, size=80)]
    reads2 = [revcomp(ref[s:s + 60]) for s in rng.randint(0, 340, size=80)]
    reads2[3] = reads2[3][:20] + "N" + reads2[3][21:]
    # '@' and '>' as first quality character must not confuse the parser
    fq = fastq(reads1)
    fq_tricky = "@t\n%s\n+\n%s\n" % (ref[:50], "@" + ">" * 49) + "@t2\n%s\n+\n%s\n" % (ref[100:150], ">" * 50)
    return [("fastq_k21", 21, [[fq, fastq(reads2)], [fq_tricky]])]
