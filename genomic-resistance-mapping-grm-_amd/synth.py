"""Seeded synthetic genomes for benchmarks and size-independent tests.

mode "P" (pan-genome): one random ancestor, a pool of SNP sites and accessory contigs;
    each genome carries each variant with a per-variant probability ~ Beta(0.5, 0.5),
    so genomes share most k-mers and the union stays ~2x one genome (like the real
    datasets in the reference's page/results/summary.json).
    "realistic" assemblies (PanGenome(indel_sites=..., contigs=(lo, hi), shuffle_contigs=True, random_strand=True)): what
    GRM's inputs look like (src/app.py:576-583: one BV-BRC multi-contig assembly per genome, src/kover.py:40-49) -- besides
    the SNPs a pool of short insertions / deletions, every genome cut into its own 20-100 contigs at its own places, the
    contigs in random order and on a random strand.  No two genomes share a coordinate frame.
mode "R" (independent): i.i.d. uniform ACGT per genome (every k-mer a singleton).
Output: FASTA images as numpy uint8 arrays, 80-column lines, uppercase.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _fasta_record(name, seq_u8, width=80):
    """seq_u8: uint8 array of letters -> uint8 array of one FASTA record"""
    n = seq_u8.size
    head = np.frombuffer((">%s\n" % name).encode(), dtype=np.uint8)
    full = n // width
    body = np.empty(full * (width + 1), dtype=np.uint8)
    if full:
        b2 = body.reshape(full, width + 1)
        b2[:, :width] = seq_u8[: full * width].reshape(full, width)
        b2[:, width] = 10
    tail = seq_u8[full * width:]
    parts = [head, body]
    if tail.size:
        parts += [tail, np.array([10], dtype=np.uint8)]
    return np.concatenate(parts)


_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGTN")] = list(b"TGCAN")


def revcomp_u8(seq_u8):
    return _COMP[seq_u8[::-1]]


class PanGenome:
    def __init__(self, genome_len=5_000_000, n_snps=50_000, n_accessory=200, accessory_len=5_000, seed=1234,
                 n_contigs=1, indel_sites=0, max_indel=10, contigs=None, shuffle_contigs=False, random_strand=False):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.genome_len = genome_len
        self.n_contigs = max(1, n_contigs)
        self.ancestor = _ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
        n_snps = min(n_snps, genome_len)
        self.snp_pos = np.sort(rng.choice(genome_len, size=n_snps, replace=False)) if n_snps else np.zeros(0, np.int64)
        shift = rng.integers(1, 4, size=n_snps, dtype=np.uint8)
        anc_code = np.searchsorted(_ACGT, self.ancestor[self.snp_pos]) if n_snps else np.zeros(0, np.int64)
        self.snp_alt = _ACGT[(anc_code + shift) % 4] if n_snps else np.zeros(0, np.uint8)
        self.snp_p = rng.beta(0.5, 0.5, size=n_snps)
        self.acc = [_ACGT[rng.integers(0, 4, size=accessory_len, dtype=np.uint8)] for _ in range(n_accessory)]
        self.acc_p = rng.beta(0.5, 0.5, size=n_accessory)
        # the "realistic" options draw from streams of their own: the defaults give the same genomes as before
        self.contigs, self.shuffle_contigs, self.random_strand = contigs, shuffle_contigs, random_strand
        rng2 = np.random.default_rng([seed, 7919])
        indel_sites = min(int(indel_sites), genome_len // (4 * max_indel + 4))
        # indel sites keep 2 * max_indel apart, so that deletions never overlap the next site
        slots = np.sort(rng2.choice(genome_len // (2 * max_indel + 2) - 1, size=indel_sites, replace=False)) if indel_sites else np.zeros(0, np.int64)
        self.indel_pos = (slots + 1) * (2 * max_indel + 2)
        self.indel_len = rng2.integers(1, max_indel + 1, size=indel_sites)
        self.indel_is_ins = rng2.random(indel_sites) < 0.5
        self.indel_ins = [_ACGT[rng2.integers(0, 4, size=int(n), dtype=np.uint8)] for n in self.indel_len]
        self.indel_p = rng2.beta(0.5, 0.5, size=indel_sites)

    def genome(self, idx):
        """FASTA image (uint8 array) of genome idx"""
        rng = np.random.default_rng(self.seed + 1 + idx)
        seq = self.ancestor.copy()
        if self.snp_pos.size:
            carry = rng.random(self.snp_pos.size) < self.snp_p
            seq[self.snp_pos[carry]] = self.snp_alt[carry]
        rng2 = np.random.default_rng([self.seed, 104729, idx])
        if self.indel_pos.size:
            carry = np.nonzero(rng2.random(self.indel_pos.size) < self.indel_p)[0]
            pieces, at = [], 0
            for s_ in carry:
                p = int(self.indel_pos[s_])
                pieces.append(seq[at:p])
                if self.indel_is_ins[s_]:
                    pieces.append(self.indel_ins[s_])
                    at = p
                else:
                    at = p + int(self.indel_len[s_])
            pieces.append(seq[at:])
            seq = np.concatenate(pieces)
        if self.contigs:
            n_contigs = int(rng2.integers(self.contigs[0], self.contigs[1] + 1))
            cuts = np.unique(rng2.integers(1, seq.size, size=n_contigs - 1)) if n_contigs > 1 else np.zeros(0, np.int64)
            n_contigs = cuts.size + 1
            bounds = np.concatenate(([0], cuts, [seq.size])).astype(np.int64)
        else:
            n_contigs = self.n_contigs
            bounds = np.linspace(0, seq.size, self.n_contigs + 1).astype(np.int64)
        contigs = [seq[bounds[c]:bounds[c + 1]] for c in range(n_contigs)]
        if self.acc:
            carry = rng.random(len(self.acc)) < self.acc_p
            names = ["g%05d_c%d" % (idx, c) for c in range(n_contigs)] + ["g%05d_acc%d" % (idx, a) for a in np.nonzero(carry)[0]]
            contigs += [self.acc[a] for a in np.nonzero(carry)[0]]
        else:
            names = ["g%05d_c%d" % (idx, c) for c in range(n_contigs)]
        if self.random_strand:
            flip = rng2.random(len(contigs)) < 0.5
            contigs = [revcomp_u8(c) if f else c for c, f in zip(contigs, flip)]
        order = rng2.permutation(len(contigs)) if self.shuffle_contigs else np.arange(len(contigs))
        return np.concatenate([_fasta_record(names[i], contigs[i]) for i in order])


def realistic(genome_len=5_000_000, seed=1234, **kw):
    """PanGenome as GRM's inputs are: SNPs + ~1 indel site per 10 kbp, 20-100 contigs per genome cut at the genome's own
    places, in random order and on random strands"""
    args = dict(genome_len=genome_len, seed=seed, indel_sites=genome_len // 10_000, contigs=(20, 100), shuffle_contigs=True,
                random_strand=True)
    args.update(kw)
    return PanGenome(**args)


# one random byte -> four letters (two bits each, low bits first)
_LUT4 = np.array([int(_ACGT[b & 3]) | (int(_ACGT[(b >> 2) & 3]) << 8) | (int(_ACGT[(b >> 4) & 3]) << 16) | (int(_ACGT[(b >> 6) & 3]) << 24)
                  for b in range(256)], dtype="<u4")


def random_genome(idx, genome_len=5_000_000, seed=1234, n_contigs=1):
    """i.i.d. uniform ACGT: the bytes of PCG64(seed + idx), two bits per base"""
    rng = np.random.default_rng(seed + idx)
    raw = np.frombuffer(rng.bytes((genome_len + 3) // 4), dtype=np.uint8)
    seq = _LUT4[raw].view(np.uint8)[:genome_len]
    bounds = np.linspace(0, seq.size, max(1, n_contigs) + 1).astype(np.int64)
    return np.concatenate([_fasta_record("r%05d_c%d" % (idx, c), seq[bounds[c]:bounds[c + 1]]) for c in range(max(1, n_contigs))])


class EcoliLike:
    """BASELINE C1: a handful of contig sets derived from one 4.6 Mbp random ancestor -- per genome 0.5 % private
    SNPs, 20-100 contigs of uneven length, a few runs of N inside contigs, 80-column FASTA (seeds 1234 + i)"""

    def __init__(self, genome_len=4_600_000, snp_rate=0.005, seed=1234, n_runs=4):
        self.seed, self.genome_len, self.snp_rate, self.n_runs = seed, genome_len, snp_rate, n_runs
        self.ancestor = _ACGT[np.random.default_rng(seed).integers(0, 4, size=genome_len, dtype=np.uint8)]

    def genome(self, idx):
        rng = np.random.default_rng(self.seed + 1 + idx)
        seq = self.ancestor.copy()
        n_snps = int(self.genome_len * self.snp_rate)
        pos = rng.choice(self.genome_len, size=n_snps, replace=False)
        code = np.searchsorted(_ACGT, seq[pos])
        seq[pos] = _ACGT[(code + rng.integers(1, 4, size=n_snps)) % 4]
        for _ in range(self.n_runs):                                   # runs of N (bad bases: k-mers over them are skipped)
            a = int(rng.integers(0, self.genome_len - 300))
            seq[a:a + int(rng.integers(1, 200))] = ord("N")
        n_contigs = int(rng.integers(20, 101))
        cuts = np.sort(rng.choice(np.arange(1, self.genome_len), size=n_contigs - 1, replace=False))
        bounds = np.concatenate(([0], cuts, [self.genome_len]))
        return np.concatenate([_fasta_record("e%03d_contig%d" % (idx, c), seq[bounds[c]:bounds[c + 1]]) for c in range(n_contigs)])


def make_genomes(n, mode="P", genome_len=5_000_000, seed=1234, **kw):
    """-> list of uint8 arrays (one FASTA image per genome)"""
    if mode == "P":
        pg = PanGenome(genome_len=genome_len, seed=seed, **kw)
        return [pg.genome(i) for i in range(n)]
    return [random_genome(i, genome_len=genome_len, seed=seed) for i in range(n)]
