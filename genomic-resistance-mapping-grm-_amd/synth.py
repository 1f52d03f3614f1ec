"""Seeded synthetic genomes for benchmarks and size-independent tests.

mode "P" (pan-genome): one random ancestor, a pool of SNP sites and accessory contigs;
    each genome carries each variant with a per-variant probability ~ Beta(0.5, 0.5),
    so genomes share most k-mers and the union stays ~2x one genome (like the real
    datasets in the reference's page/results/summary.json).
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


class PanGenome:
    def __init__(self, genome_len=5_000_000, n_snps=50_000, n_accessory=200, accessory_len=5_000, seed=1234,
                 n_contigs=1):
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

    def genome(self, idx):
        """FASTA image (uint8 array) of genome idx"""
        rng = np.random.default_rng(self.seed + 1 + idx)
        seq = self.ancestor.copy()
        if self.snp_pos.size:
            carry = rng.random(self.snp_pos.size) < self.snp_p
            seq[self.snp_pos[carry]] = self.snp_alt[carry]
        parts = []
        bounds = np.linspace(0, seq.size, self.n_contigs + 1).astype(np.int64)
        for c in range(self.n_contigs):
            parts.append(_fasta_record("g%05d_c%d" % (idx, c), seq[bounds[c]:bounds[c + 1]]))
        if self.acc:
            carry = rng.random(len(self.acc)) < self.acc_p
            for a in np.nonzero(carry)[0]:
                parts.append(_fasta_record("g%05d_acc%d" % (idx, a), self.acc[a]))
        return np.concatenate(parts)


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
