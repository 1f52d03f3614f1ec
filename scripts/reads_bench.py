"""C4-like measurement (scaled): G genomes x paired-free 150 bp reads at coverage X with 0.5 % substitution
errors, 4-line FASTQ, k=21, abundance-min 2 -> counted sets per genome (multidsk) -> matrix (dsk2kover).
Usage: python scripts/reads_bench.py [genomes] [genome_len] [coverage]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import numpy as np
import grm_amd
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
X = int(sys.argv[3]) if len(sys.argv) > 3 else 30
RL = 150
pg = synth.PanGenome(genome_len=L, seed=1234)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

def reads_fastq(g):
    rng = np.random.default_rng(77 + g)
    fa = pg.genome(g)
    seq = fa[(fa != 10)]
    seq = seq[np.isin(seq, ACGT)][:L]                       # sequence letters only (headers are short, dropped approx.)
    n = L * X // RL
    st = rng.integers(0, len(seq) - RL, size=n)
    win = seq[st[:, None] + np.arange(RL)[None, :]]
    err = rng.random(win.shape) < 0.005
    win[err] = ACGT[rng.integers(0, 4, size=int(err.sum()))]
    rec = np.empty((n, 3 + RL + 3 + RL + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + RL] = win
    rec[:, 3 + RL:6 + RL] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + RL:6 + 2 * RL] = ord("I")
    rec[:, -1] = 10
    return rec.reshape(-1)

t0 = time.time()
fq = [reads_fastq(g) for g in range(G)]
print("generated %d read sets (%.2f GB) in %.1fs" % (G, sum(f.size for f in fq) / 1e9, time.time() - t0), flush=True)
with grm_amd.Context(0) as ctx:
    ctx.timing(True)
    for rep in range(2):
        ctx.timing_reset()
        t0 = time.time()
        b = ctx.batch(G)
        for g in range(G):
            b.add_array(g, fq[g])
        b.upload()
        t1 = time.time()
        b.partition_counts(21, 2)
        sets = [b.genome_set(g) for g in range(G)]
        t2 = time.time()
        m = ctx.build_matrix(sets, True)
        t3 = time.time()
        occ = b.n_occurrences
        print({"rep": rep, "occurrences": occ, "upload_s": round(t1 - t0, 2), "count_s": round(t2 - t1, 3), "merge_s": round(t3 - t2, 3),
               "kmers_per_s_count": round(occ / (t2 - t1)), "solid_per_genome": int(np.mean([len(s) for s in sets])), "columns": m.n_kmers}, flush=True)
        per = {}
        for name, ms, units in ctx.timings():
            per[name] = per.get(name, 0) + ms
        print({k: round(v, 2) for k, v in per.items() if v > 0.5})
        m.free()
        for s in sets:
            s.free()
        b.free()
