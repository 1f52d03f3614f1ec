"""Deep read sets through the chunked multidsk -> dsk2kover flow (C4-like, few genomes at full depth):
G genomes x 150 bp reads at coverage X with 0.5 % substitution errors -> 4-line FASTQ files on disk ->
kover_dataset.counted_sets (chunks by GRM_BATCH_BYTES; deep mode per chunk) -> build_matrix.
One genome's solid set is checked against the CPU oracle.
Usage: python scripts/reads_depth_check.py [genomes] [genome_len] [coverage] [k] [abundance_min]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import numpy as np
import grm_amd
from oracle import oracle_ctypes as orc
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
X = int(sys.argv[3]) if len(sys.argv) > 3 else 100
K = int(sys.argv[4]) if len(sys.argv) > 4 else 21
AMIN = int(sys.argv[5]) if len(sys.argv) > 5 else 2
RL = 150
pg = synth.PanGenome(genome_len=L, seed=1234)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

def reads_fastq(g):
    rng = np.random.default_rng(77 + g)
    fa = pg.genome(g)
    seq = fa[(fa != 10)]
    seq = seq[np.isin(seq, ACGT)][:L]
    n = L * X // RL
    rec = np.empty((n, 3 + RL + 3 + RL + 1), dtype=np.uint8)
    for a in range(0, n, 1 << 20):                       # blocks: bounded temporaries
        b = min(n, a + (1 << 20))
        st = rng.integers(0, len(seq) - RL, size=b - a)
        win = seq[st[:, None] + np.arange(RL)[None, :]]
        err = rng.random(win.shape) < 0.005
        win[err] = ACGT[rng.integers(0, 4, size=int(err.sum()))]
        rec[a:b, 3:3 + RL] = win
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3 + RL:6 + RL] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + RL:6 + 2 * RL] = ord("I")
    rec[:, -1] = 10
    return rec.reshape(-1)

d = tempfile.mkdtemp(prefix="grm_reads_")
try:
    t0 = time.time()
    files = []
    for g in range(G):
        p = os.path.join(d, "g%03d.fastq" % g)
        reads_fastq(g).tofile(p)
        files.append([p])
        print("wrote %s (%.2f GB) %.0fs" % (p, os.path.getsize(p) / 1e9, time.time() - t0), flush=True)
    with grm_amd.Context(0) as ctx:
        ctx.timing(True)
        t0 = time.time()
        chunks = kd.plan_chunks(files, kd.DEFAULT_BATCH_BYTES)
        sets = kd.counted_sets(ctx, files, K, AMIN, kd.DEFAULT_BATCH_BYTES, lambda m: print("  " + m, flush=True))
        t1 = time.time()
        m = ctx.build_matrix(sets, True)
        t2 = time.time()
        occ = sum(s.occurrences for s in sets)
        print({"genomes": G, "coverage": X, "k": K, "abundance_min": AMIN, "chunks": [len(c) for c in chunks], "occurrences": occ,
               "count_s": round(t1 - t0, 2), "kmers_per_s": round(occ / (t1 - t0)), "merge_s": round(t2 - t1, 3),
               "solid_per_genome": [len(s) for s in sets], "columns": m.n_kmers}, flush=True)
        per = {}
        for name, ms, units in ctx.timings():
            per[name] = per.get(name, 0) + ms
        print({k: round(v, 1) for k, v in per.items() if v > 1})
        # oracle check of the last genome (single-threaded scan of ~1 GB: about a minute)
        g = G - 1
        t0 = time.time()
        km, ct, nocc = orc.count_genome([open(files[g][0], "rb").read()], K, AMIN)
        print("oracle genome %d: %d solid, %d occurrences (%.0fs)" % (g, len(ct), nocc, time.time() - t0), flush=True)
        s = sets[g]
        assert s.occurrences == nocc, (s.occurrences, nocc)
        assert s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
        cc = m.column_counts()
        assert cc.min() >= 2 and cc.max() <= G
        print("PARITY OK")
finally:
    shutil.rmtree(d, ignore_errors=True)
