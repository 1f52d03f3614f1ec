"""Ray drop-in at scale: N genome files -> cli/Ray (survey.conf grammar of src/app.py:3812-3835) ->
Surveyor/KmerMatrix.tsv, checked against the fused engine result: row count, fixed row length
(create.py:130-137), and the first / last / 1000 sampled rows cell by cell.
Usage: python scripts/tsv_scale_check.py [n_genomes] [genome_len]"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module
import numpy as np
import grm_amd
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
d = tempfile.mkdtemp(prefix="grm_tsv_")
try:
    pg = synth.PanGenome(genome_len=L, seed=99)
    arrays = []
    conf = ["-k 31", "-run-surveyor", "-output %s" % os.path.join(d, "survey.res"), "-write-kmer-matrix"]
    for g in range(n):
        a = pg.genome(g)
        arrays.append(a)
        p = os.path.join(d, "g%04d.fna" % g)
        a.tofile(p)
        conf.append("-read-sample-assembly g%04d %s" % (g, p))
    open(os.path.join(d, "survey.conf"), "w").write("\n".join(conf) + "\n")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "genomic-resistance-mapping-grm-_amd", "cli", "Ray"), os.path.join(d, "survey.conf")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    t_cli = time.time() - t0
    tsv = os.path.join(d, "survey.res", "Surveyor", "KmerMatrix.tsv")
    size = os.path.getsize(tsv)
    with grm_amd.Context(0) as ctx:
        b = ctx.batch(n)
        for g in range(n):
            b.add_array(g, arrays[g])
        b.upload()
        m = b.run(31, 1, False)
        kmers, data = m.kmers(), m.data()
        U = m.n_kmers
        m.free(); b.free()
    names = grm_amd.decode_kmers(kmers, 31)
    with open(tsv, "rb") as f:
        header = f.readline()
        assert header.decode().rstrip("\n").split("\t") == ["kmers"] + ["g%04d" % g for g in range(n)], header[:80]
        row_len = 31 + 2 * n + 1
        assert size == len(header) + U * row_len, (size, len(header), U, row_len)      # every row the same length
        rng = np.random.default_rng(0)
        for c in [0, U - 1] + list(rng.integers(0, U, size=1000)):
            f.seek(len(header) + int(c) * row_len)
            cells = f.read(row_len).decode().rstrip("\n").split("\t")
            assert cells[0] == names[c], (c, cells[0], names[c])
            bits = [(int(data[g // 64, c]) >> (63 - g % 64)) & 1 for g in range(n)]
            assert cells[1:] == [str(x) for x in bits], c
    print({"genomes": n, "columns": int(U), "tsv_GB": round(size / 1e9, 2), "cli_wall_s": round(t_cli, 2)})
    print("TSV OK")
finally:
    shutil.rmtree(d, ignore_errors=True)
