"""Times the host-side writers on a real GPU result (C5-like: 500 genomes, k=63, singletons kept,
HDF5 gzip 5) -- SURVEY 8(f) item 2.  Usage: python scripts/time_writers.py [genomes] [k]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import grm_amd
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
k = int(sys.argv[2]) if len(sys.argv) > 2 else 63
pg = synth.PanGenome(genome_len=5_000_000, seed=1234)
with grm_amd.Context(0) as ctx:
    b = ctx.batch(n)
    for g in range(n):
        b.add_array(g, pg.genome(g))
    t0 = time.time(); b.upload(); t_up = time.time() - t0
    t0 = time.time(); m = b.run(k, 1, False); t_run = time.time() - t0
    t0 = time.time(); m.data(); m.kmers(); t_d2h = time.time() - t0
    d = tempfile.mkdtemp()
    path = os.path.join(d, "c5.kover")
    ids = ["g%d" % i for i in range(n)]
    kd.write_header(path, "contigs", "l", None, None, 5, ids, None, None, None, "nothing")
    t0 = time.time(); m.write_kover_h5(path, 5, 100000); t_h5 = time.time() - t0
    size = os.path.getsize(path)
    print({"genomes": n, "k": k, "columns": m.n_kmers, "upload_s": round(t_up, 2), "gpu_pass_s": round(t_run, 3), "d2h_s": round(t_d2h, 2),
           "h5_gzip5_s": round(t_h5, 2), "h5_bytes": size, "matrix_bytes": m.n_kmers * m.n_rows * 8,
           "h5_MBps_uncompressed": round((m.n_kmers * (m.n_rows * 8 + k)) / t_h5 / 1e6, 1)})
    r = kd.KoverDatasetReader(path)
    assert r.layout("kmer_matrix")["chunks"] == (1, 100000)
    m.free(); b.free()
