"""a pan-genome far more diverse than bench.py's (many SNP sites): the dictionary tables of the record form overflow at the
default bucket count, the sizing ladder takes more buckets (level 2 again) -- prints the kernels of the first and a later pass"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
grm = importlib.import_module("genomic-resistance-mapping-grm-_amd")
synth = importlib.import_module("genomic-resistance-mapping-grm-_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
ctx = grm.Context(0)
pg = synth.PanGenome(genome_len=5_000_000, n_snps=snps, seed=99)
b = ctx.batch(n)
for i in range(n):
    b.add_array(i, pg.genome(i))
b.upload()
ctx.timing(True)
for it in range(3):
    ctx.timing_reset()
    t0 = time.time()
    m = b.run(31, 1, True)
    dt = time.time() - t0
    per = {}
    for name, ms, units in ctx.timings():
        d = per.setdefault(name, [0.0, 0])
        d[0] += ms; d[1] += 1
    print("pass %d: %.1f ms wall, columns %d, bucket geometry 0x%x" % (it, dt * 1e3, m.n_kmers, b.bucket_bits),
          {k: (round(v[0], 2), v[1]) for k, v in per.items()}, flush=True)
    m.free()
