"""How compressible is the real kmer_matrix, and with what?  (input to the HDF5-writer design)
Runs the default pan-genome batch, takes a few (1, 100000)-column chunks of the matrix and
deflates them with zlib at several levels / strategies.  Usage: python scripts/deflate_study.py [genomes]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import numpy as np
import grm_amd
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pg = synth.PanGenome(genome_len=5_000_000, seed=1234)
with grm_amd.Context(0) as ctx:
    b = ctx.batch(n)
    for g in range(n):
        b.add_array(g, pg.genome(g))
    b.upload()
    m = b.run(31, 1, True)
    data = m.data()
    m.free(); b.free()
rows, U = data.shape
chunks = [data[r, c0:c0 + 100000].tobytes() for r in (0, rows // 2, rows - 1) for c0 in (0, (U // 2) // 100000 * 100000)]
raw = sum(len(c) for c in chunks)
words = np.concatenate([np.frombuffer(c, dtype=np.uint64) for c in chunks])
print({"rows": rows, "U": U, "sample_MB": raw / 1e6, "all_ones_words": float((words == np.uint64(2**64 - 1)).mean()),
       "zero_words": float((words == 0).mean()),
       "popcount_mean": float(np.unpackbits(words.view(np.uint8)).mean() * 64)})
for name, level, strat in [("level1", 1, zlib.Z_DEFAULT_STRATEGY), ("level4", 4, zlib.Z_DEFAULT_STRATEGY), ("level5", 5, zlib.Z_DEFAULT_STRATEGY),
                           ("level9", 9, zlib.Z_DEFAULT_STRATEGY), ("rle", 4, zlib.Z_RLE), ("huffman_only", 4, zlib.Z_HUFFMAN_ONLY),
                           ("fixed+default", 4, zlib.Z_FIXED), ("filtered", 4, zlib.Z_FILTERED)]:
    t0 = time.time()
    z = 0
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strat)
        z += len(co.compress(c)) + len(co.flush())
    dt = time.time() - t0
    print("%-14s ratio %.3f  %.0f MB/s" % (name, z / raw, raw / dt / 1e6))
