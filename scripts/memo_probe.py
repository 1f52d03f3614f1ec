"""record memo statistics of one batch.run: python scripts/memo_probe.py <genomes> <genome_len> [realistic 0/1] [k]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import grm_amd
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
n, length = int(sys.argv[1]), int(sys.argv[2])
real = int(sys.argv[3]) if len(sys.argv) > 3 else 1
k = int(sys.argv[4]) if len(sys.argv) > 4 else 31
pg = synth.realistic(genome_len=length, seed=1234) if real else synth.PanGenome(genome_len=length, seed=1234)
with grm_amd.Context(0) as ctx:
    ctx.set_option("memo_stats", 1)
    b = ctx.batch(n)
    for i in range(n):
        b.add_array(i, pg.genome(i))
    b.upload()
    ctx.timing(True)
    for it in range(2):
        ctx.timing_reset()
        m = b.run(k, 1, True)
        print("columns", m.n_kmers, "occurrences", b.n_occurrences, "memo", b.memo_stats())
        m.free()
    for name, ms, units in ctx.timings():
        print("  %-20s %8.3f ms  %d" % (name, ms, units))
