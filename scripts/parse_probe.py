"""time of the parse stage alone on the headline input: python scripts/parse_probe.py [genomes]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import grm_amd
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pg = synth.PanGenome(genome_len=5_000_000, seed=1234)
with grm_amd.Context(0) as ctx:
    for kv in sys.argv[2:]:
        a, v = kv.split("=")
        ctx.set_option(a, int(v))
    b = ctx.batch(n)
    for i in range(n):
        b.add_array(i, pg.genome(i))
    b.upload()
    ctx.timing(True)
    for it in range(3):
        ctx.timing_reset()
        try:
            b.partition(31, 1)
        except Exception as e:
            print("partition:", str(e)[:100])
        print({name: round(ms, 3) for name, ms, _ in ctx.timings() if name.startswith("parse")})
