"""End-to-end wall clock of the reference's main scenario through the drop-in CLI:
N synthetic genome files on disk -> `kover dataset create from-contigs` (-> .kover, gzip 4),
next to the CPU restatement (oracle) doing count + merge on all host cores from the same files.
Usage: python scripts/e2e_cli.py [n_genomes] [genome_len]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
import grm_amd  # noqa
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
d = tempfile.mkdtemp(prefix="grm_e2e_")
pg = synth.PanGenome(genome_len=L, seed=1234)
t0 = time.time()
paths = []
for g in range(n):
    p = os.path.join(d, "g%05d.fna" % g)
    pg.genome(g).tofile(p)
    paths.append(p)
print("wrote %d files in %.1fs" % (n, time.time() - t0), flush=True)
data = os.path.join(d, "paths.tsv")
open(data, "w").writelines("g%05d\t%s\n" % (g, p) for g, p in enumerate(paths))
md = os.path.join(d, "md.tsv")
open(md, "w").writelines("g%05d\t%d\n" % (g, g % 2) for g in range(n))
out = os.path.join(d, "DATASET.kover")
t0 = time.time()
r = subprocess.run([sys.executable, os.path.join(ROOT, "genomic-resistance-mapping-grm-_amd", "cli", "kover"), "dataset", "create", "from-contigs",
                    "--genomic-data", data, "--phenotype-description", "d", "--phenotype-metadata", md, "--output", out,
                    "--kmer-size", "31", "--compression", "4", "-x"], capture_output=True, text=True)
t_gpu = time.time() - t0
print(r.stdout[-1500:], r.stderr[-500:])
print("GPU CLI wall clock: %.2fs  (.kover %d MB)" % (t_gpu, os.path.getsize(out) >> 20), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "cpu":
    from oracle import oracle_ctypes as orc
    t0 = time.time()
    bufs = [open(p, "rb").read() for p in paths]
    t_read = time.time() - t0
    cores = min(os.cpu_count() or 1, 64)
    res, cs, ms, occ = orc.pipeline(bufs, 31, 1, True, cores)
    print("CPU restatement (%d cores): read %.1fs + count %.1fs + merge %.1fs (no HDF5 written); %d columns" % (cores, t_read, cs, ms, res["kmers"].shape[0]))
subprocess.run(["rm", "-rf", d])
