"""The multidsk -> dsk2kover pair exactly as Kover drives it (kmer_count.py:28-37, kmer_pack.py:28-36),
wall clock of both calls.  Usage: python scripts/e2e_pair.py [n_genomes] [genome_len] [sets]"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module
import grm_amd  # noqa
synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
sets = len(sys.argv) > 3 and sys.argv[3] == "sets"
CLI = os.path.join(ROOT, "genomic-resistance-mapping-grm-_amd", "cli")
d = tempfile.mkdtemp(prefix="grm_pair_")
try:
    pg = synth.PanGenome(genome_len=L, seed=1234)
    paths = []
    for g in range(n):
        p = os.path.join(d, "g%05d.fna" % g)
        pg.genome(g).tofile(p)
        paths.append(p)
    lst = os.path.join(d, "list_contigs_files")
    open(lst, "w").writelines(p + "\n" for p in paths)
    env = dict(os.environ)
    if sets:
        env["GRM_MULTIDSK_SETS"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(CLI, "multidsk"), "-file", lst, "-out-dir", d, "-kmer-size", "31", "-abundance-min", "1",
                        "-out-compress", "4", "-nb-cores", "0", "-out-tmp", d, "-verbose", "0", "-progress", "True"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    t_count = time.time() - t0
    inter = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith((".h5", ".matrix")))
    h5s = [os.path.join(d, os.path.basename(os.path.splitext(p + "\n")[0]) + ".h5") for p in paths]
    list_h5 = os.path.join(d, "list_h5")
    open(list_h5, "w").writelines(p + "\n" for p in h5s)
    out = os.path.join(d, "DATASET.kover")
    kd.write_header(out, "contigs", lst, None, None, 4, ["g%05d" % g for g in range(n)], None, None, None, "singleton")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(CLI, "dsk2kover"), "-file", list_h5, "-out", out, "-filter", "singleton", "-kmer-length", "31",
                        "-compression", "4", "-chunk-size", "100000", "-nb-genomes", str(n), "-verbose", "True"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    t_pack = time.time() - t0
    print(r.stdout.strip()[-200:])
    print({"genomes": n, "mode": "per-genome sets" if sets else "combined artefact", "multidsk_s": round(t_count, 2), "dsk2kover_s": round(t_pack, 2),
           "intermediate_GB": round(inter / 1e9, 2), "kover_MB": os.path.getsize(out) >> 20})
finally:
    shutil.rmtree(d, ignore_errors=True)
