"""BASELINE configs[3] at its own size through the command surface: N read sets (default 200) x paired-end-like 150 bp reads at 100x of a
5 Mbp genome with 0.5 % substitution errors (1 GB of FASTQ each), `kover dataset create from-reads --kmer-size 21
--kmer-min-abundance 2` (bin/kover/core/kover/dataset/create.py:399-523: chunks of genomes that fit GRM_BATCH_BYTES are counted,
the solid sets merged), timed with the drop-in's own progress stamps.  The FASTQ files are made on the GPU (bench.py's generator)
and live in /dev/shm.      python scripts/c4_full_cli.py [n_genomes] [out.json]"""
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n_g = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out_json = sys.argv[2] if len(sys.argv) > 2 else None
work = os.environ.get("GRM_C4_DIR", "/dev/shm/grm_c4")
shutil.rmtree(work, ignore_errors=True)
os.makedirs(work)


def make_inputs():
    import torch
    import bench
    from importlib import import_module
    synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
    pg = synth.PanGenome(genome_len=5_000_000, seed=1234)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    total = 0
    with open(os.path.join(work, "paths.tsv"), "w") as ft, open(os.path.join(work, "md.tsv"), "w") as fm:
        for g in range(n_g):
            fa = pg.genome(g)
            seq = fa[np.isin(fa, acgt)][:5_000_000]
            img = bench._reads_fastq(torch, dev, seq, 5_000_000 * 100 // 150, 150, 7700 + g)
            d = os.path.join(work, "reads_%03d" % g)
            os.makedirs(d)
            img.tofile(os.path.join(d, "r.fastq"))
            total += img.size
            ft.write("g%03d\t%s\n" % (g, d))
            fm.write("g%03d\t%d\n" % (g, g % 2))
            if g % 20 == 19:
                print("made %d read sets, %.1f GB, %.0f s" % (g + 1, total / 1e9, time.time() - t0), flush=True)
    del torch
    return total, time.time() - t0


if __name__ == "__main__":
    # the generator runs in a child so that this process never initialises the GPU before it starts the drop-in
    if len(sys.argv) > 3 and sys.argv[3] == "--make":
        total, secs = make_inputs()
        print(json.dumps({"fastq_bytes": total, "make_s": round(secs, 1)}), flush=True)
        sys.exit(0)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n_g), "-", "--make"], capture_output=True, text=True)
    sys.stdout.write(r.stdout[-1500:])
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(1)
    made = json.loads(r.stdout.strip().splitlines()[-1])
    kover = os.path.join(work, "C4.kover")
    cmd = [sys.executable, os.path.join(ROOT, "genomic-resistance-mapping-grm-_amd", "cli", "kover"), "dataset", "create", "from-reads", "--genomic-data",
           os.path.join(work, "paths.tsv"), "--phenotype-description", "synthetic", "--phenotype-metadata", os.path.join(work, "md.tsv"), "--output", kover,
           "--kmer-size", "21", "--kmer-min-abundance", "2", "--compression", "4", "-x"]
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    res = {"genomes": n_g, "fastq_bytes": made["fastq_bytes"], "inputs_made_s": made["make_s"], "cli_wall_s": round(wall, 2), "rc": p.returncode,
           "batch_bytes": int(os.environ.get("GRM_BATCH_BYTES", str(6 * 10**9))), "progress_first": lines[:4], "progress_last": lines[-8:],
           "stderr_tail": p.stderr[-800:]}
    if p.returncode == 0:
        import grm_amd  # noqa: F401
        from importlib import import_module
        kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
        rd = kd.KoverDatasetReader(kover)
        m = rd.kmer_matrix
        carriers = kd._popcount64(m).sum(axis=0)
        res.update({"kover_bytes": os.path.getsize(kover), "columns": int(m.shape[1]), "rows": int(m.shape[0]),
                    "min_carriers": int(carriers.min()) if carriers.size else 0, "bases_per_s": round(made["fastq_bytes"] * 150 / 307 / wall, 1),
                    "genomes_per_min": round(n_g / wall * 60, 1)})
    print(json.dumps(res, indent=1), flush=True)
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
    shutil.rmtree(work, ignore_errors=True)
    sys.exit(0 if p.returncode == 0 else 1)
