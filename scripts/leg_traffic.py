"""HBM traffic of the bench legs other than the headline: merges the PMC summaries scripts/pmc_hbm_leg.sh left in gpurun_out/
(pmc_<tag>_<leg>_summary.csv) into profiles/hbm_traffic.json under "legs" and copies them to profiles/<round>/.
    python scripts/leg_traffic.py <tag, e.g. r04> <round dir, e.g. r04> [<prefix, default final>]
Nothing here is measured: every number is read from the rocprofv3 output of the named sessions."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1:3]
prefix = sys.argv[3] if len(sys.argv) > 3 else "final"
G = os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
NAMES = {"superkmer_l2_records": "superkmer_l2", "superkmer_l2_wide": "superkmer_l2", "wide_dict_build": "wh_dict_build", "matrix_entry_rows": "matrix_fill",
         "matrix_transpose": "matrix_fill"}
WORKLOADS = {"c5": "500 x 5000000 bp pan-genome (mode P), k=63, singletons kept", "c4": "8 genomes x 5000000 bp, 150 bp reads at 100x, k=21, abundance-min 2",
             "random": "1000 independent uniform-ACGT genomes x 5000000 bp, k=31", "realistic": "1000 x 5000000 bp realistic assemblies, k=31"}


def short(name):
    return name.replace("void ", "").split("(")[0].replace("grm::", "").split("<")[0].replace("_kernel", "")


tj_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
tj = json.load(open(tj_path))
legs = tj.setdefault("legs", {})
for leg, key in (("c5", "c5"), ("c4", "c4"), ("random", "random_acgt"), ("realistic", "realistic")):
    src = os.path.join(G, "pmc_%s_%s_summary.csv" % (tag, leg))
    if not os.path.exists(src):
        continue
    dst = os.path.join(out, "%s_%s_pmc_hbm.csv" % (prefix, leg))
    shutil.copy(src, dst)
    # the leg's own kernels: the launches of the timed pass are the ones with the leg's launch count (a 16-genome token headline runs first)
    traffic = {}
    for r in csv.DictReader(open(dst)):
        k = NAMES.get(short(r["kernel"]), short(r["kernel"]))
        b = float(r["fetch_bytes_per_launch_x2corrected"]) + float(r["write_bytes_per_launch"])
        traffic[k] = traffic.get(k, 0) + int(b)
    legs[key] = {"workload": WORKLOADS[leg], "commit": commit,
                 "source": "profiles/%s/%s_%s_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --only %s --steps 1; FETCH_SIZE x2 "
                           "per MI355X_MICROARCH.md; per-launch averages include the token 16-genome headline pass of the same process for kernels both legs run)"
                           % (rnd, prefix, leg, leg),
                 "bytes_per_launch": traffic}
    sq = os.path.join(G, "sq_%s_%s_summary.csv" % (tag, leg))
    if os.path.exists(sq):
        shutil.copy(sq, os.path.join(out, "%s_%s_sq_counters.csv" % (prefix, leg)))
json.dump(tj, open(tj_path, "w"), indent=1)
print({k: len(v["bytes_per_launch"]) for k, v in legs.items()})
