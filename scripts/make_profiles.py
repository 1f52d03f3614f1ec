"""Copies the rocprofv3 summaries of a profiling session from gpurun_out/ (scratch) into profiles/<round>/ (committed) and derives
the two files bench.py reads: profiles/hbm_traffic.json (roofline.traffic) and profiles/limits.json (roofline.limited_by).
    python scripts/make_profiles.py <tag of scripts/profile_round.sh> <tag of scripts/pmc_sq.sh> <round dir, e.g. r03> <prefix, e.g. final>
Nothing here is measured: every number is read from the rocprofv3 output of the named sessions."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sq_tag, rnd, prefix = sys.argv[1:5]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
G = os.path.join(ROOT, "gpurun_out")
shutil.copy(os.path.join(G, "%s_kernel_stats.csv" % tag), os.path.join(out, "%s_kernel_stats.csv" % prefix))
shutil.copy(os.path.join(G, "pmc_%s_summary.csv" % tag), os.path.join(out, "%s_pmc_hbm.csv" % prefix))
shutil.copy(os.path.join(G, "sq_%s_summary.csv" % sq_tag), os.path.join(out, "%s_sq_counters.csv" % prefix))
bench = json.loads(open(os.path.join(G, "%s_bench.json" % tag)).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, "%s_bench.json" % prefix), "w"), indent=1)
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def short(name):
    return name.replace("void ", "").split("(")[0].replace("grm::", "").split("<")[0].replace("_kernel", "")


traffic, sums = {}, 0.0
for r in csv.DictReader(open(os.path.join(out, "%s_pmc_hbm.csv" % prefix))):
    b = float(r["fetch_bytes_per_launch_x2corrected"]) + float(r["write_bytes_per_launch"])
    sums += b * int(r["launches"])
    k = short(r["kernel"])
    k = {"superkmer_l2_records": "superkmer_l2", "matrix_entry_rows": "matrix_fill"}.get(k, k)
    traffic[k] = traffic.get(k, 0) + int(b)
traffic["matrix_fill"] = traffic.get("matrix_fill", 0) + traffic.pop("matrix_transpose", 0)
cfg = bench["config"]
json.dump({"workload": {"genomes": cfg["genomes_total"], "genome_len": 5000000, "mode": "P", "k": 31},
           "commit": commit,
           "source": "profiles/%s/%s_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)" % (rnd, prefix),
           "bytes_per_launch": traffic, "sum_over_all_kernels_per_pass": int(sums)}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)

# what bounds a kernel, from the SQ counters: VALU issue time = wave instructions x 4 cycles / (1024 SIMDs x 2.4 GHz), against the
# kernel's average duration in the --stats table of the same build
dur = {}
for r in csv.DictReader(open(os.path.join(out, "%s_kernel_stats.csv" % prefix))):
    k = short(r["Name"])
    dur[k] = dur.get(k, 0.0) + float(r["AverageNs"]) * 1e-6
limits = {}
for r in csv.DictReader(open(os.path.join(out, "%s_sq_counters.csv" % prefix))):
    k = short(r["kernel"])
    if k not in dur or dur[k] < 0.3:
        continue
    valu_ms = float(r["SQ_INSTS_VALU"]) * 4 / (1024 * 2.4e9) * 1e3
    wait = float(r["SQ_WAIT_ANY"]) / max(1.0, float(r["SQ_WAVE_CYCLES"]))
    name = {"superkmer_l2_records": "superkmer_l2"}.get(k, k)
    limits[name] = ("VALU issue %.2f ms of its %.2f ms (%.2e wave instructions x 4 cycles on 1024 SIMDs at 2.4 GHz = %.0f %% of the kernel); waves wait %.0f %% of "
                    "their cycles; LDS bank-conflict cycles %.2e on %.2e LDS instructions"
                    % (valu_ms, dur[k], float(r["SQ_INSTS_VALU"]), 100 * valu_ms / dur[k], 100 * wait, float(r["SQ_LDS_BANK_CONFLICT"]), float(r["SQ_INSTS_LDS"])))
json.dump({"commit": commit, "source": "profiles/%s/%s_sq_counters.csv + %s_kernel_stats.csv" % (rnd, prefix, prefix), "limited_by": limits},
          open(os.path.join(ROOT, "profiles", "limits.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(limits, indent=1))
