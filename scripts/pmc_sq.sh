# usage: bash scripts/pmc_sq.sh <tag> [bench args...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/sq1_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 "$@" > $R/gpurun_out/sq1_$TAG.log 2>&1 || echo "sq1 failed"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/sq2_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 "$@" > $R/gpurun_out/sq2_$TAG.log 2>&1 || echo "sq2 failed"
cd $R && python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/sq?_$TAG/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ";")
        if k.startswith("grm::"):
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for d in rows.values() for c in d})
with open("gpurun_out/sq_${TAG}_summary.csv", "w") as out:
    out.write("kernel," + ",".join(names) + "\n")
    for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        out.write(k + "," + ",".join("%.4g" % d.get(c, 0) for c in names) + "\n")
print(open("gpurun_out/sq_${TAG}_summary.csv").read()[:6000])
PY
