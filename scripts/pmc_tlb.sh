# address-translation and read-latency counters of the headline pass: bash scripts/pmc_tlb.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $R/gpurun_out/tlb_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 "$@" > $R/gpurun_out/tlb_$TAG.log 2>&1 || echo "tlb pass failed"
cd $R && python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/tlb_$TAG/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("grm::"):
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open("gpurun_out/tlb_${TAG}_summary.csv", "w") as out:
    out.write("kernel,utcl1_miss,utcl1_hit,miss_rate,read_req,avg_read_latency_cycles\n")
    for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("TCP_TCC_READ_REQ_sum", 0)):
        m, h = d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0), d.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0)
        q, l = d.get("TCP_TCC_READ_REQ_sum", 0), d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0)
        out.write("%s,%.4g,%.4g,%.4f,%.4g,%.1f\n" % (k, m, h, m / max(1.0, m + h), q, l / max(1.0, q)))
print(open("gpurun_out/tlb_${TAG}_summary.csv").read()[:3000])
PY
