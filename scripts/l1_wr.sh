# level-1 time and HBM write bytes for a list of option sets: bash scripts/l1_wr.sh "<opts A>" "<opts B>" ...
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/l1wr
i=0
for OPTS in "$@"; do
  ARGS=""
  for o in $OPTS; do ARGS="$ARGS --opt $o"; done
  cd $R && timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --cpu-genomes 0 $ARGS > gpurun_out/l1wr/b_$i.json 2> gpurun_out/l1wr/b_$i.err
  python3 - "$OPTS" gpurun_out/l1wr/b_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-30s %7.2f ms |" % (sys.argv[1] or "(default)", d["ms_per_step"]), " ".join("%s %.2f" % (k.replace("superkmer_", "").replace("parse_", "p_"), v["avg_ms"]) for k, v in d["kernels"].items() if v["avg_ms"] > 0.2))
PY
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/l1wr/pmc_$i
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/l1wr/pmc_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --cpu-genomes 0 $ARGS > $R/gpurun_out/l1wr/pmc_$i.log 2>&1
  python3 - $R/gpurun_out/l1wr/pmc_$i <<'PY'
import csv, glob, sys, collections
w = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "superkmer" in k or "dict_build" in k:
            w[k] += 1024 * float(r["Counter_Value"])
print("    written GB:", {k.replace("grm::", "")[:28]: round(v / 1e9, 2) for k, v in w.items()})
PY
  i=$((i+1))
done
