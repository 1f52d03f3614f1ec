# usage: bash scripts/sweep_opts.sh <tag> "<opt list A>" "<opt list B>" ...   (on the GPU box)
# each opt list is a space-separated set of name=value knobs for one bench run; prints the per-kernel times
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/$TAG
i=0
for OPTS in "$@"; do
  ARGS=""
  for o in $OPTS; do ARGS="$ARGS --opt $o"; done
  timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --cpu-genomes 0 $EXTRA $ARGS > $R/gpurun_out/$TAG/sweep_$i.json 2> $R/gpurun_out/$TAG/sweep_$i.err || echo "run $i failed"
  python3 - "$OPTS" $R/gpurun_out/$TAG/sweep_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-40s %7.2f ms |" % (sys.argv[1] or "(default)", d["ms_per_step"]), " ".join("%s %.2f" % (k.replace("kmer_scatter_", "").replace("parse_", "p_"), v["avg_ms"]) for k, v in d["kernels"].items() if v["avg_ms"] > 0.2))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  i=$((i+1))
done
