# headline dict_build + rank budget for option sets: bash scripts/rb_sweep.sh "<opts A>" "<opts B>" ...
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/rbs
for OPTS in "$@"; do
  ARGS=""
  for o in $OPTS; do ARGS="$ARGS --opt $o"; done
  cd $R && timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --cpu-genomes 0 --rank-budget 8 $ARGS > gpurun_out/rbs/b.json 2> gpurun_out/rbs/b.err
  python3 - "$OPTS" gpurun_out/rbs/b.json <<'PY'
import json, sys
try:
    o = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    rb = o["rank_budget"]
    k = o["kernels"]
    print("%-52s pass %6.2f dict %5.2f l2 %5.2f | rank %5.2f:" % (sys.argv[1] or "(default)", o["ms_per_step"], k["dict_build"]["avg_ms"], k["superkmer_l2"]["avg_ms"], rb["rank_ms"]),
          " ".join("%s %.2f" % (a.replace("superkmer_", "").replace("parse_", "p_").replace("dict_", "d_"), b) for a, b in rb["kernels_of_rank0_ms"].items() if b > 0.1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
