# the headline leg alone, kernel times on one line: bash scripts/head_quick.sh [bench args]
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 "$@" > gpurun_out/headq.json 2> gpurun_out/headq.err
python3 - <<'PY'
import json
o = json.loads(open("gpurun_out/headq.json").read().strip().splitlines()[-1])
print("headline %.3f ms  cols %d " % (o["ms_per_step"], o["config"]["columns"]), {k: round(v["avg_ms"], 3) for k, v in o["kernels"].items() if v["avg_ms"] > 0.2})
PY
