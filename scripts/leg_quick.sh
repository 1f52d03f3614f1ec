# one bench leg alone, kernel times on one line: bash scripts/leg_quick.sh <c5|c4|random|realistic> [bench args]
LEG=$1; shift
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --only $LEG --cpu-genomes 0 "$@" > gpurun_out/legq.json 2> gpurun_out/legq.err
python3 - $LEG <<'PY'
import json, sys
o = json.loads(open("gpurun_out/legq.json").read().strip().splitlines()[-1])
key = {"random": "random_acgt"}.get(sys.argv[1], sys.argv[1])
c = o[key]
print("%s %.2f ms  bit_exact %s " % (key, c["ms_per_step"], c.get("bit_exact_sample")), {k: round(v["avg_ms"], 2) for k, v in c["kernels"].items() if v["avg_ms"] > 0.2})
PY
