# e2e leg of bench.py with different numbers of upload threads / slab sizes: bash scripts/e2e_upload_sweep.sh "<threads> <slab_kb>" ...
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/e2es
for CFG in "$@"; do
  set -- $CFG
  cd $R && GRM_UPLOAD_THREADS=$1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 --opt upload_slab_kb=$2 > gpurun_out/e2es/b.json 2> gpurun_out/e2es/b.err
  python3 - "$CFG" <<'PY'
import json, sys
o = json.loads(open("gpurun_out/e2es/b.json").read().strip().splitlines()[-1])
e = o["e2e"]
print("threads slab_kb = %-12s e2e %.3f s  phases %s" % (sys.argv[1], e["seconds"], e["phases_s"]))
PY
done
