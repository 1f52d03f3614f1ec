# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of ONE bench leg:
#   bash scripts/pmc_hbm_leg.sh <tag> <leg> [bench args...]   (leg: random / c4 / c5 / realistic)
# -> gpurun_out/pmc_<tag>_summary.csv   (FETCH_SIZE x2 per MI355X_MICROARCH.md, as scripts/pmc_summary.py does)
TAG=$1; LEG=$2; shift; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 1 --warmup 0 --only $LEG --cpu-genomes 0 "$@" > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1 || echo "pmc $C of $LEG failed"
done
cd $R && python3 scripts/pmc_summary.py $TAG
