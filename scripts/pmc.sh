# usage: bash scripts/pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_{fetch,write}/
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 "$@" > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1 || echo "pmc $C failed"
done
cd $R && python3 scripts/pmc_summary.py $TAG
