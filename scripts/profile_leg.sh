# kernel-trace statistics of ONE bench leg: bash scripts/profile_leg.sh <tag> <leg>   (leg: random / c4 / c5 / realistic)
TAG=$1; LEG=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${LEG}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --only $LEG --cpu-genomes 0 > $R/gpurun_out/${TAG}_${LEG}_stats.log 2>&1; echo "stats exit $?"
cp $R/gpurun_out/${TAG}_${LEG}_stats/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_${LEG}_kernel_stats.csv
head -12 $R/gpurun_out/${TAG}_${LEG}_kernel_stats.csv | cut -c1-160
