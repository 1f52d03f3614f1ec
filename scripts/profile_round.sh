# usage: bash scripts/profile_round.sh <tag>   (on the GPU box; writes gpurun_out/<tag>_*)
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 500 python3 bench.py --steps 5 --warmup 2 --no-e2e --rank-budget 0 --cpu-genomes 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-e2e --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 > $R/gpurun_out/${TAG}_stats.log 2>&1; echo "stats exit $?"
cd $R && bash scripts/pmc.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1; echo "pmc exit $?"
cp gpurun_out/${TAG}_stats/*/*_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-600
# the files -> .kover leg under the profiler: the encoder kernels of the HDF5 writer (grm_deflate.hip)
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_e2e_stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-random --no-realistic --no-c4 --no-c5 --rank-budget 0 --cpu-genomes 0 > $R/gpurun_out/${TAG}_e2e_stats.log 2>&1; echo "e2e stats exit $?"
cp $R/gpurun_out/${TAG}_e2e_stats/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_e2e_kernel_stats.csv
