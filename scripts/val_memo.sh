mkdir -p gpurun_out/r02/val1
timeout -k 10 280 python3 scripts/diverse_check.py 200 2000000 > gpurun_out/r02/val1/diverse.log 2>&1; tail -3 gpurun_out/r02/val1/diverse.log
EXTRA="--mode R --genomes 100" bash scripts/sweep_opts.sh r02/val1R "" "rec_memo=0"
EXTRA="--genomes 128" bash scripts/sweep_opts.sh r02/val1P128 "" "rec_memo=0"
EXTRA="--genomes 300 --k 21" bash scripts/sweep_opts.sh r02/val1k21 "" "rec_memo=0"
