mkdir -p gpurun_out
for o in "groups_per_thread=4"; do
  timeout -k 10 300 python bench.py --genomes 400 --steps 3 --warmup 1 --cpu-sample 0 --opt $o > gpurun_out/sweep_$o.log 2>&1
  tail -1 gpurun_out/sweep_$o.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$o', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if v['avg_ms']>1})
"
done
