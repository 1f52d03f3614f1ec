"""rank-budget part of a bench.py line: python scripts/show_rb.py <file>"""
import json
import sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rb = o["rank_budget"]
print(o["ms_per_step"], rb["stage_ms_slowest_rank"], rb["rank_ms"], rb["speedup_bound_vs_1gpu"])
print(rb["kernels_of_rank0_ms"])
