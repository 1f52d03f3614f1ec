"""compact view of a bench.py JSON line: python scripts/show_bench.py <file>"""
import json
import sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
KEEP = ("record_memo", "bit_exact", "floors", "kover_gzip5", "cpu_baseline", "phases_s", "cpu")


def short(d):
    return {k: (v if not isinstance(v, dict) or k in KEEP else "...") for k, v in d.items() if k != "kernels"}


print(json.dumps(short(o))[:2000])
print("headline kernels", {k: v["avg_ms"] for k, v in o["kernels"].items()})
print("roofline", json.dumps(o.get("roofline"))[:900])
for leg in ("realistic", "c5", "c4", "random_acgt", "e2e", "weak", "collective"):
    if leg in o and o[leg]:
        print(leg, json.dumps(short(o[leg]))[:2200])
        if isinstance(o[leg], dict) and "kernels" in o[leg]:
            print("    kernels", {k: v["avg_ms"] for k, v in o[leg]["kernels"].items()})
