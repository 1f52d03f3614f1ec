import collections, csv, glob, sys
tag = sys.argv[1]
rows = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/*/*_counter_collection.csv" % (tag, c)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ";")        # template arguments: no commas in a CSV cell
            if not k.startswith("grm::"):
                continue
            d = rows.setdefault(k, collections.defaultdict(float))
            d[c] += float(r["Counter_Value"])
            d[c + "_n"] += 1
out = ["kernel,launches,fetch_bytes_per_launch_x2corrected,write_bytes_per_launch"]
for k, d in sorted(rows.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    n = max(d["FETCH_SIZE_n"], d["WRITE_SIZE_n"], 1)
    out.append("%s,%d,%.0f,%.0f" % (k, n, 2 * 1024 * d["FETCH_SIZE"] / max(d["FETCH_SIZE_n"], 1), 1024 * d["WRITE_SIZE"] / max(d["WRITE_SIZE_n"], 1)))
open("gpurun_out/pmc_%s_summary.csv" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
