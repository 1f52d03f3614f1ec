import collections, csv, glob, sys
# per kernel: FETCH_SIZE / WRITE_SIZE per launch, averaged over the launches of the leg's own batch -- a leg run with `--only` starts with
# a token 16-genome headline pass whose launches of the shared kernels (parse, level 1) are a hundredth of the leg's: launches below
# half of the kernel's largest are left out of the average
tag = sys.argv[1]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/*/*_counter_collection.csv" % (tag, c)):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ";")        # template arguments: no commas in a CSV cell
            if not k.startswith("grm::"):
                continue
            d = r.get("Dispatch_Id") or r.get("Dispatch_ID") or str(len(per_dispatch))
            per_dispatch[d] += float(r["Counter_Value"])          # (one row per counter instance)
            names[d] = k
        for d, v in per_dispatch.items():
            vals.setdefault(names[d], {}).setdefault(c, []).append(v)
out = ["kernel,launches,fetch_bytes_per_launch_x2corrected,write_bytes_per_launch"]
rows = []
for k, d in vals.items():
    def avg_large(xs):
        if not xs:
            return 0.0, 0
        big = [x for x in xs if x >= 0.5 * max(xs)]
        return sum(big) / len(big), len(big)
    f, nf = avg_large(d.get("FETCH_SIZE", []))
    w, nw = avg_large(d.get("WRITE_SIZE", []))
    rows.append((k, max(nf, nw, 1), 2 * 1024 * f, 1024 * w))
for k, n, f, w in sorted(rows, key=lambda r: -(r[2] + r[3])):
    out.append("%s,%d,%.0f,%.0f" % (k, n, f, w))
open("gpurun_out/pmc_%s_summary.csv" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
