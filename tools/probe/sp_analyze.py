import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sample_desc" in r["Kernel_Name"] or "grid_sampler" in r["Kernel_Name"]]
print("calls:", len(idx))
start = idx[-4] + 1
seg = rows[start:idx[-1] + 1]
ncall = 3
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"][:100]][0] += d; agg[r["Kernel_Name"][:100]][1] += 1
tot = sum(v[0] for v in agg.values())
print("steady kernel time per call: %.2f ms" % (tot / ncall / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
    print("%8.3f ms %4d x  %s" % (v[0] / ncall / 1e6, v[1] // ncall, k))
