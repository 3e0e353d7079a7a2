"""Collapse rocprofv3 --pmc counter_collection CSVs into one row per (kernel, counter): mean per dispatch.
usage: pmc_summarize.py out.csv dir1 [dir2 ...]"""
import csv, glob, sys, collections
out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: [0.0, 0])
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(k in name for k in ("attn_", "linear_dw", "rows_", "dual_softmax", "ln_gelu", "gemm_nt")):
                continue
            a = agg[(name[:90], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
    for (k, c), (s, n) in sorted(agg.items()):
        w.writerow([k, c, f"{s / n:.1f}", n])
print(open(out).read())
