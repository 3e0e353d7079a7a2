"""Mean per dispatch of the tools/collect_sq.sh counters for the attention kernels."""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "fwd" if "attn_fwd" in n else "dq" if ("bwd_dq" in n or "attn_dq3" in n) else "dkv" if "bwd_dkv" in n else None
        if k:
            a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
ks = sorted({k for k, _ in agg}); cs = sorted({c for _, c in agg})
print("counter".ljust(30), *[k.rjust(14) for k in ks])
for c in cs:
    print(c.ljust(30), *[f"{agg[(k, c)][0] / max(agg[(k, c)][1], 1):14.0f}" for k in ks])
