"""Do kernels of the replayed step OVERLAP in time?  Reads a rocprofv3 kernel-trace CSV and reports kernels that start before the
previous one (in start order) has ended, per queue / stream pair.  usage: overlap_check.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
keys = rows[0].keys()
sk, ek = [k for k in keys if k.lower().startswith("start")][0], [k for k in keys if k.lower().startswith("end")][0]
nk = [k for k in keys if "kernel_name" in k.lower() or k == "Kernel_Name"][0]
qk = [k for k in keys if "queue" in k.lower()]
rows.sort(key=lambda r: int(r[sk]))
ov = 0
pairs = collections.Counter()
queues = collections.Counter(r[qk[0]] for r in rows) if qk else {}
last_end, last = 0, None
sizes = []
for idx, r in enumerate(rows):
    s, e = int(r[sk]), int(r[ek])
    if last is not None and s < last_end:
        ov += 1
        pairs[(last[nk][:50], r[nk][:50])] += 1
        sizes.append((last_end - s, idx, last[nk][:40], r[nk][:40], int(last[ek]) - int(last[sk]), e - s))
    if e > last_end:
        last_end, last = e, r
print(f"{len(rows)} kernel dispatches, {ov} start before the running maximum end; queues: {dict(queues)}")
for (a, b), n in pairs.most_common(12):
    print(f"  {n:5d} x  [{a}]  still running when  [{b}]  starts")

sizes.sort(reverse=True)
print("largest overlaps (ns of overlap, dispatch index, producer, consumer, producer duration ns, consumer duration ns):")
for t in sizes[:15]:
    print("  ", t)
import statistics
print("median overlap ns:", statistics.median(x[0] for x in sizes) if sizes else None)
