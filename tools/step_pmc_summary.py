"""profiles/<tag>_step_pmc.csv from tools/collect_step_pmc.sh: per kernel (mean per dispatch) MFMA-pipe busy fraction,
LDS bank-conflict share, VALU instructions per MFMA, VALU-active share of the wave cycles, HBM bytes
(FETCH_SIZE x 2 + WRITE_SIZE KiB, the gfx950 correction of MI355X_MICROARCH.md).  usage: step_pmc_summary.py <tag> <dir>"""
import collections, csv, glob, os, sys
tag, root = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:90]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = []
for k, c in agg.items():
    m = {n: v[0] / max(v[1], 1) for n, v in c.items()}
    n = max(v[1] for v in c.values())
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append({
        "kernel": k, "dispatches": n, "gui_cycles": round(gui),
        "mfma_busy_frac": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui / 8), 4) if gui else "",
        "lds_conflict_share": round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4) if m.get("SQ_LDS_IDX_ACTIVE") else "",
        "valu_per_mfma": round((m.get("SQ_INSTS_VALU", 0.0) - m.get("SQ_INSTS_MFMA", 0.0)) / m["SQ_INSTS_MFMA"], 2) if m.get("SQ_INSTS_MFMA") else "",
        "valu_active_share_of_wave_cycles": round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") else "",
        "hbm_mb": round((m.get("FETCH_SIZE", 0.0) * 2 + m.get("WRITE_SIZE", 0.0)) * 1024 / 1e6, 1),
        "total_gui": gui * n})
rows.sort(key=lambda r: -r["total_gui"])
out = os.path.join(ROOT, "profiles", f"{tag}_step_pmc.csv")
with open(out, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=[k for k in rows[0] if k != "total_gui"])
    w.writeheader()
    for r in rows[:60]:
        r = dict(r); r.pop("total_gui"); w.writerow(r)
for r in rows[:16]:
    print(r["kernel"][:60].ljust(60), r["dispatches"], r["mfma_busy_frac"], r["lds_conflict_share"], r["valu_per_mfma"], r["valu_active_share_of_wave_cycles"], r["hbm_mb"])
