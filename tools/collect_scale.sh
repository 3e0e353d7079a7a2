#!/bin/bash
# Evidence for the multi-GPU claims the day an 8-GPU node is available (the 1-GPU test boxes cannot run it):
#   bash tools/collect_scale.sh <tag> [N=8]
# 1. bench.py at 1/2/4/N GPUs -> gpurun_out/scale_<tag>/n<k>.json (the driver computes efficiency from `value`);
# 2. rocprofv3 --kernel-trace (+ RCCL API trace) of the N-GPU run, one output directory per rank: the per-rank kernel
#    timelines show whether the gradient-bucket all-reduces (ncclDevKernel_AllReduce_Sum_f32_*) run UNDER the transformer
#    backward (attn_dq3 / attn_bwd_dkv / gemm_st) -- the overlap DESIGN.md section 6 claims -- and how long the last
#    bucket stays on the critical path in front of the fused Adam;
# 3. the same with the step un-captured (--no-graph) for the launch-gap comparison.
# (Counters are not collected here: --pmc is never combined with runtime / API tracing.)
set -u
TAG=${1:-r03}
N=${2:-8}
REPO=$(pwd)
OUT=$REPO/gpurun_out/scale_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 1 2 4 $N; do
  [ "$k" -gt "$N" ] && continue
  python bench.py --gpus $k --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs > "$OUT/n$k.json" 2> "$OUT/n$k.err" || echo "n=$k failed"
done
# N = 1 through the launcher must be the single-process number (same code path, no collective): within 2 %
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs > "$OUT/single.json" 2> "$OUT/single.err" || echo "single-process run failed"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
rd = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
try:
    a, b = rd(f"{out}/n1.json")["value"], rd(f"{out}/single.json")["value"]
    print(f"N=1 via the launcher {a} vs single process {b}: {abs(a - b) / b * 100:.2f} %")
    assert abs(a - b) <= 0.02 * b, "N=1 differs from the single-process number by more than 2 %"
    for k in (2, 4, 8):
        try:
            d = rd(f"{out}/n{k}.json")
        except Exception:
            continue
        dp = d.get("data_parallel", {})
        assert sorted(dp.get("ranks_seen", [])) == list(range(k)), f"N={k}: ranks seen {dp.get('ranks_seen')}"
        print(f"N={k}: value {d['value']}  ranks {dp.get('ranks_seen')}  buckets {dp.get('buckets')}  all-reduce busbw {dp.get('allreduce_busbw_GBps')} GB/s")
except Exception as e:
    print("scale check:", e)
PY
cd /tmp
for mode in graph nograph; do
  extra=""; [ "$mode" = nograph ] && extra="--no-graph"
  timeout 900 rocprofv3 --kernel-trace --rccl-trace --stats -d "$OUT/trace_$mode" -o trace --output-format csv -- \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
    "$REPO/bench.py" --gpus $N --steps 6 --warmup 3 --matcher-only --no-cpu-baseline --no-roofline --no-other-configs $extra \
    > "$OUT/trace_$mode.log" 2>&1 || echo "trace ($mode) failed"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for mode in ("graph", "nograph"):
    for f in sorted(glob.glob(f"{out}/trace_{mode}/**/*kernel_trace.csv", recursive=True))[:1]:
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        ar = [r for r in rows if "AllReduce" in r["Kernel_Name"] or "allreduce" in r["Kernel_Name"].lower()]
        bw = [r for r in rows if "attn_" in r["Kernel_Name"] or "gemm_st" in r["Kernel_Name"]]
        if not ar or not bw:
            print(mode, "no all-reduce / backward kernels found in", f)
            continue
        lo, hi = int(bw[0]["Start_Timestamp"]), int(bw[-1]["End_Timestamp"])
        inside = sum(min(int(r["End_Timestamp"]), hi) - max(int(r["Start_Timestamp"]), lo) for r in ar
                     if int(r["End_Timestamp"]) > lo and int(r["Start_Timestamp"]) < hi)
        total = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ar)
        print(f"{mode}: {len(ar)} all-reduce kernels, {total / 1e6:.3f} ms, {100.0 * inside / max(total, 1):.1f} % of it inside the "
              f"span of the attention / GEMM kernels")
PY
