#!/bin/bash
# Collect the HBM-traffic / MFMA-busy counters behind bench.py's roofline objects, on the GPU box:
#     gpurun -- 'bash tools/collect_pmc.sh r02'
# Separate rocprofv3 --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: they cannot share a pass;
# --pmc is never combined with --sys-trace / runtime tracing) over `python bench.py --roofline-only`, plus one
# --kernel-trace --stats pass for the average kernel durations.  Summaries: profiles/<tag>_roofline_pmc.csv,
# profiles/<tag>_roofline_kernel_stats.csv and profiles/roofline_traffic.json (read back by bench.py).
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT" "$REPO/profiles"
export TMPDIR=/tmp
CMD="python $REPO/bench.py --roofline-only"
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d "$OUT/pmc$i" -o pmc --output-format csv -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $CMD > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
cd "$REPO"
python tools/pmc_to_traffic.py "$TAG" "$OUT"
# only gpurun_out/ travels back from the GPU box: hand the summaries over through it (copy them into profiles/ locally)
mkdir -p "$REPO/gpurun_out/profiles"
cp "$REPO/profiles/${TAG}_roofline_pmc.csv" "$REPO/profiles/${TAG}_roofline_kernel_stats.csv" "$REPO/profiles/roofline_traffic.json" \
   "$REPO/profiles/${TAG}_roofline_traffic.json" "$REPO/gpurun_out/profiles/" 2>/dev/null
