#!/bin/bash
# Per-kernel time of a whole train step, launched kernel by kernel (rocprofv3 --kernel-trace --stats):
#     gpurun -- 'bash tools/collect_step_stats.sh <tag> [lightglue|superglue|gluestick]'
# -> profiles/<tag>_<model>_step_kernel_stats.csv (lightglue: the headline scope P incl. the extractor; the other two:
#    their matcher step, BASELINE configs[3] / configs[4]).  15 timed steps + 3 warm-up steps in the trace.
set -u
TAG=${1:-r03}
MODEL=${2:-lightglue}
REPO=$(pwd)
OUT=$REPO/gpurun_out/stepstats_${TAG}_$MODEL
mkdir -p "$OUT" "$REPO/profiles"
export TMPDIR=/tmp
EXTRA=""
[ "$MODEL" != lightglue ] && EXTRA="--model $MODEL"
CMD="python $REPO/bench.py --steps 15 --warmup 3 --no-graph --no-roofline --no-other-configs --no-cpu-baseline $EXTRA"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $CMD > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
cd "$REPO"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "profiles/${TAG}_${MODEL}_step_kernel_stats.csv"; mkdir -p gpurun_out/profiles; cp "$f" "gpurun_out/profiles/${TAG}_${MODEL}_step_kernel_stats.csv"; head -12 "$f" | cut -c1-150; else echo "no kernel_stats.csv"; tail -5 "$OUT/stats.log"; fi
tail -1 "$OUT/stats.log" | cut -c1-400
