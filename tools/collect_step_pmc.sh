#!/bin/bash
# Per-kernel counters over the whole headline step (extractor + ground truth + matcher step, launched kernel by kernel):
#     gpurun -- 'bash tools/collect_step_pmc.sh r02d'    then locally: python tools/step_pmc_summary.py r02d gpurun_out/steppmc_r02d
# Separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one), never combined with tracing.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/steppmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
MODEL=${2:-lightglue}
EXTRA=""; [ "$MODEL" != lightglue ] && EXTRA="--model $MODEL"
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-other-configs --no-cpu-baseline $EXTRA"
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d "$OUT/pass$i" -o pmc --output-format csv -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
