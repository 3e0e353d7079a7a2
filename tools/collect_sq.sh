#!/bin/bash
# Issue-side counters of the roofline kernels (attention fwd / dQ / dK-dV): where the wave cycles go.
#   gpurun -- 'bash tools/collect_sq.sh r02b'   -> gpurun_out/sq_<tag>/pass*/  (summarise with tools/probe/sq_summary.py)
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $REPO/bench.py --roofline-only"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d "$OUT/pass$i" -o pmc --output-format csv -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
