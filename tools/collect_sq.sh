#!/bin/bash
# Issue-side counters of the attention kernels (fwd / dQ / dK-dV) of one probe build: where the wave cycles go.
#   gpurun -- 'bash tools/collect_sq.sh <tag> tools/probe/libv_x.so [fwd|bwd|both]'
#   -> gpurun_out/sq_<tag>/pass*/  (summarise with tools/probe/sq_summary.py gpurun_out/sq_<tag>)
set -u
TAG=${1:-r03}
LIB=$(readlink -f ${2:-glue-factory_amd/libgf_amd.so})
WHAT=${3:-both}
REPO=$(pwd)
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $REPO/tools/probe/attn_once.py $LIB $WHAT"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_VALU_MFMA SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d "$OUT/pass$i" -o pmc --output-format csv -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
python $REPO/tools/probe/sq_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
