#!/bin/bash
# GPU box: every GPU-side state of the "does the HIP-trained SuperGlue / GlueStick equal the reference-trained one" experiment
# (round-5 review item 1).  States go to gpurun_out/learn/ (<= 64 MiB travel back); evaluated offline in the build container by
# tools/probe/learn_anchor_report.py with the reference module.
set -x
cd "$(dirname "$0")/../.."
O=gpurun_out/learn
mkdir -p $O
export TMPDIR=/tmp
python tools/probe/learn_save_state.py superglue $O/sg_hip_fp32.pt fp32 > $O/sg_hip_fp32.log 2>&1
python tools/probe/learn_save_state.py superglue $O/sg_hip_bf16.pt bf16 > $O/sg_hip_bf16.log 2>&1
python tools/probe/learn_third_arithmetic.py superglue cuda float32 $O/sg_rocm_fp32.pt > $O/sg_rocm_fp32.log 2>&1
python tools/probe/learn_third_arithmetic.py superglue cuda float64 $O/sg_rocm_fp64.pt > $O/sg_rocm_fp64.log 2>&1
python tools/probe/learn_save_state.py gluestick $O/gs_hip_fp32.pt fp32 > $O/gs_hip_fp32.log 2>&1
python tools/probe/learn_save_state.py gluestick /tmp/gs_hip_bf16.pt bf16 > $O/gs_hip_bf16.log 2>&1
tail -n 4 $O/*.log
