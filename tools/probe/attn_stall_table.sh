#!/bin/bash
# Stall budget of the self-attention backward (round-5 review item 4): probe builds of the attention translation units with
# ONE ingredient of the dK/dV or the dQ loop removed (GF_DKV_ABL / GF_DQ3_ABL bits, csrc/attention.hip, attention_bwd3.hip),
# timed in one process against the shipped form (tools/probe/time_attn.py, B2 = 64 images x 4 heads x 2048^2, scale = ln 2).
#   build container:  bash tools/probe/attn_stall_table.sh build
#   GPU box:          bash tools/probe/attn_stall_table.sh run > gpurun_out/attn_stall.txt
set -e
cd "$(dirname "$0")"
V="base:  dkv1:-DGF_DKV_ABL=1 dkv2:-DGF_DKV_ABL=2 dkv4:-DGF_DKV_ABL=4 dkv8:-DGF_DKV_ABL=8 dkv16:-DGF_DKV_ABL=16 dkv24:-DGF_DKV_ABL=24 dkv29:-DGF_DKV_ABL=29 dkv31:-DGF_DKV_ABL=31 dq1:-DGF_DQ3_ABL=1 dq2:-DGF_DQ3_ABL=2 dq4:-DGF_DQ3_ABL=4 dq8:-DGF_DQ3_ABL=8 dq16:-DGF_DQ3_ABL=16 dq24:-DGF_DQ3_ABL=24 dq29:-DGF_DQ3_ABL=29 dq31:-DGF_DQ3_ABL=31"
if [ "$1" = build ]; then
  args=()
  for v in $V; do args+=("st_${v%%:*}" "${v#*:}"); done
  # four at a time (8 build-container CPUs, 3 translation units each)
  for ((i = 0; i < ${#args[@]}; i += 8)); do bash build_attn_variants.sh "${args[@]:i:8}"; done
  ls -la libv_st_*.so | wc -l
else
  # (one process per group: a probe build that faults must not cost the others)
  python time_attn.py ./libv_st_base.so@ln2 2>&1 | grep -v amdgpu.ids
  for g in "dkv1 dkv2 dkv4" "dkv8 dkv16 dkv24" "dkv29 dkv31" "dq1 dq2 dq4" "dq8 dq16 dq24" "dq29 dq31"; do
    libs=(./libv_st_base.so@ln2)
    for v in $g; do libs+=("./libv_st_$v.so@ln2"); done
    python time_attn.py "${libs[@]}" 2>&1 | grep -v "amdgpu.ids\|hipBLASLt" || true
  done
fi
