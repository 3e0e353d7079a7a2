#!/bin/bash
# probe builds of the three attention translation units: tools/probe/libv_<name>.so
#   usage: build_attn_variants.sh name "-DFLAG ..." [name "-DFLAG ..."] ...
# (attention_fwd3.hip / attention_bwd3.hip are always built with -fno-slp-vectorize, as in csrc/Makefile)
set -e
cd "$(dirname "$0")/../../glue-factory_amd/csrc"
out=$(cd ../../tools/probe && pwd)
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I."
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( $CC $flags -c attention.hip -o $out/v_${name}_a.o && $CC $flags -fno-slp-vectorize -c attention_fwd3.hip -o $out/v_${name}_f.o && \
    $CC $flags -fno-slp-vectorize -c attention_bwd3.hip -o $out/v_${name}_b.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/v_${name}_a.o $out/v_${name}_f.o $out/v_${name}_b.o -o $out/libv_$name.so && rm -f $out/v_${name}_?.o ) &
done
wait
