#!/bin/bash
# probe builds of csrc/attention_xbwd.hip: tools/probe/libx_<name>.so    usage: build_xbwd_variants.sh name "-DFLAG ..." ...
set -e
cd "$(dirname "$0")/../../glue-factory_amd/csrc"
out=$(cd ../../tools/probe && pwd)
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -fno-slp-vectorize $flags -shared attention_xbwd.hip -o $out/libx_$name.so ) &
done
wait
