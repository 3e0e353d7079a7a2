#!/bin/bash
# builds probe variants of the library: tools/probe/libv_<name>.so  (usage: build_variants.sh name "-DFLAG ..." ...)
set -e
cd /root/repo/glue-factory_amd/csrc
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. $flags -shared attention.hip -o /root/repo/tools/probe/libv_$name.so &
done
wait
