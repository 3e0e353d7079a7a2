#!/bin/bash
# builds probe variants of ONE source of the library: tools/probe/libv_<name>.so
#   usage: build_variants.sh <source.hip> name "-DFLAG ..." [name "-DFLAG ..."] ...
# (A/B runs must share a process: box-to-box clocks differ by +-5 %, see README.md)
set -e
cd "$(dirname "$0")/../../glue-factory_amd/csrc"
src=$1; shift
out=$(cd ../../tools/probe && pwd)
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. $flags -shared $src -o $out/libv_$name.so &
done
wait
