#!/bin/bash
# probe builds of the whole library with csrc/sinkhorn.hip compiled under extra flags: tools/probe/libs_<name>.so
# usage: build_sk_variants.sh name "-DSKR_ABL=1" name2 "..." ...   (the other objects come from the last `make`)
set -e
cd "$(dirname "$0")/../../glue-factory_amd/csrc"
out=$(cd ../../tools/probe && pwd)
others=$(ls *.o | grep -v '^sinkhorn.o$')
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -fno-slp-vectorize $flags -c sinkhorn.hip -o /tmp/sk_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/sk_$name.o -o $out/libs_$name.so ) &
done
wait
