#!/bin/bash
# Builds librsis_hip_<tag>.so variants that differ only in conv_wino.hip's -D switches (timing experiments; see README.md).
#   tools/exp/wino/build_variants.sh tag1="-DWINO_ABL=1" tag2="-DWINO_NOUT=3" ...
set -e
cd "$(dirname "$0")/../../../rsis_amd/csrc"
make -j8 >/dev/null
for kv in "$@"; do
  tag="${kv%%=*}"; flags="${kv#*=}"
  mkdir -p build_var
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function $flags -c conv_wino.hip -o build_var/conv_wino_$tag.o
  objs=$(ls build/*.o | grep -v conv_wino.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librsis_hip_$tag.so $objs build_var/conv_wino_$tag.o -ldl
  echo "built ../lib/librsis_hip_$tag.so ($flags)"
done
