#!/bin/bash
# Ablation builds of the split GEMM (G16_ABL bit mask, csrc/gemm_split.hip) as separate libraries next to the product one:
#   tools/abl/<mask>/libdupl_hip.so ; run the stand-alone bench against one with LD_LIBRARY_PATH=tools/abl/<mask>
set -e
cd "$(dirname "$0")/.."
for m in "$@"; do
  mkdir -p tools/abl/$m
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DG16_ABL=$m -c dupl_amd/csrc/gemm_split.hip -o tools/abl/$m/gemm_split.o
  objs=$(ls dupl_amd/csrc/_obj/*.o | grep -v gemm_split.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/abl/$m/libdupl_hip.so tools/abl/$m/gemm_split.o $objs
done
