#!/bin/bash
# Build an experiment copy of the library with extra -D flags:  scripts/build_ablation.sh <tag> -DOG_GEMM_ABL=1 ...
# -> openglue_amd/lib/libog_<tag>.so, selected at run time with OPENGLUE_AMD_LIB=<path>.  Profiling only.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
tmp=$(mktemp -d)
for f in gemm_f32 gemm_f16x3 mlp_fused attention linear_attention sinkhorn sinkhorn_resident sinkhorn_train batchnorm_train matches features api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -Wno-unused-function "$@" -c openglue_amd/csrc/$f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openglue_amd/lib/libog_$tag.so $tmp/*.o
rm -rf $tmp
echo openglue_amd/lib/libog_$tag.so
