#!/bin/bash
# Experiment copies of the library that differ only in csrc/mlp_fused.hip:  scripts/build_mlp_ablation.sh <tag> -DOG_MLP_ABL=4 ...
# -> openglue_amd/lib/libog_<tag>.so (OPENGLUE_AMD_LIB=<path> selects it).  Needs the regular build's objects (python -m openglue_amd.build).
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -Wno-unused-function "$@" -c openglue_amd/csrc/mlp_fused.hip -o /tmp/mlp_fused_$tag.o
objs=$(ls openglue_amd/lib/*.o | grep -v mlp_fused)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openglue_amd/lib/libog_$tag.so $objs /tmp/mlp_fused_$tag.o
echo openglue_amd/lib/libog_$tag.so
