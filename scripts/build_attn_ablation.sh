#!/bin/bash
# Build an experiment copy of the library in which only attention.hip gets the extra -D flags: scripts/build_attn_ablation.sh <tag> -DOG_PIPE_ABL=1
# -> openglue_amd/lib/libog_<tag>.so (the other objects are the regular build's), selected at run time with OPENGLUE_AMD_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
python -m openglue_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c openglue_amd/csrc/attention.hip -o /tmp/attention_$tag.o 2> >(grep -v "not a recognized feature" >&2)
objs=$(ls openglue_amd/lib/*.o | grep -v "/attention.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openglue_amd/lib/libog_$tag.so $objs /tmp/attention_$tag.o
echo openglue_amd/lib/libog_$tag.so
