#!/bin/bash
# Build an experiment copy of the library with extra -D flags:  scripts/build_ablation.sh <tag> -DOG_GEMM_ABL=1 ...
# -> openglue_amd/lib/libog_<tag>.so, selected at run time with OPENGLUE_AMD_LIB=<path>.  Profiling only.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
tmp=$(mktemp -d)
for f in $(python -c 'from openglue_amd.build import SOURCES; print(" ".join(s[:-4] for s in SOURCES))'); do
  per=$(python -c "from openglue_amd.build import PER_FILE_FLAGS; print(' '.join(PER_FILE_FLAGS.get('$f.hip', [])))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -Wno-unused-function $per "$@" -c openglue_amd/csrc/$f.hip -o $tmp/$f.o 2> >(grep -v "not a recognized feature" >&2) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openglue_amd/lib/libog_$tag.so $tmp/*.o
rm -rf $tmp
echo openglue_amd/lib/libog_$tag.so
