#!/bin/bash
# fused message-MLP kernel: A/B of experiment libraries (microbench with its correctness line)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
: > $OUT/mlp_ab.log
for tag in "" "$@"; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ -n "$tag" ] && lib=openglue_amd/lib/libog_$tag.so
  echo "--- ${tag:-as built}" >> $OUT/mlp_ab.log
  OPENGLUE_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_mlp_fused.py 2>&1 | grep "M=\|err\|Error" >> $OUT/mlp_ab.log
done
cat $OUT/mlp_ab.log
