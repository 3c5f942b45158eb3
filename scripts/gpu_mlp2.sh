#!/bin/bash
# fused message-MLP kernel: compile-time ablations (microbench of each experiment library) + the per-stage cycle trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
: > $OUT/mlp_ablations.log
for tag in "" nostore noconv noxdma nodma nofrag nodma_nofrag nomfma mfma_only; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ -n "$tag" ] && lib=openglue_amd/lib/libog_$tag.so
  echo "--- ${tag:-as built}" >> $OUT/mlp_ablations.log
  OPENGLUE_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_mlp_fused.py 2>&1 | grep "M=" >> $OUT/mlp_ablations.log
done
cat $OUT/mlp_ablations.log
OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/libog_trace.so timeout 300 python scripts/trace_mlp.py > $OUT/mlp_trace.log 2>&1
cat $OUT/mlp_trace.log
