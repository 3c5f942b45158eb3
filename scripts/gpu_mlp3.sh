#!/bin/bash
# fused message-MLP kernel: stage tests, microbench (+ two ablation builds), per-stage cycle trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "mlp_block" > $OUT/pytest_mlp.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_mlp.log
grep -E "passed|failed|error|FAILED|ERROR|rc=|assert" $OUT/pytest_mlp.log | tail -12
: > $OUT/mlp_ablations.log
for tag in "" nodma nomfma nostore; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ -n "$tag" ] && lib=openglue_amd/lib/libog_$tag.so
  echo "--- ${tag:-as built}" >> $OUT/mlp_ablations.log
  OPENGLUE_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_mlp_fused.py 2>&1 | grep "M=\|err" >> $OUT/mlp_ablations.log
done
cat $OUT/mlp_ablations.log
OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/libog_trace.so timeout 300 python scripts/trace_mlp.py 65536 > $OUT/mlp_trace.log 2>&1
cat $OUT/mlp_trace.log
