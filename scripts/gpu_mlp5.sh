#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "mlp_block or stage_taps or fixture" > $OUT/pytest_mlp.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_mlp.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_mlp.log | tail -5
bash scripts/gpu_mlp4.sh "$@"
OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/libog_trace.so timeout 300 python scripts/trace_mlp.py 65536 2>&1 | grep -v "^  median" | tail -12
