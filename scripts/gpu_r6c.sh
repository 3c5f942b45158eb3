#!/bin/bash
# round 6, call c: the pipelined attention tile loop -- edge cases against float64, then A/B timing (one process per mode)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06c; mkdir -p $OUT
for mode in 0 1 0 1; do
  OG_ATTN_PIPE=$mode timeout 600 python scripts/check_attention_pipe.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_attention_pipe_ab.log
done
grep "worst\|us per launch\|Error\|error\|assert" $OUT/${TAG}_attention_pipe_ab.log | head -60
