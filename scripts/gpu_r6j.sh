#!/bin/bash
# round 6, call j: the eight-wave / 256-query form of the pipelined attention loop (OG_ATTN_PIPE=2) against the phase form and the four-wave pipelined form
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06j; mkdir -p $OUT; rm -f $OUT/${TAG}_attention_pipe8_ab.log
for mode in 0 2 1 0 2; do
  OG_ATTN_PIPE=$mode timeout 600 python scripts/check_attention_pipe.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_attention_pipe8_ab.log
done
grep "worst\|us per launch\|Error\|error\|assert" $OUT/${TAG}_attention_pipe8_ab.log | cut -c1-230 | head -60
