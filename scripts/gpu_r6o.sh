#!/bin/bash
# round 6, call o: the 16x16x32 pipelined attention kernel (OG_ATTN_P16=1) -- edge cases against float64, timing against the phase form
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06o; mkdir -p $OUT; rm -f $OUT/${TAG}_attention_p16_ab.log
for mode in 0 1 0 1; do
  OG_ATTN_P16=$mode timeout 600 python scripts/check_attention_pipe.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_attention_p16_ab.log
done
grep "worst\|us per launch\|Error\|error\|assert" $OUT/${TAG}_attention_p16_ab.log | cut -c1-230 | head -40
grep "p16" $OUT/${TAG}_attention_p16_ab.log | grep -v "us per" | head -32 | cut -c1-120
