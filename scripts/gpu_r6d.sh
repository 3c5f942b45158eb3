#!/bin/bash
# round 6, call d: where the pipelined attention loop's time goes -- timing builds without the in-step DMA (1), without the barrier (2), without both (3)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06d; mkdir -p $OUT
run() { echo "== $1 pipe=$2"; OPENGLUE_AMD_LIB=$3 OG_ATTN_PIPE=$2 timeout 300 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids; }
{
for rep in 1 2; do
run regular 0 openglue_amd/lib/libopenglue_amd.so
run regular 1 openglue_amd/lib/libopenglue_amd.so
run no-dma 1 openglue_amd/lib/libog_pipe_abl1.so
run no-barrier 1 openglue_amd/lib/libog_pipe_abl2.so
run no-dma-no-barrier 1 openglue_amd/lib/libog_pipe_abl3.so
done
} > $OUT/${TAG}_attention_pipe_ablation.log 2>&1
cat $OUT/${TAG}_attention_pipe_ablation.log
