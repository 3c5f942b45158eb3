#!/bin/bash
# round 6, call q: the fused message MLP with its six LDS-DMA pieces per stage spread over the free slots (one per slot) instead of 4 + 2 back to back
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
{
for rep in 1 2 3; do
for tag in regular mlp_spread; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ $tag != regular ] && lib=openglue_amd/lib/libog_$tag.so
  echo "== $tag"; OPENGLUE_AMD_LIB=$lib timeout 300 python scripts/bench_mlp_fused.py 2>&1 | grep -v amdgpu.ids | grep "M="
done; done
} | tee gpurun_out/r06q_mlp_dma_spread.log
