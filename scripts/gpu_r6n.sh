#!/bin/bash
# round 6, call n: the real attention kernel with every 32x32x16 MFMA issued as two 16x16x32 (timing build, results wrong), phase and pipelined forms
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
{
for rep in 1 2 3; do
for tag in regular abl16; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ $tag != regular ] && lib=openglue_amd/lib/libog_$tag.so
  for pipe in 0 1; do echo -n "$tag pipe=$pipe: "; OG_ATTN_PIPE=$pipe OPENGLUE_AMD_LIB=$lib timeout 300 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done; done
} | tee gpurun_out/r06n_attention_abl16.log
