#!/bin/bash
# round 6, call l: static wave priority for every other attention workgroup (timing, results unchanged)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
{
for rep in 1 2 3; do
for tag in regular prio1 prio6 prio7; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ $tag != regular ] && lib=openglue_amd/lib/libog_$tag.so
  echo -n "$tag: "; OPENGLUE_AMD_LIB=$lib timeout 300 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids | tail -1
done; done
} | tee gpurun_out/r06l_attention_static_prio.log
