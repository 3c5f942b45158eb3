#!/bin/bash
# round 5, third GPU call: the GPU suite after the test fix, and what bounds proj_stream_kernel<256>: micro-benchmark of the launch forms with
# the ablation builds (no stores / no MFMA / no LDS-DMA after the prologue).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05c}"; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
tail -25 $OUT/${TAG}_pytest_gpu.log
{
for lib in "" abl1 abl4 abl8 abl5; do
  echo "=== lib ${lib:-regular} (OG_PROJ_STREAM=1)"
  if [ -n "$lib" ]; then export OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/libog_$lib.so; else unset OPENGLUE_AMD_LIB; fi
  OG_PROJ_STREAM=1 timeout 300 python scripts/bench_proj.py 2>&1 | grep -v amdgpu.ids
done
unset OPENGLUE_AMD_LIB
echo "=== regular, OG_PROJ_STREAM=0 (tile GEMMs)"
OG_PROJ_STREAM=0 timeout 300 python scripts/bench_proj.py 2>&1 | grep -v amdgpu.ids
} > $OUT/${TAG}_proj_micro.log 2>&1
cat $OUT/${TAG}_proj_micro.log
