#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ timeout 120 python scripts/bench_gemm.py | sed "s/^/abl=0  /"
  for f in openglue_amd/lib/libog_gabl_*.so; do [ -e "$f" ] || continue; n=${f##*_}; n=${n%.so}; OPENGLUE_AMD_LIB=$PWD/$f timeout 120 python scripts/bench_gemm.py | sed "s/^/abl=$n  /"; done; } 2>&1 | grep -v "amdgpu.ids" | grep "fc0\|qkv " | tee gpurun_out/gemm_abl2.log
