#!/bin/bash
# compile-time ablations of the attention kernel (profiling only; results are wrong by construction)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 120 python scripts/bench_attention.py | sed 's/^/abl=0  /'
  for f in openglue_amd/lib/libog_abl_*.so; do n=${f##*_}; n=${n%.so}; OPENGLUE_AMD_LIB=$PWD/$f timeout 120 python scripts/bench_attention.py | sed "s/^/abl=$n  /"; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_ablate.log
