#!/bin/bash
# per-tile time of the 256-tile GEMM vs the number of busy CUs, default build and the compile-time ablations (scripts/build_ablation.sh gabl_N)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp OG_GEMM_TILE=256
{ timeout 200 python scripts/bench_gemm_scaling.py | sed 's/^/abl=0  /'
  for f in openglue_amd/lib/libog_gabl_*.so; do [ -e "$f" ] || continue; n=${f##*_}; n=${n%.so}; OPENGLUE_AMD_LIB=$PWD/$f timeout 200 python scripts/bench_gemm_scaling.py | sed "s/^/abl=$n  /"; done; } 2>&1 | grep "tiles=" | tee gpurun_out/gemm_scaling.log
