#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for v in 1 4; do for ab in 0 1; do echo "== variant $v ablate $ab"; OG_GEMM_VARIANT=$v OG_GEMM_ABLATE=$ab timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v "amdgpu.ids\|variant"; done; done | tee gpurun_out/gemm_ab.log
OG_GEMM_VARIANT=4 timeout 600 python -m pytest tests -m gpu -q -k "f16x3 or forward" -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/gemm_ab.log
