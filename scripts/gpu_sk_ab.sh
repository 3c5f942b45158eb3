#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 300 python -m pytest tests -m gpu -q -k "sinkhorn" -p no:cacheprovider | tail -3; for sh in 32,1024,1024,100 32,2048,2048,50 8,4096,4096,50; do OG_SK_SHAPE=$sh timeout 300 python scripts/bench_sinkhorn.py; done; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/sk_ab.log
