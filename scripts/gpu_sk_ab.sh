#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ for rg in 1 2; do for rw in 4 8 16; do OG_SINKHORN_RG=$rg OG_SINKHORN_RW=$rw timeout 300 python scripts/bench_sinkhorn.py; done; done; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/sk_ab.log
