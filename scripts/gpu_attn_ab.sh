#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ for i in 1 2; do timeout 300 python scripts/bench_attention.py; OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/libopenglue_amd_setprio.so timeout 300 python scripts/bench_attention.py; done; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/attn_ab.log
