#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OPENGLUE_AMD_LIB=openglue_amd/lib/libog_attn_trace.so timeout 300 python scripts/trace_attention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_trace_dma.log
