#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 300 python scripts/bench_attention.py; timeout 600 python -m pytest tests -m gpu -q -k "attention or forward_against" -p no:cacheprovider | tail -3; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/attn_ab.log
