#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 600 python scripts/bench_streams.py; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/misc.log
