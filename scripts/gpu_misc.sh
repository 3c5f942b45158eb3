#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -q -k "prepare_features or compact" -p no:cacheprovider 2>&1 | tail -25; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/misc.log
