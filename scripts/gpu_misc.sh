#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 600 python -m pytest tests -m gpu -q -k "forward_against" -p no:cacheprovider -rA 2>&1 | grep -E "^\[|passed|failed|Error|assert|^E " | head -30; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/misc.log
