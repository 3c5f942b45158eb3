#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 600 python -m pytest tests -m gpu -q -k "edge_shapes" -p no:cacheprovider 2>&1 | tail -30; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/misc.log
