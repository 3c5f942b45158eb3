#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -q -k "ragged" -p no:cacheprovider 2>&1 | tail -15; timeout 600 python bench.py --config C5 --steps 3 --warmup 1; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/misc.log
