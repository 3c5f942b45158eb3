#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -q -k "larger_baseline" -p no:cacheprovider -rA | grep -E "^\[C|passed|failed|Error"; 
  timeout 600 python bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline; timeout 600 python bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c34.log
