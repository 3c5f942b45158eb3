#!/bin/bash
# A/B of 256-tile GEMM variants inside one gpurun call: $2 = the environment switch to flip (default OG_GEMM_FAST_EPI)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "${1:-gemm or fixture or c2}" 2>&1 | tail -4
SW=${2:-OG_GEMM_FAST_EPI}
for v in 0 1; do
  echo "== $SW=$v"
  env $SW=$v timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
  env $SW=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
done
