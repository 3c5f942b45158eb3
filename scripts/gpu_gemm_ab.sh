#!/bin/bash
# A/B of the 256-tile GEMM epilogues inside one gpurun call (OG_GEMM_FAST_EPI=0: generic epilogue)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "${1:-gemm or fixture or c2}" 2>&1 | tail -4
for v in 0 1; do
  echo "== OG_GEMM_FAST_EPI=$v"
  OG_GEMM_FAST_EPI=$v timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
  OG_GEMM_FAST_EPI=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
done
