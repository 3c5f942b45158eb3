#!/bin/bash
# A/B of the dh=64 attention kernels inside one gpurun call: register-staged (OG_ATTN_DMA=0) vs LDS-DMA (default)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "${1:-attention or fixture or c5 or ragged or c2}" 2>&1 | tail -5
for v in "OG_ATTN_DMA=0" "OG_ATTN_DMA=1"; do
  echo "== bench $v"
  env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
done
