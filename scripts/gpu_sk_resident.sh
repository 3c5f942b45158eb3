#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
echo "== resident tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sinkhorn" -p no:cacheprovider -s 2>&1 | grep -E "^\[sinkhorn res|passed|failed|Error|error|assert" | head -40
echo "== microbench resident"; OG_SINKHORN_RESIDENT=1 timeout 120 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids
echo "== microbench streaming"; OG_SINKHORN_RESIDENT=0 timeout 120 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
