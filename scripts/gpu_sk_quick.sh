#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "sinkhorn or forward or c2 or C3 or ragged or edge" -p no:cacheprovider 2>&1 | tail -5
timeout 120 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sk.log
OG_SINKHORN_ROBUST=1 timeout 120 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids | sed 's/^/robust-only  /' | tee -a gpurun_out/sk.log
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/bench_quick.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])"
