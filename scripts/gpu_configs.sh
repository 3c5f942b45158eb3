#!/bin/bash
# bench lines of all BASELINE configs on one GPU (no CPU baseline) + single-pair latency
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for c in C1 C2 C3 C4 C5; do timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1; done; } | tee gpurun_out/bench_configs.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d.get('stages_ms'))"
timeout 300 python scripts/bench_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency.log
