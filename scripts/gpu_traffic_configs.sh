#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the uniform BASELINE configs 3 and 4 (streaming Sinkhorn), for their roofline objects
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in C3 C4; do
  OG_TRAFFIC_CONFIG=$c bash scripts/gpu_traffic.sh gpurun_out/traffic_$c > gpurun_out/traffic_$c.log 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python scripts/parse_traffic.py gpurun_out/traffic_$c gpurun_out/traffic_$lc.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('gpurun_out/traffic_$lc.json')); print('$c', {k: v['hbm_bytes_per_launch'] for k, v in d.items() if isinstance(v, dict) and 'hbm_bytes_per_launch' in v})"
done
for c in C3 C4; do timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['stages_ms'])"; done
