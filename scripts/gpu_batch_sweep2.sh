#!/bin/bash
# per-pair time of the C2 model vs the number of pairs handed to one call (Infinity-Cache residency of the activations vs chip occupancy)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/batch_sweep.jsonl
for b in 32 16 8 4 64; do
  timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'batch': $b, 'pairs_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'ms_per_pair': round(d['ms_per_step'] / $b, 4), 'stages': d['stages_ms']}))" | tee -a gpurun_out/batch_sweep.jsonl
done
