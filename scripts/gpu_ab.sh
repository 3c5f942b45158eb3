#!/bin/bash
# A/B of one environment switch inside ONE call (boxes differ by 3-4 %): usage gpu_ab.sh VAR A_VALUE B_VALUE [reps]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
VAR=$1; A=$2; B=$3; REPS=${4:-2}
rm -f $OUT/ab.jsonl
for r in $(seq 1 $REPS); do
  for v in $A $B; do
    env $VAR=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'var': '$VAR', 'val': '$v', 'value': d['value'], 'ms': d['ms_per_step'], 'stages': d['stages_ms']}))" | tee -a $OUT/ab.jsonl
  done
done
