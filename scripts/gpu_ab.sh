#!/bin/bash
# A/B/... of one environment switch inside ONE call (boxes differ by 3-4 %):  [REPS=2] [CFG="--config C4"] gpu_ab.sh VAR VALUE_A VALUE_B [VALUE_C ...]
# e.g. experiment builds:  gpu_ab.sh OPENGLUE_AMD_LIB openglue_amd/lib/libog_x.so openglue_amd/lib/libopenglue_amd.so
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
VAR=$1; shift
REPS=${REPS:-2}
rm -f $OUT/ab.jsonl
for r in $(seq 1 $REPS); do
  for v in "$@"; do
    env $VAR=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${CFG:-} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'var': '$VAR', 'val': '$v', 'value': d['value'], 'ms': d['ms_per_step'], 'stages': d['stages_ms']}))" | tee -a $OUT/ab.jsonl
  done
done
