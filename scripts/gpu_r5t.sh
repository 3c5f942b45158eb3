#!/bin/bash
# round 5: ragged launches rounded up to whole 256-row tiles (fast epilogue forms of the tile GEMMs) -- ragged parity tests + C5 A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "ragged or c5 or C5" > $OUT/r05t_pytest.log 2>&1; echo "rc=$?" >> $OUT/r05t_pytest.log; tail -6 $OUT/r05t_pytest.log
: > $OUT/r05t_bench_ab.jsonl
for rep in 1 2; do
  for env in "OG_RAGGED_PAD=0" "OG_X=0"; do
    echo "== $env C5" >> $OUT/r05t_bench_ab.jsonl
    env $env timeout 600 python bench.py --config C5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 >> $OUT/r05t_bench_ab.jsonl
  done
done
python - <<PY
import json
for l in open("gpurun_out/r05t_bench_ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l); print(d["value"], d["ms_per_step"], d["stages_ms"])
PY
