#!/bin/bash
# round 5: at K = 256 the stream projection kernel for the self-layer launches only (OG_PROJ_STREAM=2) against the default (tile GEMMs), alternating inside one call
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r05s_bench_ab.jsonl
for rep in 1 2; do
for cfg in "C2" "C3" "C5"; do
  for env in "OG_X=0" "OG_PROJ_STREAM=2"; do
    echo "== $env $cfg" >> $OUT/r05s_bench_ab.jsonl
    env $env timeout 600 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 >> $OUT/r05s_bench_ab.jsonl
  done
done
done
python - <<PY
import json
for l in open("gpurun_out/r05s_bench_ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    print(d["metric"][:28], d["value"], d["ms_per_step"], d["stages_ms"]["gemm_f16x3"])
PY
OG_PROJ_STREAM=2 timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "c2 or C2 or reference_fixture or c5" 2>&1 | tail -3
