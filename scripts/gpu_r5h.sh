#!/bin/bash
# round 5: 8 rows per wave for 9..16 pairs -- Sinkhorn tests + A/B at B = 12 / 16
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05h}"; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "sinkhorn" > $OUT/${TAG}_pytest_sinkhorn.log 2>&1; echo "pytest(sinkhorn) rc=$?" >> $OUT/${TAG}_pytest_sinkhorn.log
tail -6 $OUT/${TAG}_pytest_sinkhorn.log
: > $OUT/${TAG}_bench_ab.jsonl
for cfg in "C2 --batch 12" "C2 --batch 16"; do
  for env in "OG_SINKHORN_FEW=0" "OG_X=0"; do
    echo "== $env $cfg" >> $OUT/${TAG}_bench_ab.jsonl
    env $env timeout 600 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_ab.jsonl
  done
done
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    print(d["metric"], d["config"].get("pairs_per_gpu"), d["value"], d["ms_per_step"], d["stages_ms"])
PY
