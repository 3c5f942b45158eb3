#!/bin/bash
# round 5: key-split launches with up to two workgroups per CU (OG_ATTN_GS_MAXWG = 512, default) against one (256)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05j}"; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "single_pair_regime or d128 or (forward_against_reference_fixture) or attention or c_caller" > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log
tail -4 $OUT/${TAG}_pytest.log
: > $OUT/${TAG}_bench_ab.jsonl
for cfg in "S128 --batch 1" "S128 --batch 2" "C2 --batch 1" "C2 --batch 2" "C2 --batch 4" "C4 --batch 1"; do
  for env in "OG_ATTN_GS_MAXWG=256" "OG_X=0"; do
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
    print(d["metric"], d["config"].get("pairs_per_gpu"), d["value"], d["ms_per_step"], d["stages_ms"]["attention"])
PY
