#!/bin/bash
# round 5: the workgroup-level key split of the attention kernel at dh = 32 (single pairs of the 128-d family) -- tests with the split forced each way, A/B bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05i}"; mkdir -p $OUT
for gs in 0 2 4 ""; do
  echo "=== OG_ATTN_GSPLIT=${gs:-default}" >> $OUT/${TAG}_pytest.log
  if [ -n "$gs" ]; then export OG_ATTN_GSPLIT=$gs; else unset OG_ATTN_GSPLIT; fi
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "single_pair_regime or d128 or (forward_against_reference_fixture) or attention" >> $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log
done
unset OG_ATTN_GSPLIT
grep -E "===|passed|failed|rc=|single-pair" $OUT/${TAG}_pytest.log | head -40
: > $OUT/${TAG}_bench_ab.jsonl
for cfg in "S128 --batch 1" "S128 --batch 2" "S128 --batch 4" "C4 --batch 1"; do
  for env in "OG_ATTN_GSPLIT=0" "OG_X=0"; do
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
