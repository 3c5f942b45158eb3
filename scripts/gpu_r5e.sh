#!/bin/bash
# round 5: proj_stream_kernel v2 (x fragments staged through LDS, epilogue planes deferred + staggered) -- stage tests, trace, micro-benchmark, whole-step A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05e}"; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "proj_block or d128 or c4 or stage_taps" > $OUT/${TAG}_pytest_new.log 2>&1; echo "pytest(new) rc=$?" >> $OUT/${TAG}_pytest_new.log
tail -8 $OUT/${TAG}_pytest_new.log
OG_PROJ_STREAM=1 OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/libog_trace.so timeout 300 python scripts/trace_proj.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_proj_trace.log; cat $OUT/${TAG}_proj_trace.log
{ echo "=== OG_PROJ_STREAM=1"; OG_PROJ_STREAM=1 timeout 300 python scripts/bench_proj.py 2>&1 | grep -v amdgpu.ids; } > $OUT/${TAG}_proj_micro.log 2>&1; cat $OUT/${TAG}_proj_micro.log
: > $OUT/${TAG}_bench_ab.jsonl
for cfg in "C2" "C3" "C4" "S128" "C5"; do
  for env in "OG_PROJ_STREAM=0" "OG_PROJ_STREAM=1" "OG_X=0"; do
    echo "== $env $cfg" >> $OUT/${TAG}_bench_ab.jsonl
    env $env timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_ab.jsonl
  done
done
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    print(d["metric"], d["config"].get("pairs_per_gpu"), d["value"], d["ms_per_step"], d["stages_ms"], "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
