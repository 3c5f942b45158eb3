#!/bin/bash
# round 5, first GPU call: the 128-d kernel family -- its stage tests first (fail early), then the whole GPU suite, then bench lines at C4 and at the
# reference's SIFT operating point (S128: 128-d x 2048 kpts x 20 iterations) with the fused / small-batch kernels ON and OFF inside ONE call (A/B).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05a}"; mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "mlp_block or proj_block or stage_taps or d128 or large_shapes" > $OUT/${TAG}_pytest_new.log 2>&1; echo "pytest(new) rc=$?" >> $OUT/${TAG}_pytest_new.log
tail -25 $OUT/${TAG}_pytest_new.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
tail -30 $OUT/${TAG}_pytest_gpu.log
: > $OUT/${TAG}_bench_ab.jsonl
for env in "" "OG_MLP_FUSED=0" ; do
  for cfg in "C4" "S128" "S128 --batch 1" "S128 --batch 4"; do
    echo "== $env $cfg" | tee -a $OUT/${TAG}_bench_ab.jsonl
    env $env timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_ab.jsonl
  done
done
env OG_PROJ_SMALL=0 OG_MLP_SMALL=0 timeout 600 python bench.py --config S128 --batch 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_ab.jsonl
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    print(d["metric"], d["config"].get("pairs_per_gpu"), d["value"], d["ms_per_step"], d["stages_ms"], "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read()); print("BENCH", d["value"], d["ms_per_step"], d["step_ms_spread"], d["stages_ms"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("multi_process"))
print(json.dumps(d["roofline_other"]["sinkhorn"])[:1500])
PY
for c in C4 S128; do
  ( cd /tmp && rm -rf /tmp/prof_${TAG}_$c && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > /tmp/prof_${TAG}_$c.log 2>&1 )
  f=$(find /tmp/prof_${TAG}_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_$c.csv; echo "== $c"; head -12 $OUT/${TAG}_kernel_stats_$c.csv | cut -c1-170; else tail -5 /tmp/prof_${TAG}_$c.log; fi
done
