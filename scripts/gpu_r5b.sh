#!/bin/bash
# round 5, second GPU call: proj_stream_kernel (the batch q / k / v projection) -- stage tests, whole suite, A/B bench lines inside ONE call
# (OG_PROJ_STREAM=0 = the tile GEMMs of round 4), kernel stats, and the fp8 / i8 variants of the MFMA energy probe.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG="${1:-r05b}"; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "proj_block" > $OUT/${TAG}_pytest_new.log 2>&1; echo "pytest(new) rc=$?" >> $OUT/${TAG}_pytest_new.log
tail -12 $OUT/${TAG}_pytest_new.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
tail -25 $OUT/${TAG}_pytest_gpu.log
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
for c in C2 C4; do
  ( cd /tmp && rm -rf /tmp/prof_${TAG}_$c && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > /tmp/prof_${TAG}_$c.log 2>&1 )
  f=$(find /tmp/prof_${TAG}_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_$c.csv; echo "== $c"; head -10 $OUT/${TAG}_kernel_stats_$c.csv | cut -c1-170; else tail -5 /tmp/prof_${TAG}_$c.log; fi
done
timeout 300 ./openglue_amd/lib/probe_mfma_energy > $OUT/${TAG}_probe_mfma_energy.log 2>&1; cat $OUT/${TAG}_probe_mfma_energy.log
