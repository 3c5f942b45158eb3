#!/bin/bash
# round 6, final call: smoke, the whole GPU suite, the driver's bench line, the training-step lines, kernel stats of C2, a roctx marker capture and the granted
# shader clocks -- on the round's final build.   usage: OG_COMMIT=<hash> gpu_r6z.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06z; mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
tail -40 $OUT/${TAG}_pytest_gpu.log > $OUT/${TAG}_pytest_gpu_tail.txt; rm $OUT/${TAG}_pytest_gpu.log; tail -4 $OUT/${TAG}_pytest_gpu_tail.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read()); print("BENCH", d["value"], d["ms_per_step"], d["stages_ms"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
print("TRAIN", d.get("training_step", {}).get("ms_per_step"), d.get("training_step_reference_shape", {}).get("ms_per_step"))
PY
( cd /tmp && rm -rf /tmp/prof_z && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_z -o run -- python $GRAFT_REPO_ROOT/bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-training-step > /tmp/prof_z.log 2>&1 )
f=$(find /tmp/prof_z -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_C2.csv; head -6 $f | cut -c1-150; fi
# roctx ranges of the stages next to the kernel trace (no counters in this pass)
( cd /tmp && rm -rf /tmp/prof_m && OG_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --output-format csv -d /tmp/prof_m -o run -- python $GRAFT_REPO_ROOT/bench.py --config C1 --steps 2 --warmup 1 --no-cpu-baseline --no-training-step > /tmp/prof_m.log 2>&1 )
f=$(find /tmp/prof_m -name "*marker*trace*.csv" | head -1); if [ -n "$f" ]; then head -40 $f | cut -c1-200 > $OUT/${TAG}_roctx_marker_trace_head.csv; wc -l $f; head -12 $OUT/${TAG}_roctx_marker_trace_head.csv; else echo "no marker csv"; ls -R /tmp/prof_m | head; tail -5 /tmp/prof_m.log; fi
timeout 300 python scripts/clock_under_load.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_clock_under_load.log; tail -6 $OUT/${TAG}_clock_under_load.log | cut -c1-200
for b in 4 16; do B=$b timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step"; done > $OUT/${TAG}_train_step.log; cat $OUT/${TAG}_train_step.log
