#!/bin/bash
# round-6 reference measurement: smoke, the bench line of the driver (defaults: 50 steps / 10 warm-up, CPU baseline), C1/C3/C4/C5 lines WITH
# their CPU baselines, B=1 / B=4 / B=64 lines at the C2 shape, rocprofv3 kernel stats of C2..C5, optionally ("pmc") the SQ counter and
# traffic passes; kernel stats of B=1 / B=4 steps; the granted shader clock per kernel.   usage: OG_COMMIT=<hash> gpu_round6.sh <tag> [pmc] [tests]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
TAG="${1:-r06x}"
mkdir -p $OUT
nproc > $OUT/${TAG}_host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/${TAG}_host.txt
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
if [[ " $* " == *" tests "* ]]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
  tail -16 $OUT/${TAG}_pytest_gpu.log
fi
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read()); print("BENCH", d["value"], d["ms_per_step"], d["step_ms_spread"], d["stages_ms"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"]); print("TRAIN", d.get("training_step"), d.get("training_step_reference_shape"))
PY
: > $OUT/${TAG}_bench_configs.jsonl
for c in C1 C3 C4 C5 S128 S256 C4i20; do timeout 900 python bench.py --config $c --steps 20 --warmup 5 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_configs.jsonl; done
# the reference's own operating point for its 128-d family, ONE pair per call (inference.py:214-235) and four; C2 shape at B = 1 / 2 / 4 / 16 / 64
for b in 1 4; do timeout 600 python bench.py --config S128 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-training-step 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_configs.jsonl; done
# ... and for its 256-d family (SuperPoint: 2048 keypoints, 20 iterations), one pair per call
timeout 600 python bench.py --config S256 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-training-step 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_configs.jsonl
for b in 1 2 4 16 64; do timeout 600 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-training-step 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_configs.jsonl; done
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_configs.jsonl"):
    d = json.loads(l); cb = d.get("cpu_baseline") or {}
    print(d["metric"], d["config"].get("pairs_per_gpu"), d["value"], d["ms_per_step"], d["stages_ms"], "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "| cpu", cb.get("value"), cb.get("cores"))
PY
for c in C2 C3 C4 C5 S128; do
  ( cd /tmp && rm -rf /tmp/prof_${TAG}_$c && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-training-step > /tmp/prof_${TAG}_$c.log 2>&1 )
  f=$(find /tmp/prof_${TAG}_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_$c.csv; echo "== $c"; head -9 $OUT/${TAG}_kernel_stats_$c.csv | cut -c1-150; else tail -5 /tmp/prof_${TAG}_$c.log; fi
done
# the small-batch kernels (mlp_small_kernel, proj_small_kernel, the key-split attention): kernel stats of a single-pair and a four-pair step
( cd /tmp && rm -rf /tmp/prof_${TAG}_S128B1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_S128B1 -o run -- python $GRAFT_REPO_ROOT/bench.py --config S128 --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-training-step > /tmp/prof_${TAG}_S128B1.log 2>&1 )
f=$(find /tmp/prof_${TAG}_S128B1 -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_S128_B1.csv; echo "== S128 B=1"; head -7 $OUT/${TAG}_kernel_stats_S128_B1.csv | cut -c1-150; fi
for b in 1 4; do
  ( cd /tmp && rm -rf /tmp/prof_${TAG}_B$b && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_B$b -o run -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-training-step > /tmp/prof_${TAG}_B$b.log 2>&1 )
  f=$(find /tmp/prof_${TAG}_B$b -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_B$b.csv; echo "== B=$b"; head -7 $OUT/${TAG}_kernel_stats_B$b.csv | cut -c1-150; fi
done
# the training step: time at 4 and 16 pairs, kernel stats of the 4-pair run
for b in 4 16; do B=$b timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step"; done > $OUT/${TAG}_train_step.log; cat $OUT/${TAG}_train_step.log
( cd /tmp && rm -rf /tmp/prof_${TAG}_train && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_train -o run -- python $GRAFT_REPO_ROOT/scripts/bench_train_step.py > /tmp/prof_${TAG}_train.log 2>&1 )
f=$(find /tmp/prof_${TAG}_train -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $OUT/${TAG}_train_kernel_stats.csv; echo "== training step (4 steps)"; head -12 $OUT/${TAG}_train_kernel_stats.csv | cut -c1-150; fi
# the shader clock the chip grants each hot kernel and the whole C2 step (power cap)
timeout 300 python scripts/clock_under_load.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_clock_under_load.log; tail -3 $OUT/${TAG}_clock_under_load.log | cut -c1-200
if [[ " $* " == *" pmc "* ]]; then
  bash scripts/gpu_pmc.sh > $OUT/${TAG}_pmc.log 2>&1; tail -3 $OUT/${TAG}_pmc.log
  bash scripts/gpu_traffic.sh > $OUT/${TAG}_traffic.log 2>&1
  python scripts/parse_traffic.py gpurun_out/traffic gpurun_out/traffic_c2.json > /dev/null 2>&1
  for c in C3 C4; do OG_TRAFFIC_CONFIG=$c bash scripts/gpu_traffic.sh gpurun_out/traffic_$c > $OUT/${TAG}_traffic_$c.log 2>&1; python scripts/parse_traffic.py gpurun_out/traffic_$c gpurun_out/traffic_${c,,}.json > /dev/null 2>&1; done
  ls -la gpurun_out/traffic_c*.json gpurun_out/pmc_summary.json
fi
