#!/bin/bash
# Round-2, second session: all GPU tests, bench line, A/B of the compile-time GEMM epilogues, GEMM microbench, other configs, kernel trace.
# usage: gpu_round3.sh [tests|notests] [configs|noconfigs]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $OUT/host.txt 2>&1
if [ "${1:-tests}" = "tests" ]; then
  echo "== pytest"
  timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 600 -p no:cacheprovider --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  grep -E "passed|failed|error|FAILED|ERROR|^\[|rc=|s call" $OUT/pytest_gpu.log | tail -70
fi
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log
tail -2 $OUT/bench.log | cut -c1-600
python - <<'PY'
import json
for f in ("gpurun_out/bench.log",):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print("BENCH", d["value"], d["ms_per_step"], d["stages_ms"], "frac", d["roofline"]["frac"])
PY
echo "== bench with the run-time epilogues (OG_GEMM_SPEC_EPI=0)"
OG_GEMM_SPEC_EPI=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_spec0.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_spec0.json").read()); print("SPEC0", d["value"], d["ms_per_step"], d["stages_ms"])
PY
echo "== gemm microbench (spec epilogues, then run-time epilogues)"
timeout 300 python scripts/bench_gemm.py > $OUT/gemm_micro.log 2>&1; tail -9 $OUT/gemm_micro.log
OG_GEMM_SPEC_EPI=0 timeout 300 python scripts/bench_gemm.py > $OUT/gemm_micro_spec0.log 2>&1; tail -9 $OUT/gemm_micro_spec0.log
if [ "${2:-configs}" = "configs" ]; then
  echo "== configs"
  rm -f $OUT/bench_configs.jsonl
  for c in C1 C3 C4 C5; do timeout 600 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_configs.jsonl; done
  python - <<'PY'
import json
for l in open("gpurun_out/bench_configs.jsonl"):
    try:
        d = json.loads(l); print(d["metric"], d["value"], d["ms_per_step"], d["stages_ms"])
    except Exception as e:
        print("bad line", e, l[:200])
PY
fi
echo "== rocprof"
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" >> $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
head -20 $OUT/kernel_stats.csv | cut -c1-200
