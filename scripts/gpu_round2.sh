#!/bin/bash
# Round-2 measurement call: smoke, all GPU tests, bench lines of every config, kernel trace, PMC + traffic passes.
# usage: gpu_round2.sh [tests|notests] [pmc|nopmc]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $OUT/host.txt 2>&1
if [ "${1:-tests}" = "tests" ]; then
  echo "== smoke" > $OUT/smoke.log
  timeout 600 python __graft_entry__.py smoke >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
  tail -3 $OUT/smoke.log
  echo "== pytest"
  timeout 2400 python -m pytest tests -m gpu -q -rA --timeout 900 -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  grep -E "passed|failed|error|FAILED|ERROR|^\[|rc=|s call" $OUT/pytest_gpu.log | tail -60
fi
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log
tail -2 $OUT/bench.log | cut -c1-1500
echo "== gemm microbench"
timeout 300 python scripts/bench_gemm.py > $OUT/gemm_micro.log 2>&1; cat $OUT/gemm_micro.log | tail -8
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
echo "== rocprof"
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" >> $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
head -16 $OUT/kernel_stats.csv | cut -c1-200
if [ "${2:-pmc}" = "pmc" ]; then
  echo "== pmc"
  bash scripts/gpu_pmc.sh
  echo "== traffic"
  bash scripts/gpu_traffic.sh > $OUT/traffic.log 2>&1
  python scripts/parse_traffic.py gpurun_out/traffic gpurun_out/traffic_c2.json > /dev/null 2>&1; head -c 1500 gpurun_out/traffic_c2.json
fi
echo "== probes"
for p in lds_read_patterns mfma_valu_interleave store_pattern dispatch_ramp; do
  [ -x openglue_amd/lib/probe_$p ] && timeout 60 openglue_amd/lib/probe_$p > $OUT/probe_$p.log 2>&1
done
tail -4 $OUT/probe_lds_read_patterns.log 2>/dev/null
