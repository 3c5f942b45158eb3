#!/bin/bash
# One gpurun call: environment facts, smoke, GPU parity tests, a bench line, a rocprofv3 kernel trace.
# Everything is logged under gpurun_out/ (merged back into the build container).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $OUT/host.txt 2>&1
echo "== smoke" > $OUT/smoke.log
timeout 600 python __graft_entry__.py smoke >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|FAILED|ERROR|^\[|rc=" $OUT/pytest_gpu.log | tail -40
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log
tail -2 $OUT/bench.log
if [ "${1:-}" != "noprof" ]; then
  echo "== rocprof"
  rm -rf $OUT/prof; mkdir -p $OUT/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" >> $OUT/rocprof.log
  tail -2 $OUT/rocprof.log
  find $OUT/prof -name "*kernel_stats*" | head; 
  for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
  # the raw per-dispatch trace is large: keep only the stats
  find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
  head -30 $OUT/kernel_stats.csv
fi
