#!/bin/bash
# quick check after a kernel change: GPU tests (optionally filtered), GEMM microbench, one bench line, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${1:+-k "$1"} 2>&1 | tail -4
echo "== microbench"; timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_quick.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('value_incl_h2d'))"
echo "== kernel stats"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/rocprof.log 2>&1
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/kernel_stats.csv; done
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/kernel_stats.csv")))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s}%")
PY
