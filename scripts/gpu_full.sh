#!/bin/bash
# whole GPU suite + bench line + rocprof kernel stats (the round's reference call)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
TAG="${1:-x}"
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_$TAG.log
tail -4 $OUT/pytest_gpu_$TAG.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_$TAG.json
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read()); print("BENCH", d["value"], d["ms_per_step"], d["stages_ms"]); print(json.dumps(d.get("roofline"))[:600])
PY
cd /tmp && rm -rf /tmp/prof_$TAG && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$TAG.csv && head -14 $OUT/kernel_stats_$TAG.csv | cut -c1-150
