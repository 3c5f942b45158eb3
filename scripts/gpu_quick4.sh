#!/bin/bash
# quick call: selected GPU tests + one bench line.  usage: gpu_quick4.sh "<pytest -k expression>"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
K="${1:-timeout or status}"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "$K" > $OUT/pytest_quick.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_quick.log
grep -E "passed|failed|error|FAILED|ERROR|rc=|assert|Error" $OUT/pytest_quick.log | tail -15
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_quick.err | tail -1 > $OUT/bench_quick.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_quick.json").read()); print("BENCH", d["value"], d["ms_per_step"], d["stages_ms"]); print(json.dumps(d["roofline_other"]["sinkhorn"])[:500])
PY
tail -3 $OUT/bench_quick.err
