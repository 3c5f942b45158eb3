#!/bin/bash
# first GPU contact of the fused message-MLP kernel: its stage tests, the microbench, whole-path fixtures, one bench line (fused / unfused)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "mlp_block" > $OUT/pytest_mlp.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_mlp.log
grep -E "mlp_block|passed|failed|error|FAILED|ERROR|rc=|assert" $OUT/pytest_mlp.log | tail -25
timeout 300 python scripts/bench_mlp_fused.py > $OUT/mlp_micro.log 2>&1; tail -8 $OUT/mlp_micro.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fixture or c2 or ragged" > $OUT/pytest_path.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_path.log
grep -E "passed|failed|error|FAILED|ERROR|rc=" $OUT/pytest_path.log | tail -8
for f in 1 0; do
  OG_MLP_FUSED=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_fused$f.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_fused$f.json").read()); print("BENCH fused=$f", d["value"], d["ms_per_step"], d["stages_ms"])
PY
done
