#!/bin/bash
# training path after a change: its parity tests, the training-step bench, kernel stats of the bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=${1:-train}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_slice.py -m gpu -q --timeout 600 -p no:cacheprovider -s > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
grep -E "passed|failed|FAILED|rc=|worst|Error" $OUT/${TAG}_pytest.log | tail -24
timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | tee $OUT/${TAG}_step.log
rm -rf $OUT/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python scripts/bench_train_step.py > $OUT/${TAG}_prof.log 2>&1
f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/${TAG}_kernel_stats.csv; head -14 "$f" | cut -c1-200; else echo "no kernel stats"; tail -5 $OUT/${TAG}_prof.log; fi
