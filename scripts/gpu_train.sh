#!/bin/bash
# training path: its parity tests, then the training-step bench with the round-2 kernels (exact fp32, materialised attention) and the round-3 ones
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_slice.py -m gpu -q --timeout 600 -p no:cacheprovider -s > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_train.log
grep -E "passed|failed|FAILED|rc=|worst|Error" $OUT/pytest_train.log | tail -20
: > $OUT/train_step.log
OG_TRAIN_F16X3=0 OG_TRAIN_FLASH=0 timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | sed 's/^/round-2 kernels: /' >> $OUT/train_step.log
OG_TRAIN_F16X3=1 OG_TRAIN_FLASH=0 timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | sed 's/^/split-f16 convs:  /' >> $OUT/train_step.log
OG_TRAIN_F16X3=0 OG_TRAIN_FLASH=1 timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | sed 's/^/flash forward:    /' >> $OUT/train_step.log
timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | sed 's/^/both (default):   /' >> $OUT/train_step.log
cat $OUT/train_step.log
