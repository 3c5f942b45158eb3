#!/bin/bash
# round 5: the training step with its glue on single launches (strided flash backward, split / merge / split-K reduce kernels): parity tests + step time
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=${1:-r05w}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_train_slice.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log; tail -15 $OUT/${TAG}_pytest.log
: > $OUT/${TAG}_train.log
for rep in 1 2; do
  for b in 4 16; do
    B=$b timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | cut -c1-140 >> $OUT/${TAG}_train.log
  done
done
cat $OUT/${TAG}_train.log
timeout 300 python scripts/profile_train_glue.py > $OUT/${TAG}_train_glue.log 2>&1; grep -n "ms  x" $OUT/${TAG}_train_glue.log | head -48 | cut -c1-110
