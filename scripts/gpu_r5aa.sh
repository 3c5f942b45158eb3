#!/bin/bash
# round 5: Sinkhorn backward with 256 instead of 64 row workgroups per pair (cross-wave column sums in LDS before the atomics)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r05aa; mkdir -p $OUT
timeout 600 python -m pytest tests/test_train_slice.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log; tail -3 $OUT/${TAG}_pytest.log
: > $OUT/${TAG}_train.log
for rep in 1 2; do
  for g in 64 128 256 512; do
    for b in 4 16; do
      echo "== OG_SK_BWD_ROWS_GRID=$g B=$b" >> $OUT/${TAG}_train.log
      B=$b OG_SK_BWD_ROWS_GRID=$g timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | cut -c1-120 >> $OUT/${TAG}_train.log
    done
  done
done
cat $OUT/${TAG}_train.log
