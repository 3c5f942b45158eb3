#!/bin/bash
# round 5: training step after the BatchNorm statistics kernels were re-tiled, the BatchNorm output is kept, stacked weights are cached; split-K target sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r05y; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "gemm or train or encoder or scores or batchnorm or colsum or grad" > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log
: > $OUT/${TAG}_train.log
for rep in 1 2; do
  for env in "OG_TRAIN_KEEP_BN=0" "OG_TRAIN_SPLITK_WGS=512" "OG_X=0" "OG_TRAIN_SPLITK_WGS=1024" "OG_TRAIN_SPLITK_WGS=1536"; do
    for b in 4 16; do
      echo "== $env B=$b" >> $OUT/${TAG}_train.log
      B=$b env $env timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | cut -c1-140 >> $OUT/${TAG}_train.log
    done
  done
done
cat $OUT/${TAG}_train.log
timeout 300 python scripts/profile_train_glue.py > $OUT/${TAG}_train_glue.log 2>&1; grep -n "ms  x" $OUT/${TAG}_train_glue.log | head -60 | cut -c1-110
