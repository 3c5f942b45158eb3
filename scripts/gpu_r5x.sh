#!/bin/bash
# round 5: fp32 GEMM with 64 x 64 tiles for launches of few workgroups; split-K reduce with four waves per column group
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r05x; mkdir -p $OUT
OG_GEMM_F32_BM=64 timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "gemm or train or encoder or scores or kmajor or conv" > $OUT/${TAG}_pytest_bm64.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_bm64.log; tail -4 $OUT/${TAG}_pytest_bm64.log
timeout 600 python -m pytest tests/test_train_slice.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log; tail -3 $OUT/${TAG}_pytest.log
for env in "OG_GEMM_F32_BM=128" "OG_GEMM_F32_BM=64" "OG_X=0"; do
  echo "== $env"; env $env timeout 300 python scripts/bench_gemm_f32_small.py 2>&1 | tail -8
done > $OUT/${TAG}_gemm_micro.log 2>&1
cat $OUT/${TAG}_gemm_micro.log
: > $OUT/${TAG}_train.log
for rep in 1 2; do
  for env in "OG_GEMM_F32_BM=128" "OG_GEMM_F32_SHORT_BELOW=256" "OG_X=0" "OG_GEMM_F32_SHORT_BELOW=1024" "OG_GEMM_F32_BM=64"; do
    for b in 4 16; do
      echo "== $env B=$b" >> $OUT/${TAG}_train.log
      B=$b env $env timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | cut -c1-120 >> $OUT/${TAG}_train.log
    done
  done
done
cat $OUT/${TAG}_train.log
