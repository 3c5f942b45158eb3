#!/bin/bash
# round 5: the training step -- fp32 GEMM tile width A/B (OG_GEMM_F32_BN), the step's glue by torch operator / call site, train parity tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_train_slice.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/r05u_pytest.log 2>&1; echo "rc=$?" >> $OUT/r05u_pytest.log; tail -4 $OUT/r05u_pytest.log
: > $OUT/r05u_train_ab.log
for rep in 1 2; do
  for env in "OG_GEMM_F32_BN=128" "OG_X=0" "OG_GEMM_F32_BN=64"; do
    echo "== $env" >> $OUT/r05u_train_ab.log
    env $env timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" >> $OUT/r05u_train_ab.log
  done
done
cat $OUT/r05u_train_ab.log
timeout 300 python scripts/profile_train_glue.py > $OUT/r05u_train_glue.log 2>&1; tail -75 $OUT/r05u_train_glue.log | cut -c1-200
