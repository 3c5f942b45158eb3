#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
( cd /tmp && rm -rf /tmp/prof_train && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o run -- python $GRAFT_REPO_ROOT/scripts/bench_train_step.py > /tmp/prof_train.log 2>&1 )
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
cp $f $OUT/train_kernel_stats.csv; head -30 $OUT/train_kernel_stats.csv | cut -c1-200; wc -l $OUT/train_kernel_stats.csv; tail -2 /tmp/prof_train.log
