#!/bin/bash
# round 6, call k: flash backward at two workgroups per CU (dS transpose tiles and hand-over area inside the dead Q | dO tiles): training tests + A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06k; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_slice.py -x -q -m gpu 2>&1 | tail -6
{
for rep in 1 2; do
  echo "== one workgroup per CU (-DOG_ATTN_BWD_WGS=1)"; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_bwd_wgs1.so B=4 timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step"
  echo "== two workgroups per CU (default)"; B=4 timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step"
done
} > $OUT/${TAG}_train_step_bwd_wgs_ab.log 2>&1; cat $OUT/${TAG}_train_step_bwd_wgs_ab.log
( cd /tmp && rm -rf /tmp/prof_train && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o run -- python $GRAFT_REPO_ROOT/scripts/bench_train_step.py > /tmp/prof_train.log 2>&1 )
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $OUT/${TAG}_train_kernel_stats.csv; head -6 $f | cut -c1-160; fi
