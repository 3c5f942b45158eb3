#!/bin/bash
# final checks of the round: the GPU suite on the two-launch message-MLP path as well (OG_MLP_FUSED=0), then the reference call
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
OG_MLP_FUSED=0 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "fixture or stage_taps or trained or c5 or c3 or ragged or edge or favor or hipgraph" > $OUT/pytest_unfused.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_unfused.log
echo "== OG_MLP_FUSED=0"; tail -3 $OUT/pytest_unfused.log
bash scripts/gpu_round3.sh "$@"
