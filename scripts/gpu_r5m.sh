#!/bin/bash
# round 5: the whole GPU suite on the final build, then again with the non-default paths forced (proj_stream_kernel at both widths inside og_forward;
# the two-launch message MLP; 16 rows per wave everywhere in the resident Sinkhorn)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/r05m_pytest_default.log 2>&1; echo "rc=$?" >> $OUT/r05m_pytest_default.log; tail -3 $OUT/r05m_pytest_default.log
OG_PROJ_STREAM=1 timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -k "not train" > $OUT/r05m_pytest_proj_stream1.log 2>&1; echo "rc=$?" >> $OUT/r05m_pytest_proj_stream1.log; tail -3 $OUT/r05m_pytest_proj_stream1.log
OG_MLP_FUSED=0 OG_SINKHORN_FEW=0 timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -k "reference_fixture or configs or stage_taps or single_pair or ragged" > $OUT/r05m_pytest_unfused.log 2>&1; echo "rc=$?" >> $OUT/r05m_pytest_unfused.log; tail -3 $OUT/r05m_pytest_unfused.log
