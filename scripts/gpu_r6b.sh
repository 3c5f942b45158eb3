#!/bin/bash
# round 6, call b: pipelined-tile probe; the key-split hand-over with scattered parts; the whole GPU suite on the build with the MX template + capacity-keyed workspace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06b; mkdir -p $OUT
timeout 300 openglue_amd/lib/probe_attn_pipe > $OUT/${TAG}_probe_attn_pipe.log 2>&1; echo "probe rc=$?"; cat $OUT/${TAG}_probe_attn_pipe.log | cut -c1-230
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "placement_independent or single_pair_regime" 2>&1 | tail -15
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/${TAG}_pytest_gpu_tail.txt; tail -8 $OUT/${TAG}_pytest_gpu_tail.txt
