#!/bin/bash
# the whole GPU suite; tail + parity notes -> gpurun_out/<tag>_pytest_gpu_tail.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r06}; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -60 > gpurun_out/${TAG}_pytest_gpu_tail.txt; tail -6 gpurun_out/${TAG}_pytest_gpu_tail.txt
