#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE cycles / kernel duration
cd "${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf gpurun_out/clock; mkdir -p gpurun_out/clock; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clock -o p -- python scripts/traffic_driver.py > gpurun_out/clock/run.log 2>&1; echo rc=$?
ls gpurun_out/clock
