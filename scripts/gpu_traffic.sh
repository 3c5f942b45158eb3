#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/traffic; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic/$c -o p -- python scripts/traffic_driver.py > gpurun_out/traffic/$c.log 2>&1; echo "$c rc=$?"
  find gpurun_out/traffic/$c -name "*kernel_trace.csv" -delete
done
ls gpurun_out/traffic/*
