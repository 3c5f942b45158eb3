#!/bin/bash
# usage: [OG_TRAFFIC_CONFIG=C3] gpu_traffic.sh [outdir]   (default gpurun_out/traffic)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; D=${1:-gpurun_out/traffic}; mkdir -p $D; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $D/$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$c -o p -- python scripts/traffic_driver.py > $D/$c.log 2>&1; echo "$c rc=$?"
  find $D/$c -name "*kernel_trace.csv" -delete
done
ls $D/*
