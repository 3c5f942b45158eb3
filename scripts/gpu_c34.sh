#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_larger_baseline_configs | tail -2;
  for c in C2 C3 C4; do timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline; done;
  rm -rf gpurun_out/prof3; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -o t -- python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cp gpurun_out/prof3/t_kernel_stats.csv gpurun_out/kernel_stats_c3.csv; find gpurun_out/prof3 -name "*kernel_trace.csv" -delete; head -8 gpurun_out/kernel_stats_c3.csv | cut -c1-160; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c34.log
