#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ OG_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline;
  OG_BENCH_FORCE_DIST=1 timeout 600 python bench.py --config C5 --steps 3 --warmup 1; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/misc.log
