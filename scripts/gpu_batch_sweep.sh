#!/bin/bash
# Per-pair time of the C2 workload as a function of the batch handed to one og_forward call
# (is the 32-pair working set too large for the 256 MB Infinity Cache / per-XCD L2?).
mkdir -p gpurun_out
for b in 2 4 8 16 32 64; do
  python bench.py --config C2 --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1
done | tee gpurun_out/batch_sweep.jsonl
