#!/bin/bash
# round 4: the rewritten on-chip-resident Sinkhorn (linear-domain state, any width <= 4096, rounds of co-resident pairs):
# parity tests, micro-benchmarks resident vs streaming for the BASELINE shapes, phase traces.   usage: gpu_r4_sk.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; TAG="${1:-r04a}"; OUT=gpurun_out/${TAG}_sk.log
{
echo "== sinkhorn tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sinkhorn" -p no:cacheprovider -s 2>&1 | grep -E "^\[sinkhorn|passed|failed|Error|error|assert|FAILED" | head -80
for shape in 32,1024,1024,100 64,1024,1024,100 32,2048,2048,100 8,4096,4096,100 1,1024,1024,100 4,1024,1024,100 16,1280,1280,100; do
  for r in 1 0; do echo "== microbench shape $shape resident=$r"; OG_SK_SHAPE=$shape OG_SINKHORN_RESIDENT=$r timeout 300 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
bash scripts/build_ablation.sh sk_trace -DOG_SK_TRACE=1 > /dev/null 2>&1
for shape in 32,1024,1024,100 8,2048,2048,100 2,4096,4096,100; do
  echo "== trace $shape"; OG_SK_SHAPE=$shape OPENGLUE_AMD_LIB=openglue_amd/lib/libog_sk_trace.so timeout 300 python scripts/trace_sinkhorn.py 2>&1 | grep -v amdgpu.ids
done
} > $OUT 2>&1
tail -120 $OUT
