#!/bin/bash
# round 4: the rewritten on-chip-resident Sinkhorn (linear-domain state, any width <= 4096, rounds of co-resident pairs, ragged):
# parity tests, micro-benchmarks resident vs streaming for the BASELINE shapes, phase traces, bench lines.   usage: gpu_r4_sk.sh <tag> [full]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; TAG="${1:-r04a}"; OUT=gpurun_out/${TAG}_sk.log
{
echo "== sinkhorn + ragged tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -k "sinkhorn or ragged or c5" -p no:cacheprovider -s 2>&1 | grep -E "^\[sinkhorn|passed|failed|Error|error|assert|FAILED" | head -80
for shape in 32,1024,1024,100 64,1024,1024,100 32,2048,2048,100 8,4096,4096,100 1,1024,1024,100; do
  for r in 1; do echo "== microbench shape $shape resident=$r"; OG_SK_SHAPE=$shape OG_SINKHORN_RESIDENT=$r timeout 300 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
bash scripts/build_ablation.sh sk_trace -DOG_SK_TRACE=1 > /dev/null 2>&1
for shape in 32,1024,1024,100 8,2048,2048,100 2,4096,4096,100; do
  echo "== trace $shape"; OG_SK_SHAPE=$shape OPENGLUE_AMD_LIB=openglue_amd/lib/libog_sk_trace.so timeout 300 python scripts/trace_sinkhorn.py 2>&1 | grep -v amdgpu.ids
done
if [ "${2:-}" = "full" ]; then
  for c in C2 C3 C4 C5; do echo "== bench $c"; timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])"; done
fi
} > $OUT 2>&1
tail -150 $OUT
