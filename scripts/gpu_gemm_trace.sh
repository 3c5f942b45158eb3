#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
echo "== trace big2"; OPENGLUE_AMD_LIB=openglue_amd/lib/libog_gemm_trace.so timeout 300 python scripts/trace_gemm.py > gpurun_out/gemm_trace_big2.log 2>&1; echo "trace rc=$?"
grep -v amdgpu.ids gpurun_out/gemm_trace_big2.log
echo "== trace big2, HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 OPENGLUE_AMD_LIB=openglue_amd/lib/libog_gemm_trace.so timeout 300 python scripts/trace_gemm.py > gpurun_out/gemm_trace_big2_devkernarg.log 2>&1
grep -v amdgpu.ids gpurun_out/gemm_trace_big2_devkernarg.log | grep -E "===|prologue|stage period|epilogue|block life|CUs used|entry times"
echo "== trace big (first generation)"; OG_GEMM_BIG2=0 OPENGLUE_AMD_LIB=openglue_amd/lib/libog_gemm_trace.so timeout 300 python scripts/trace_gemm.py > gpurun_out/gemm_trace_big1.log 2>&1
grep -v amdgpu.ids gpurun_out/gemm_trace_big1.log | grep -E "===|prologue|stage period|epilogue|block life|CUs used|entry times"
echo "== microbench big2 devkernarg"; HIP_FORCE_DEV_KERNARG=1 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
echo "== microbench big1 devkernarg"; HIP_FORCE_DEV_KERNARG=1 OG_GEMM_BIG2=0 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
echo "== bench devkernarg"; HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
echo "== bench default"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
