#!/bin/bash
# f16x3 GEMM: correctness tests, microbench of the default build and of the compile-time ablation builds
# (scripts/build_ablation.sh gabl_N -DOG_GEMM_ABL=N; profiling only), optional SQ counters (PMC=1).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc_gemm; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "f16x3 or split_f16 or forward or ragged" -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/gemm_tests.log
{ timeout 120 python scripts/bench_gemm.py | sed 's/^/abl=0  /'
  for f in openglue_amd/lib/libog_gabl_*.so; do [ -e "$f" ] || continue; n=${f##*_}; n=${n%.so}; OPENGLUE_AMD_LIB=$PWD/$f timeout 120 python scripts/bench_gemm.py | sed "s/^/abl=$n  /"; done; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/gemm_ablate.log
if [ "$PMC" = 1 ]; then
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_gemm/$name -o p -- python scripts/bench_gemm.py > gpurun_out/pmc_gemm/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
find gpurun_out/pmc_gemm -name "*kernel_trace.csv" -delete
fi
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/bench_quick.json | cut -c1-400
