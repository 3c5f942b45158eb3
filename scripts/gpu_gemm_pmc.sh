#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf gpurun_out/pmc_gemm; mkdir -p gpurun_out/pmc_gemm; export TMPDIR=/tmp
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_gemm/$name -o p -- python scripts/bench_gemm.py > gpurun_out/pmc_gemm/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
run sq3 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_DATA_FIFO_FULL
find gpurun_out/pmc_gemm -name "*kernel_trace.csv" -delete
