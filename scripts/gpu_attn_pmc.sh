#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf gpurun_out/pmc_attn; mkdir -p gpurun_out/pmc_attn; export TMPDIR=/tmp
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_attn/$name -o p -- python scripts/bench_attention.py > gpurun_out/pmc_attn/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
run sq3 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT
find gpurun_out/pmc_attn -name "*kernel_trace.csv" -delete
