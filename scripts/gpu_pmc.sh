#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA)_[A-Z0-9_]+" | sort -u > gpurun_out/pmc/counters.txt
wc -l gpurun_out/pmc/counters.txt
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -o p -- python scripts/bench_attention.py > gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
find gpurun_out/pmc -name "*counter_collection.csv" | head
