#!/bin/bash
# SQ / GRBM counter passes over two whole hot-path steps at C2 (scripts/traffic_driver.py), each pass its own rocprofv3 run
# (--pmc only with --kernel-trace: gpurun refuses --pmc together with the sys / hip / memory trace domains).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -o p -- python scripts/traffic_driver.py > gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
run grbm GRBM_GUI_ACTIVE
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
python scripts/parse_pmc.py gpurun_out/pmc gpurun_out/pmc_summary.json | tail -60
