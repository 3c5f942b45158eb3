#!/bin/bash
# round 6, call p: the whole-path parity suites with the non-default attention paths FORCED through og_forward: the pipelined loop, the 16x16x32 kernel,
# the workgroup-level key split at 4 / 2 parts with the parts of a tile dealt to different XCDs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r06p_pytest_forced_paths.txt; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider 2>&1 | tail -22 >> $OUT; }
run OG_ATTN_PIPE=1
run OG_ATTN_P16=1
run OG_ATTN_GSPLIT=4 OG_ATTN_GS_SCATTER=1
run OG_ATTN_GSPLIT=2 OG_ATTN_GS_SCATTER=1
grep "^==\|passed\|failed" $OUT
