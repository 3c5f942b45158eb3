#!/bin/bash
# One gpurun call: environment facts, smoke, GPU parity tests, a bench line.  Everything is logged
# under gpurun_out/ (merged back into the build container).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; free -g | head -2; } > $OUT/host.txt 2>&1
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -20; rocm-smi --showmeminfo vram | head -8; } > $OUT/gpu.txt 2>&1
echo "== smoke" > $OUT/smoke.log
timeout 600 python __graft_entry__.py smoke >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
tail -5 $OUT/smoke.log
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|PASSED|FAILED|ERROR|\[c|\[mid|\[flags|\[nodesc" $OUT/pytest_gpu.log | tail -70
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log
tail -3 $OUT/bench.log
