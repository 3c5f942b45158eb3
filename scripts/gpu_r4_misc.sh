#!/bin/bash
# round 4: the whole GPU suite (incl. every pair of C3/C4 against the CPU oracle), the training step eager vs graphed, small-batch lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; TAG="${1:-r04g}"; OUT=gpurun_out/${TAG}_misc.log
{
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider 2>&1 | tail -25
echo "== training step"; timeout 600 python scripts/bench_train_step.py 2>&1 | grep -v amdgpu.ids | tail -3
for b in 1 4; do echo "== bench C2 --batch $b"; timeout 600 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms_spread'], d['stages_ms'])"; done
} > $OUT 2>&1
tail -60 $OUT
