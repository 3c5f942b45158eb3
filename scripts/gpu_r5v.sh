#!/bin/bash
# round 5: where the fp32 GEMM should switch from 128 x 64 to 128 x 128 tiles: training step at 4 and 16 pairs, inference C2 / B = 1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r05v_train_ab.log
for rep in 1 2; do
  for b in 4 16; do
    for t in 0 512 1024 2048 4096 100000000; do
      echo "== B=$b OG_GEMM_F32_NARROW_BELOW=$t" >> $OUT/r05v_train_ab.log
      B=$b OG_GEMM_F32_NARROW_BELOW=$t timeout 300 python scripts/bench_train_step.py 2>&1 | grep "training step" | cut -c1-120 >> $OUT/r05v_train_ab.log
    done
  done
done
cat $OUT/r05v_train_ab.log
for t in 0 512 100000000; do
  echo "== inference OG_GEMM_F32_NARROW_BELOW=$t"
  OG_GEMM_F32_NARROW_BELOW=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stages_ms'])"
  OG_GEMM_F32_NARROW_BELOW=$t timeout 300 python bench.py --config C2 --batch 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stages_ms'])"
done 2>&1 | tee $OUT/r05v_infer_ab.log
