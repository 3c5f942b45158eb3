#!/bin/bash
# round 5: kernel stats of the C2 / B = 1 / B = 4 inference steps again (the r05z run's included the bench line's training leg)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r05z; mkdir -p $OUT
( cd /tmp && rm -rf /tmp/prof_C2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_C2 -o run -- python $GRAFT_REPO_ROOT/bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-training-step > /tmp/prof_C2.log 2>&1 )
f=$(find /tmp/prof_C2 -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_C2.csv; head -9 $f | cut -c1-150; fi
for b in 1 4; do
  ( cd /tmp && rm -rf /tmp/prof_B$b && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_B$b -o run -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-training-step > /tmp/prof_B$b.log 2>&1 )
  f=$(find /tmp/prof_B$b -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats_B$b.csv; echo "== B=$b"; head -7 $f | cut -c1-150; fi
done
