#!/bin/bash
# round 5: SQ counter passes over two C4 steps (the 128-d kernels: mlp_fused_kernel<128>, proj_stream_kernel<128>, attention at dh = 32) + col_argmax A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
OG_TRAFFIC_CONFIG=C4 bash scripts/gpu_pmc.sh > $OUT/r05k_pmc_c4.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/r05k_pmc_summary_c4.json; tail -5 $OUT/r05k_pmc_c4.log
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "matches or extract or reference_fixture or ragged_pairs_of_the_128d" > $OUT/r05k_pytest.log 2>&1; echo "rc=$?" >> $OUT/r05k_pytest.log; tail -4 $OUT/r05k_pytest.log
for c in C4 C2; do
  ( cd /tmp && rm -rf /tmp/prof_k_$c && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > /tmp/prof_k_$c.log 2>&1 )
  f=$(find /tmp/prof_k_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/r05k_kernel_stats_$c.csv; grep -E "col_argmax|sinkhorn_scores" $OUT/r05k_kernel_stats_$c.csv | cut -c1-200; fi
done
