#!/bin/bash
# round-3 reference measurement: smoke, the whole GPU suite, the bench line (with the CPU baseline), C1/C3/C4/C5 lines, the fused-MLP
# microbench, rocprofv3 kernel stats of the bench command, optionally ("pmc") the SQ counter and traffic passes.
# usage: OG_COMMIT=<hash> gpu_round3.sh <tag> [pmc]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
TAG="${1:-r03x}"
mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.log
tail -4 $OUT/${TAG}_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read()); print("BENCH", d["value"], d["ms_per_step"], d["stages_ms"], "cpu", d["cpu_baseline"]["value"])
PY
: > $OUT/${TAG}_bench_configs.jsonl
for c in C1 C3 C4 C5; do timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_bench_configs.jsonl; done
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_configs.jsonl"):
    d = json.loads(l); print(d["metric"], d["value"], d["ms_per_step"], d["stages_ms"])
PY
timeout 300 python scripts/bench_mlp_fused.py > $OUT/${TAG}_mlp_micro.log 2>&1; grep "M=" $OUT/${TAG}_mlp_micro.log
timeout 300 python scripts/bench_gemm.py > $OUT/${TAG}_gemm_micro.log 2>&1; tail -9 $OUT/${TAG}_gemm_micro.log
( cd /tmp && rm -rf /tmp/prof_$TAG && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/prof_$TAG.log 2>&1 )
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $OUT/${TAG}_kernel_stats.csv; head -16 $OUT/${TAG}_kernel_stats.csv | cut -c1-160; else tail -5 /tmp/prof_$TAG.log; find /tmp/prof_$TAG | head; fi
if [ "${2:-}" = "pmc" ]; then
  bash scripts/gpu_pmc.sh > $OUT/${TAG}_pmc.log 2>&1; tail -3 $OUT/${TAG}_pmc.log
  bash scripts/gpu_traffic.sh > $OUT/${TAG}_traffic.log 2>&1
  python scripts/parse_traffic.py gpurun_out/traffic gpurun_out/traffic_c2.json > /dev/null 2>&1; ls -la gpurun_out/traffic_c2.json gpurun_out/pmc_summary.json
fi
