#!/bin/bash
# streaming Sinkhorn geometry (rows per workgroup, non-temporal loads): tests, microbench at the config-3 / config-4 shapes, bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "sinkhorn or configs or ragged" -p no:cacheprovider 2>&1 | tail -3
{ for shape in 32,2048,2048,100 8,4096,4096,100 16,1500,1800,100 2,257,1000,100; do
    OG_SK_SHAPE=$shape timeout 100 python scripts/bench_sinkhorn.py 2>&1 | grep shape | sed "s/^/auto          /"
    OG_SK_FAST_ROWS=32 OG_SK_FAST_NT=0 OG_SK_SHAPE=$shape timeout 100 python scripts/bench_sinkhorn.py 2>&1 | grep shape | sed "s/^/rows=32 nt=0   /"
  done; } | tee gpurun_out/sk_stream.log
rm -f gpurun_out/bench_configs.jsonl
for c in C3 C4 C5; do timeout 600 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_configs.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/bench_configs.jsonl"):
    d = json.loads(l); print(d["metric"], d["value"], d["ms_per_step"], d["stages_ms"])
PY
