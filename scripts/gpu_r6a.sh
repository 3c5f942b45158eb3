#!/bin/bash
# round 6, call a: hardware facts for the 8-bit P V cross products (probe) + A/B of the attention kernel's MX form with torch-made 8-bit V rows
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; TAG=r06a; mkdir -p $OUT
timeout 120 openglue_amd/lib/probe_mx_pv > $OUT/${TAG}_probe_mx_pv.log 2>&1; echo "probe rc=$?"
grep -v "^  lane" $OUT/${TAG}_probe_mx_pv.log | head -40
sed -n '/pattern 1/,/pattern 2/p' $OUT/${TAG}_probe_mx_pv.log | head -36
timeout 600 python scripts/bench_attention_mx.py > $OUT/${TAG}_attention_mx_ab.log 2>&1; echo "ab rc=$?"
cat $OUT/${TAG}_attention_mx_ab.log | grep -v amdgpu.ids
timeout 600 python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline --no-training-step > $OUT/${TAG}_bench_C2.json 2> $OUT/${TAG}_bench_C2.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06a_bench_C2.json").read().strip().splitlines()[-1])
    print("C2:", d["value"], d["unit"], d["ms_per_step"], "ms;", {k: v for k, v in d.get("stage_ms", {}).items()} if "stage_ms" in d else "")
except Exception as e:
    print("bench parse failed", e)
PY
