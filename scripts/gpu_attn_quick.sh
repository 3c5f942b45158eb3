#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "attention or forward or ragged or c2 or C3" -p no:cacheprovider 2>&1 | tail -4
timeout 120 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn.log
for f in openglue_amd/lib/libog_aabl_*.so; do [ -e "$f" ] || continue; n=${f##*_}; n=${n%.so}; OPENGLUE_AMD_LIB=$PWD/$f timeout 120 python scripts/bench_attention.py 2>&1 | grep -v amdgpu.ids | sed "s/^/abl=$n  /" | tee -a gpurun_out/attn.log; done
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/bench_quick.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'])"
