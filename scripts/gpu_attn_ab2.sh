#!/bin/bash
# A/B of attention builds inside one gpurun call: tests on the default build, then microbench + bench line per library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "attention or fixture or c5 or ragged or c2 or c3" 2>&1 | tail -3
for tag in "" "$@" ""; do
  lib=openglue_amd/lib/libopenglue_amd.so; [ -n "$tag" ] && lib=openglue_amd/lib/libog_$tag.so
  echo "== ${tag:-as built}"
  OPENGLUE_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_attention.py 2>&1 | grep "us "
  OPENGLUE_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
done
