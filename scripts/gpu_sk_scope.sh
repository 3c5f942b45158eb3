#!/bin/bash
# Sinkhorn resident kernel: XCD-local vs agent-scope granule exchange (tests, microbench, phase trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sinkhorn_resident" -p no:cacheprovider 2>&1 | tail -2
OG_SINKHORN_AGENT_SCOPE=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sinkhorn_resident" -p no:cacheprovider 2>&1 | tail -2
echo "== XCD-local"; OG_SINKHORN_RESIDENT=1 timeout 120 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids
echo "== agent scope"; OG_SINKHORN_AGENT_SCOPE=1 OG_SINKHORN_RESIDENT=1 timeout 120 python scripts/bench_sinkhorn.py 2>&1 | grep -v amdgpu.ids
if [ -f openglue_amd/lib/libog_sk_trace.so ]; then
  export OPENGLUE_AMD_LIB=openglue_amd/lib/libog_sk_trace.so
  echo "== trace, XCD-local"; timeout 120 python scripts/trace_sinkhorn.py 2>&1 | grep -v amdgpu.ids
  echo "== trace, agent scope"; OG_SINKHORN_AGENT_SCOPE=1 timeout 120 python scripts/trace_sinkhorn.py 2>&1 | grep -v amdgpu.ids
fi
