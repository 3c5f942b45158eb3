#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{ echo "== 32x32x16"; timeout 300 openglue_amd/lib/probe_attn_pipe; echo "== every MFMA as two 16x16x32 (timing only)"; timeout 300 openglue_amd/lib/probe_attn_pipe16; } 2>&1 | tee gpurun_out/r06m_probe_attn_pipe_16x16x32.log | cut -c1-60,100-240
