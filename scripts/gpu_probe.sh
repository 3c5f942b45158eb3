#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 openglue_amd/lib/probe_attn_pipe 2>&1 | tee gpurun_out/r06e_probe_attn_pipe.log | cut -c1-240
