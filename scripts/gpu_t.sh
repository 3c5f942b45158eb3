#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "head_size_128" 2>&1 | tail -8
