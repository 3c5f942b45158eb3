#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_attention_forms.py tests/test_gpu_configs.py -x -q -m gpu -k "forms or pipelined or block_scaled or workspace_is_keyed or roctx or placement" 2>&1 | tail -15
