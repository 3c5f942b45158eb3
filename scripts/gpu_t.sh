#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_train_slice.py -x -q -m gpu -k "softmax_attention_backward" 2>&1 | tail -8
