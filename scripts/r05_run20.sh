#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_small_calls.py -q -m gpu -x 2>&1 | tail -25
