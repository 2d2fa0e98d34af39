#!/bin/bash
# round 5, call 12: after the read-back merge / cached attributes: search + dense tests, latency
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/test_gpu_index.py tests/test_gpu_search_scale.py tests/test_gpu_parity.py tests/test_gpu_faults.py tests/test_gpu_margin_modes.py tests/test_gpu_schedules.py -q -m gpu -k "search or index or nns or reader or route or fault or survive or dense or coverage or launch_map" 2>&1 | tail -5
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-120
python scripts/exp_latency.py 8 300 2>&1 | tail -2 | cut -c1-120
python scripts/exp_latency.py 64 300 2>&1 | tail -2 | cut -c1-120
python scripts/exp_latency.py 1000 50 2>&1 | tail -2 | cut -c1-120
