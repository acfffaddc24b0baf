#!/usr/bin/env bash
# round 6, lease Z — the two peaked-head runs that lease Y failed, with their report lines
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06z}
timeout 1200 python -m pytest tests/test_gpu_parity_batched.py -m gpu -q -k "peaked_logits and (ds-1.3b or cl-7b)" -rA 2>&1 | grep -v "^PASSED\|^SKIPPED" | cut -c1-700 > "$OUT/${R}_pytest.txt"; grep -n "^E  \|passed\|failed\|^FAILED\|peaked weight set" "$OUT/${R}_pytest.txt" | cut -c1-700 | head -20
