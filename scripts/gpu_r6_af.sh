#!/usr/bin/env bash
# round 6, lease AF — the peaked-head runs with the symmetric criterion (device vs bf16 oracle, both against the fp32 oracle's greedy token)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06af}
( time timeout 1500 python -m pytest tests/test_gpu_parity_batched.py -m gpu -q -k "peaked_logits" -rA ) 2>&1 | grep -v "^PASSED\|^SKIPPED" | cut -c1-900 > "$OUT/${R}_pytest.txt"; grep -n "^E  \|passed\|failed\|^FAILED\|peaked weight set\|^real" "$OUT/${R}_pytest.txt" | cut -c1-900 | head -20
