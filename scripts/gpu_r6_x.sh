#!/usr/bin/env bash
# round 6, lease X — the prefill switch test in full, then the whole GPU suite on the new prefill path
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06x}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "prefill_kernel_switches" 2>&1 | tail -60 | cut -c1-220 > "$OUT/${R}_pytest_switch.txt"; grep -n "^E \|passed\|failed\|rel_l2" "$OUT/${R}_pytest_switch.txt" | head
( time timeout 2400 python -m pytest tests -m gpu -q --durations=10 -rA ) 2>&1 | grep -v "^PASSED\|^SKIPPED" > "$OUT/${R}_pytest_gpu_full.txt"; tail -22 "$OUT/${R}_pytest_gpu_full.txt" | cut -c1-200
