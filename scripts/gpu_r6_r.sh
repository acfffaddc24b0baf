#!/usr/bin/env bash
# round 6, lease R — the whole GPU suite with the oracle's torch pool sized from the container's CFS quota (tests/conftest.py), then smoke()
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06r}
( time timeout 2400 python -m pytest tests -m gpu -q --durations=15 -rA ) 2>&1 | grep -v "^PASSED\|^SKIPPED" > "$OUT/${R}_pytest_gpu_full.txt"; tail -25 "$OUT/${R}_pytest_gpu_full.txt" | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/${R}_smoke.txt"
