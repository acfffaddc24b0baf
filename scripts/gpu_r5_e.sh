#!/usr/bin/env bash
# round 5, lease E: the whole GPU suite as the driver runs it (pytest -m gpu, one process, collection order of tests/conftest.py:
# newest kernels first, full-size CPU-oracle comparisons last) on the final kernels, with durations and the figures the tests print;
# then smoke().
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA -p no:cacheprovider --durations=25 ) > "$OUT/r05_pytest_gpu.log" 2>&1
echo "pytest exit $?"
{ grep -E "^[0-9]+ (passed|failed)|passed|failed" "$OUT/r05_pytest_gpu.log" | tail -1; grep -E "^real" "$OUT/r05_pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/r05_pytest_gpu.log"
  echo "== slowest"; grep -E "^[0-9.]+s (call|setup)" "$OUT/r05_pytest_gpu.log" | head -25
  echo "== figures printed by the tests"
  grep -vE "^$|^PASSED|^SKIPPED|^=|^-|^_|amdgpu.ids|[Ww]arning|^  |^[.sFE]+ +\[|^[0-9.]+s (call|setup)|Captured stdout" "$OUT/r05_pytest_gpu.log" | cut -c1-2500 | head -400; } > "$OUT/r05_pytest_gpu_summary.txt"
head -3 "$OUT/r05_pytest_gpu_summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/r05_smoke.txt"
