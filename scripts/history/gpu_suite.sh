#!/usr/bin/env bash
# Full GPU parity suite + smoke() (about a minute on an MI355X); the summary goes to profiles/r01_pytest_gpu_summary.txt.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"
{ grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log"
  sed -E 's/^[.sFE]+//' "$OUT/pytest_gpu.log" | grep -vE "^$|passed|failed|[Ww]arning|^  |^=|^-" | head -300; } > "$OUT/pytest_gpu_summary.txt"
head -3 "$OUT/pytest_gpu_summary.txt"; wc -l "$OUT/pytest_gpu_summary.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
