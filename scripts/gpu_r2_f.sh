#!/usr/bin/env bash
# round 2, call F: the whole GPU suite (incl. the full-size oracle comparisons), smoke, the default bench line
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/r2f_pytest.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/r2f_pytest.log" | tail -2; grep -E "^FAILED|^ERROR" "$OUT/r2f_pytest.log"
grep -E "vs fp32|rel_l2|greedy [0-9]+/" "$OUT/r2f_pytest.log" | cut -c1-260
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > "$OUT/r2f_bench.log" 2> "$OUT/r2f_bench.err"; echo "bench exit $?"; tail -c 6000 "$OUT/r2f_bench.log"; tail -5 "$OUT/r2f_bench.err"
