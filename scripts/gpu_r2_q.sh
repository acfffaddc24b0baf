#!/usr/bin/env bash
# round 2, call Q: the whole GPU suite + smoke on the current tree
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/r2q_pytest.log" 2>&1
echo "pytest exit $?"; tail -8 "$OUT/r2q_pytest.log" | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
