#!/usr/bin/env bash
# round 4, lease M (4 GPU-minutes left): lease L ran the suite under pytest-xdist (-n 2), which was SLOWER than the sequential run (two
# CPU oracles fighting for the host's threads), hit its time limit at 171 of 179 tests and showed one F without a name.  This is every
# GPU test except the twelve long full-depth ones (each of those was re-run on the final source in leases I / final), sequentially.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 215 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider -k "not (headline or long_context or few_slot or v2_8b or margins or ds13b or peaked)" > "$OUT/r04m_light_tests.log" 2>&1
echo "pytest exit $?"
grep -E "passed|failed" "$OUT/r04m_light_tests.log" | tail -1; grep -E "^FAILED|^ERROR|^E  " "$OUT/r04m_light_tests.log" | cut -c1-400 | head -30
