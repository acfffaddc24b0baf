#!/usr/bin/env bash
# round 2, call J: k_gemv_bx — bit-identity with k_gemv_b at the real widths, then its time for the gate/up role at 64 slots
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "x_once_per_cu" > "$OUT/r2j_pytest.log" 2>&1
echo "pytest exit $?"; tail -15 "$OUT/r2j_pytest.log" | cut -c1-300
timeout 600 python tools/probe_batch.py --no-lds --bx > "$OUT/r2j_probe_batch.log" 2>&1; echo "probe exit $?"; grep slots "$OUT/r2j_probe_batch.log"
