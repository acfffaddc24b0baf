#!/usr/bin/env bash
# round 6, lease G: the whole GPU suite with the tightened parity constants (envelope 1.15 x, halved slacks, reference goldens at 1e-2,
# near-tie flips against the oracle's own gap histogram), the peaked-head identity test for every BASELINE model, both engines, k_gemv_bc
# on by default; then smoke().
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -q --durations=25 -rA 2>&1 | grep -v "^PASSED\|^SKIPPED" > "$OUT/r06g_pytest_gpu_full.txt"
tail -60 "$OUT/r06g_pytest_gpu_full.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/r06g_smoke.txt"
