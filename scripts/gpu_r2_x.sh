#!/usr/bin/env bash
# round 2, call X: k_gemm_px — bit-identity with k_gemm_mfma, prefill time
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "streamed_prefill_gemm" > "$OUT/r2x_pytest.log" 2>&1
echo "pytest exit $?"; tail -12 "$OUT/r2x_pytest.log" | cut -c1-300
timeout 600 python tools/bench_prefill.py 2>&1 | grep gemm_px
