#!/usr/bin/env bash
# round 2, call H: k_gemm_dma — bit-identity with k_gemm_mfma, then ViT / prefill time per tile and ring
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "op_gemm" > "$OUT/r2h_pytest.log" 2>&1
echo "pytest exit $?"; tail -6 "$OUT/r2h_pytest.log"
timeout 600 python tools/tune_gemm.py --model detikzify-ds-7b > "$OUT/tune_gemm_ds7b.log" 2>&1; echo "tune exit $?"; grep impl "$OUT/tune_gemm_ds7b.log"
