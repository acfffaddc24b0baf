#!/usr/bin/env bash
# round 2, call U: batched ViT + combined reward calls (test), then the MCTS phase of bench.py
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "vit_batch_rows or resume_slot or vit_features or pipeline_end_to_end or selfsim" > "$OUT/r2u_pytest.log" 2>&1
echo "pytest exit $?"; tail -5 "$OUT/r2u_pytest.log" | cut -c1-300
bash scripts/gpu_r2_r.sh 2>&1 | tail -9
