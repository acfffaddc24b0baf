#!/usr/bin/env bash
# round 2, call S: dtk_resume_slot on the device, then the MCTS phase of bench.py with returning trees resumed in place
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -s -k "resume_slot or batch_engine or pipeline_end_to_end" > "$OUT/r2s_pytest.log" 2>&1
echo "pytest exit $?"; grep -E "resume in place|passed|failed|Error" "$OUT/r2s_pytest.log" | cut -c1-300 | tail -8
bash scripts/gpu_r2_r.sh
