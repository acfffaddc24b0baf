#!/usr/bin/env bash
# round 2: k_gemv_bx with fp8 weights — bit-identity, cl-7b fp8 batched phase with / without
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "x_once_per_cu or fp8" > "$OUT/r2f8_pytest.log" 2>&1
echo "pytest exit $?"; tail -4 "$OUT/r2f8_pytest.log" | cut -c1-300
for bx in 0 1; do
  DTK_OPTIONS="gemv_bx=$bx" timeout 900 python bench.py --model detikzify-cl-7b --weight-format fp8 --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2f8_bench.log" 2> "$OUT/r2f8_bench.err"
  python - "$OUT/r2f8_bench.log" "gemv_bx=$bx" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print(sys.argv[2], "| value", round(d["value"], 1), "| batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]))
PY
done
