#!/usr/bin/env bash
# round 2, call Z: GQA-fused batched attention — bit-identity (tiny v2 G = 2, v2-8b shapes G = 4), v2-8b batched bench phase
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "gqa_heads or v2_" > "$OUT/r2z_pytest.log" 2>&1
echo "pytest exit $?"; tail -5 "$OUT/r2z_pytest.log" | cut -c1-300
for f in 0 1; do
  DTK_OPTIONS="gqa_fused=$f" timeout 900 python bench.py --model detikzify-v2-8b --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2z_bench.log" 2> "$OUT/r2z_bench.err"
  python - "$OUT/r2z_bench.log" "gqa_fused=$f" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print(sys.argv[2], "| value", round(d["value"], 1), "| batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]))
PY
done
