#!/usr/bin/env bash
# Short closing run: the host-path tests touched by the engine / metric / processor changes, then BASELINE config 5's
# literal shape on one GPU (cl-7b fp8, 8 images x 4 rollouts in one batched decode).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 170 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider \
  -k "pipeline_end_to_end or batch_engine or kv_fork_prefix or simulate_parallel or engine_prefix or several_images" > "$OUT/pytest_last.log" 2>&1
echo "pytest exit $?"; tail -4 "$OUT/pytest_last.log"
timeout 150 python bench.py --model detikzify-cl-7b --weight-format fp8 --no-cpu-baseline --steps 1 --warmup 1 --probe-tokens 4 \
  --batch 32 --batch-images 8 > "$OUT/bench_cl7b_fp8_8img.log" 2> "$OUT/bench_cl7b_fp8_8img.err"
echo "bench exit $?"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_cl7b_fp8_8img.log").read().strip().splitlines()[-1]); b = d["batched_rollouts"]
    print("cl-7b fp8 8 images x 4:", {k: b.get(k) for k in ("rollouts_per_sec", "images_in_flight", "prefix_encodes_both_passes", "decode_steps", "engine_seconds", "error")})
except Exception as e:
    print("no bench line", e)
PY
tail -3 "$OUT/bench_cl7b_fp8_8img.err"
