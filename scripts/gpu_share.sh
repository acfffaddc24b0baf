#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "shared_prefix or batch or fork or parallel or 32_slot or fp8" 2>&1 | tail -5
for cfg in "detikzify-ds-7b bf16 32" "detikzify-cl-7b fp8 32" "detikzify-v2-8b bf16 32"; do set -- $cfg
timeout 900 python bench.py --model $1 --weight-format $2 --steps 1 --warmup 1 --no-cpu-baseline --batch $3 --probe-tokens 16 > "$OUT/bench_b.log" 2> "$OUT/bench_b.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_b.log").read().strip().splitlines()[-1]); b=d["batched_rollouts"]; print("$1 $2 B=$3: tok/s", round(d["value"],1), "| batched rollouts/s", round(b["rollouts_per_sec"],2), "tok/s", round(b["tokens_per_sec"]), "steps", b["decode_steps"], "ms/batch", round(b["ms_per_batch"]), b["engine_seconds"])
PY
done
