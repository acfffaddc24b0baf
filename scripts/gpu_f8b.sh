#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fp8 or batch" 2>&1 | tail -8
for fmt in fp8 bf16; do
timeout 900 python bench.py --model detikzify-cl-7b --weight-format $fmt --steps 1 --warmup 1 --no-cpu-baseline --batch 16 --probe-tokens 16 > "$OUT/bench_b_$fmt.log" 2> "$OUT/bench_b_$fmt.err"; tail -2 "$OUT/bench_b_$fmt.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_b_$fmt.log").read().strip().splitlines()[-1]); print("$fmt cl-7b: tok/s", round(d["value"],1), {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if "batch" in k or "rollout" in k})
PY
done
