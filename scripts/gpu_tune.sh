#!/usr/bin/env bash
# GPU session: quick parity tests, GEMV variant sweep, A/B of the attention-combine placement.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -s -k "not full_size" -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^FAILED" "$OUT/pytest_gpu.log"
timeout 600 python tools/tune_gemv.py --model ${BENCH_MODEL:-detikzify-ds-7b} --out "$OUT/tune_gemv.json" 2>&1 | tee "$OUT/tune_gemv.log"
for mode in inkernel consumer; do
  DTK_ATTN_COMBINE=$mode timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_$mode.log" 2> "$OUT/bench_$mode.err"
  echo "combine=$mode exit $?"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$mode.log").read().strip().splitlines()[-1])
    print("$mode", "tok/s", round(d["value"],1), "decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), "roofline", d.get("roofline"))
except Exception as e: print("parse fail", e)
PY
done
