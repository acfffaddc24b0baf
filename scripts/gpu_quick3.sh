#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "not full_size and not ds13b" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/tune_gemv.py --model detikzify-ds-7b --out "$OUT/tune_gemv.json" 2>&1 | grep -v same | tail -6
timeout 300 python bench.py --model detikzify-ds-7b --steps 2 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 16 > "$OUT/bench_q.log" 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/bench_q.log").read().strip().splitlines()[-1]); print("bf16 ds-7b: tok/s", round(d["value"],1), "decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), "gate/up us", round(d["roofline"].get("avg_launch_us",0),2))
PY
timeout 300 python bench.py --model detikzify-cl-7b --weight-format fp8 --steps 2 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 16 > "$OUT/bench_f8.log" 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/bench_f8.log").read().strip().splitlines()[-1]); print("fp8 cl-7b: tok/s", round(d["value"],1), "decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), "gate/up us", round(d["roofline"].get("avg_launch_us",0),2))
PY
