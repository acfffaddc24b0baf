#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for v in 0 1; do
  DTK_F8_VARIANT=$v timeout 300 python bench.py --model detikzify-cl-7b --weight-format fp8 --steps 1 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 16 > "$OUT/bench_f8_v$v.log" 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/bench_f8_v$v.log").read().strip().splitlines()[-1]); print("f8 variant $v: decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), "gate/up us", round(d["roofline"].get("avg_launch_us",0),2))
PY
done
timeout 300 python bench.py --model detikzify-ds-7b --steps 1 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 16 > "$OUT/bench_q.log" 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/bench_q.log").read().strip().splitlines()[-1]); print("bf16 ds-7b: decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), "gate/up us", round(d["roofline"].get("avg_launch_us",0),2))
PY
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "not full_size and not ds13b" -p no:cacheprovider 2>&1 | tail -3
