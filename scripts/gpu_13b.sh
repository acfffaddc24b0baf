#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for fm in 0 4096; do
DTK_ATTN_FULL_MAX=$fm timeout 600 python bench.py --model detikzify-ds-1.3b --no-cpu-baseline --batch 0 --steps 2 > "$OUT/bench_13b.log" 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_13b.log").read().strip().splitlines()[-1]); print("ds-1.3b attn_full_max=$fm: tok/s", round(d["value"],1), "decode", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
done
