#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
timeout 600 python bench.py --model detikzify-v2-8b --steps 2 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 16 > "$OUT/bench_v2.log" 2> "$OUT/bench_v2.err"; tail -3 "$OUT/bench_v2.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_v2.log").read().strip().splitlines()[-1]); print("v2-8b: tok/s", round(d["value"],1), "decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), {k: d[k] for k in d if "prefill" in k or "frac" in k or "hbm" in k})
PY
