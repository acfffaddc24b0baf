#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "not full_size and not ds13b" -p no:cacheprovider 2>&1 | tail -4
for m in generic fast; do
  DTK_SAMPLER=$m timeout 300 python bench.py --sample --steps 1 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 4 > "$OUT/bench_s_$m.log" 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/bench_s_$m.log").read().strip().splitlines()[-1]); print("sampler=$m sampling decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
done
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 4 > "$OUT/bench_g.log" 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/bench_g.log").read().strip().splitlines()[-1]); print("greedy decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
