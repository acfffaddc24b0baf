#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for pf in 0 1 2 4; do
  DTK_PREFETCH=$pf timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 4 > "$OUT/bench_pf$pf.log" 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/bench_pf$pf.log").read().strip().splitlines()[-1]); print("prefetch=$pf decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
done
DTK_PREFETCH=2 timeout 600 python -m pytest tests -m gpu -q --tb=short -k "greedy or graph or decode_logits" -p no:cacheprovider 2>&1 | tail -2
