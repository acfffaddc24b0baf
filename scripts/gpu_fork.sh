#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -s -k "batch or parallel or fork" -p no:cacheprovider 2>&1 | tail -6
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --probe-tokens 8 > "$OUT/bench_fork.log" 2>"$OUT/bench_fork.err"; tail -2 "$OUT/bench_fork.err"
python - <<PY
import json
d=json.loads(open("$OUT/bench_fork.log").read().strip().splitlines()[-1]); print(json.dumps(d.get("batched_rollouts"), indent=0))
PY
