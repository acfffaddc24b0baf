#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sampl or batch or fork or engine" 2>&1 | tail -4
for extra in "" "--sample"; do
timeout 600 python bench.py $extra --no-cpu-baseline --batch 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ds-7b $extra decode tok/s', round(d['decode_tokens_per_sec_per_gpu'],1))"
done
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 32 --probe-tokens 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batched_rollouts']; print('ds-7b B=32 rollouts/s', round(b['rollouts_per_sec'],2), b['engine_seconds'])"
