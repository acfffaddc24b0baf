#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sampl or batch or fork or engine or multiblock" 2>&1 | tail -4
for m in detikzify-ds-7b detikzify-v2-8b; do for extra in "" "--sample"; do
timeout 600 python bench.py --model $m $extra --no-cpu-baseline --batch 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m $extra decode tok/s', round(d['decode_tokens_per_sec_per_gpu'],1))"
done; done
