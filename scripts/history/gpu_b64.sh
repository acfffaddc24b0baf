#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "slot or batch or fork or fp8 or engine or shared" 2>&1 | tail -5
for f in bf16 fp8; do for b in 32 64; do echo "$f $(timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --batch $b --steps 48 --weight-format $f --fork 2>&1 | tail -1)"; done; done
for b in 48 64; do
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch $b --probe-tokens 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batched_rollouts']; print('ds-7b B=$b rollouts/s', round(b['rollouts_per_sec'],2), 'tok/s', round(b['tokens_per_sec']), b['engine_seconds'])"
done
