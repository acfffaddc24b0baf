#!/usr/bin/env bash
set -uo pipefail
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "32_slot or batch or fork or fp8 or v2_batched" 2>&1 | tail -6
timeout 600 python tools/probe_batch.py 2>&1 | tail -4
for f in bf16 fp8; do for b in 16 32; do echo "$f $(timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --batch $b --steps 48 --weight-format $f 2>&1 | tail -1)"; done; done
