#!/usr/bin/env bash
set -uo pipefail
export PYTHONUNBUFFERED=1
for f in bf16 fp8; do for b in 32; do echo "$f $(timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --batch $b --steps 48 --weight-format $f 2>&1 | tail -1)"; done; done
