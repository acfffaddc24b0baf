#!/usr/bin/env bash
set -uo pipefail
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "batch or fp8 or 32_slot" 2>&1 | tail -3
for ks in 4 8; do for f in bf16 fp8; do for b in 16 32; do echo "ks=$ks $f $(DTK_GB_RESID_KS=$ks timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --batch $b --steps 48 --weight-format $f 2>&1 | tail -1)"; done; done; done
