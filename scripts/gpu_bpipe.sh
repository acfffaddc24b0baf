#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fp8 or batch or v2_batched" 2>&1 | tail -5
for w in 16 8; do for f in bf16 fp8; do echo "resid_waves=$w $f: $(DTK_GB_RESID_WAVES=$w timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --batch 16 --steps 48 --weight-format $f 2>&1 | tail -1)"; done; done
