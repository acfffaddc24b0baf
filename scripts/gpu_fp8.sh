#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -s -k "fp8 or sample or greedy or variants" -p no:cacheprovider > "$OUT/pytest_fp8.log" 2>&1
echo "pytest exit $?"; tail -12 "$OUT/pytest_fp8.log"
timeout 600 python bench.py --model detikzify-cl-7b --weight-format fp8 --steps 2 --warmup 1 --no-cpu-baseline --batch 0 > "$OUT/bench_cl7b_fp8.log" 2> "$OUT/bench_cl7b_fp8.err"; echo "exit $?"; tail -3 "$OUT/bench_cl7b_fp8.err"; tail -c 1800 "$OUT/bench_cl7b_fp8.log"
timeout 600 python bench.py --model detikzify-ds-7b --steps 2 --warmup 1 --no-cpu-baseline --batch 0 > "$OUT/bench_quick.log" 2>/dev/null; tail -c 700 "$OUT/bench_quick.log" | head -c 400
