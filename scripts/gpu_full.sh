#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
BM=${BENCH_MODEL:-detikzify-ds-7b}
timeout 1800 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^FAILED|ds-1.3b|ds-7b|rccl" "$OUT/pytest_gpu.log" | head
timeout 600 python tools/tune_gemv.py --model $BM --out "$OUT/tune_gemv.json" 2>&1 | tail -12
timeout 600 python bench.py --model $BM --no-cpu-baseline > "$OUT/bench.log" 2> "$OUT/bench.err"; tail -c 1500 "$OUT/bench.log"
timeout 600 python bench.py --model detikzify-ds-1.3b --no-cpu-baseline > "$OUT/bench_13b.log" 2> "$OUT/bench_13b.err"; tail -c 1200 "$OUT/bench_13b.log"
