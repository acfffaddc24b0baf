#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
BM=${BENCH_MODEL:-detikzify-ds-7b}
timeout 900 python -m pytest tests -m gpu -q --tb=short -s -k "batch or parallel" -p no:cacheprovider > "$OUT/pytest_batch.log" 2>&1
echo "pytest(batch) exit $?"; tail -40 "$OUT/pytest_batch.log"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s -k "not batch and not parallel" -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest(rest) exit $?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^FAILED|ds-1.3b:|rccl" "$OUT/pytest_gpu.log" | head
timeout 900 python bench.py --model $BM --no-cpu-baseline > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench exit $?"; tail -5 "$OUT/bench.err"; tail -c 2500 "$OUT/bench.log"
