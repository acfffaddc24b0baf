#!/usr/bin/env bash
# One GPU-box session: parity tests, smoke, a short bench, a rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [quick|full]
set -uo pipefail
MODE=${1:-quick}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
echo "== rocminfo"; (rocminfo | grep -E "Marketing Name|gfx" | head -4) 2>&1 | tee "$OUT/device.txt"
nproc | tee -a "$OUT/device.txt"; free -g | head -2 | tee -a "$OUT/device.txt"

echo "== pytest -m gpu"
if [ "$MODE" = quick ]; then KARGS=(-k "not full_size"); else KARGS=(); fi
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s "${KARGS[@]}" -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/pytest_gpu.log"
grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/smoke.log"

echo "== bench"
BM=${BENCH_MODEL:-detikzify-ds-7b}
timeout 900 python bench.py --model $BM --steps 2 --warmup 1 > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/bench.err"
tail -c 3000 "$OUT/bench.log"

echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$REPO/bench.py" --model $BM --steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline > "$OUT/prof_bench.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"
find "$OUT/prof" -name "*stats*" | head; 
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do head -30 "$f"; done
