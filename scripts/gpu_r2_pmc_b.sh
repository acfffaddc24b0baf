#!/usr/bin/env bash
# round 2: HBM fetch bytes per launch of the 64-slot batched kernels (own --pmc pass)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_pb" -o pmc -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 8 --fork > "$OUT/r2_pmc_b.log" 2>&1; echo "rocprof pmc exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_pb -name 'pmc_results.db' | head -1)" "$OUT/r02_batch64_pmc_fetch.csv" --pmc > /dev/null; rm -rf "$OUT/prof_pb"; head -8 "$OUT/r02_batch64_pmc_fetch.csv" | cut -c1-220
