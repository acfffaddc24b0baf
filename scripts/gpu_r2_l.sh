#!/usr/bin/env bash
# round 2, call L: k_gemv_bx with a loader wave — bit-identity, gate/up timing per block shape, per-kernel profile of the 64-slot step
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "x_once_per_cu" > "$OUT/r2l_pytest.log" 2>&1
echo "pytest exit $?"; tail -15 "$OUT/r2l_pytest.log" | cut -c1-300
timeout 600 python tools/probe_batch.py --no-lds --bx > "$OUT/r2l_probe_batch.log" 2>&1; echo "probe exit $?"; grep slots "$OUT/r2l_probe_batch.log"
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS="gemv_bx=1" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_bx" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_bx.log" 2>&1; echo "rocprof exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_bx -name trace_results.db | head -1)" "$OUT/r02_batch64_bx_kernel_stats.csv" > /dev/null 2>&1
rm -rf "$OUT/prof_bx"; head -6 "$OUT/r02_batch64_bx_kernel_stats.csv" | cut -c1-150
