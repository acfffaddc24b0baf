#!/usr/bin/env bash
# round 4, lease L: (1) MFMA-busy counters (own --pmc pass) of the 64-slot cl-7b fp8 step on the fp8 matrix cores; (2) the FULL GPU suite
# on the final source, with the printed parity figures (lease G ran it at the start of the session: 136 passed, 2 failed).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/prof_mx_mfma" -o pmc -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 6 --fork --model detikzify-cl-7b --weight-format fp8 > "$OUT/prof_mx_mfma.log" 2>&1
echo "rocprof exit $?"
db=$(ls "$OUT"/prof_mx_mfma/*/*.db "$OUT"/prof_mx_mfma/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_batch64_fp8_mx_pmc_mfma.csv" --pmc > /dev/null
rm -rf "$OUT/prof_mx_mfma"; grep -E "mxu|mxk" "$OUT/r04_batch64_fp8_mx_pmc_mfma.csv" | cut -c1-200
cd "$REPO"
# two worker processes on the one GPU (pytest-xdist): most of the suite's 22 minutes are the CPU oracle's 32-layer passes; -rP = the tests' printed figures in the report
timeout 1300 python -m pytest tests -m gpu -q --tb=short -rP -p no:cacheprovider -n 2 --durations=8 > "$OUT/r04_pytest_gpu.log" 2>&1
echo "pytest exit $?"
{ grep -E "passed|failed" "$OUT/r04_pytest_gpu.log" | tail -1; grep -E "^FAILED|^ERROR" "$OUT/r04_pytest_gpu.log"
  sed -E 's/^[.sFE]+//' "$OUT/r04_pytest_gpu.log" | grep -vE "^$|passed|failed|[Ww]arning|^  |^=|^-|^_|amdgpu.ids|^\[gw|bringing up nodes" | cut -c1-2500 | head -400; } > "$OUT/r04_pytest_gpu_summary.txt"
head -4 "$OUT/r04_pytest_gpu_summary.txt" | cut -c1-300
