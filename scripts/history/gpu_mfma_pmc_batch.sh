#!/usr/bin/env bash
# MFMA-busy counters (own --pmc pass) for the batched-decode step at 32 slots
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d "$OUT/prof_mfma_b" -o pmc -- python "$REPO/tools/bench_batch.py" --batch 32 --steps 6 --fork > "$OUT/prof_mfma_b.log" 2>&1
echo "rocprof exit $?"
