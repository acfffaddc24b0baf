#!/usr/bin/env bash
# HBM read traffic (rocprofv3 --pmc FETCH_SIZE, own pass) of the batched-decode kernels at 32 slots
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_batch_pmc" -o pmc -- python "$REPO/tools/bench_batch.py" --batch 32 --steps 8 > "$OUT/prof_batch_pmc.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"; python tools/prof_summary.py "$OUT/prof_batch_pmc/pmc_results.db" "$OUT/batch32_pmc_fetch.csv" --pmc | grep -i "gemv_b\|attn_decode_b\|rmsnorm_b" | cut -c1-330
