#!/usr/bin/env bash
# round 6, lease AB — long prompts: a sliced role as chunks of 512 rows (SK blocks + reduction each) or as ONE launch with the slices folded in registers
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06ab}
timeout 600 python tools/bench_prefill.py --rows 600 1100 1900 --sets "sk_sl_min_rows=100000;sk_sl_min_rows=512;sk_sl_min_rows=1024;prefill_sk=0;sk_sl_min_rows=100000;sk_sl_min_rows=512" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$OUT/${R}_long_prompts.txt"
