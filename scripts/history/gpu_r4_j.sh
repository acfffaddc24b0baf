#!/usr/bin/env bash
# round 4, lease J: the fp8 matrix-core tests after their restatement (envelope), then two knob sweeps on the 64-slot step that cost
# no code: compute waves per block of k_gemv_mxu per role, and the shared-prefix attention on the matrix cores (k_attn_prefix_b:
# "no gain" in round 2, when attention was 15 % of a step — it is 25 % of the fp8 matrix-core step now).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity_mx.py -q -p no:cacheprovider -s --tb=short 2>&1 | grep -vE "amdgpu.ids|^$" | cut -c1-1200 | tail -60 | tee "$OUT/r04j_mx_tests.txt"
step() { echo "-- $1 | $2: $(DTK_OPTIONS="$2" timeout 300 python tools/bench_batch.py --batch 64 --fork --steps 32 $1 2>&1 | tail -1)"; }
FP8="--model detikzify-cl-7b --weight-format fp8"
{
  for o in "mx_nc_qkv=1" "mx_nc_qkv=3" "mx_nc_gu=2" "mx_nc_gu=4" "mx_nc_lm_head=3"; do step "$FP8" "$o"; done
  for o in "prefix_mfma=0" "prefix_mfma=1,pfx_splits=2" "prefix_mfma=1,pfx_splits=4"; do step "$FP8" "$o"; step "" "$o"; done
  for o in "tail_threads=128" "tail_threads=512"; do step "$FP8" "$o"; done
} | tee "$OUT/r04j_sweeps.txt"
