#!/usr/bin/env bash
# round 4, lease D: why does the ViT take 2.2 ms per image whatever GEMM kernel runs?  Kernel traces of the 8-image pass with
# k_gemm_glds (default) and k_gemm_g3, and k_gemm_g3 with its fills / its MFMAs left out.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- python "$REPO/tools/bench_vit.py" --only 8 > "$OUT/prof_$name.log" 2>&1
  local db; db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_${name}_kernel_stats.csv" > /dev/null
  rm -rf "$OUT/prof_$name"
  echo "== $name"; grep -E "gemm|attention|layernorm|im2col" "$OUT/r04_${name}_kernel_stats.csv" | head -8 | cut -c1-150
}
DTK_OPTIONS="gemm_impl=3" prof vit8_glds
DTK_OPTIONS="gemm_impl=4" prof vit8_g3
DTK_OPTIONS="gemm_impl=4" DTK_G3_PROBE=1 prof vit8_g3_nofill
DTK_OPTIONS="gemm_impl=4" DTK_G3_PROBE=2 prof vit8_g3_nomfma
DTK_OPTIONS="gemm_impl=4" DTK_G3_PROBE=3 prof vit8_g3_neither
