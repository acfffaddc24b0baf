#!/usr/bin/env bash
# round 6, lease W — k_gemm_g3's epilogue through LDS (row-contiguous 16-byte stores): op tests, prefill + batched ViT off / on, kernel trace
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06w}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "op_gemm or prefill or vit_features or vit_batch or headline or prefix or fork" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head -30
SETS="gemm_epi_direct=1;gemm_epi_direct=0;gemm_epi_direct=1;gemm_epi_direct=0"
timeout 600 python tools/bench_prefill.py --sets "$SETS" 2>&1 | grep -v Warning | tee "$OUT/${R}_prefill_ds7b.txt"
for e in 1 0; do echo "== gemm_epi_direct=$e"; DTK_OPTIONS="gemm_epi_direct=$e" timeout 600 python tools/bench_vit.py 2>&1 | grep -i "auto\|batched"; done | tee "$OUT/${R}_vit.txt"
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift 1
  for try in 1 2 3; do
    rm -rf "$OUT/prof_$name"
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
    local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_$name.csv" > /dev/null && break
  done
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"
  grep -i "gemm_g3\|sk_reduce" "$OUT/${R}_$name.csv" | cut -c1-170
}
prof epi_lds python "$REPO/tools/bench_prefill.py" --sets "prefill_sk=1" --reps 3 --rows 16
