#!/usr/bin/env bash
# round 6, lease U — k_gemm_g3's W stage from the fragment-major weight copies: op tests, model-level tests, prefill times off / on, kernel trace
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06u}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "op_gemm or prefill or prefix or fork or headline or ds13b or full_size_incremental or reference_models_own or greedy_decode or v2 or fp8" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head -30
SETS="prefill_sk=0,gemm_wt=0;prefill_sk=1,gemm_wt=0;prefill_sk=1,gemm_wt=1;prefill_sk=0,gemm_wt=1;prefill_sk=1,gemm_wt=1"
timeout 600 python tools/bench_prefill.py --sets "$SETS" 2>&1 | grep -v Warning | tee "$OUT/${R}_prefill_ds7b.txt"
timeout 600 python tools/bench_prefill.py --model detikzify-cl-7b --weight-format fp8 --sets "$SETS" 2>&1 | grep -v Warning | tee "$OUT/${R}_prefill_cl7b_fp8.txt"
timeout 600 python tools/bench_prefill.py --model detikzify-ds-1.3b --sets "$SETS" 2>&1 | grep -v Warning | tee "$OUT/${R}_prefill_ds13b.txt"
B="--steps 2 --warmup 1 --new-tokens 32 --no-cpu-baseline --skip-batched --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-config5 --no-rank-shapes --probe-tokens 4"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...; rocprofv3 on this image sometimes dies with a segmentation fault before the program starts: three tries
  local name=$1; shift 1
  for try in 1 2 3; do
    rm -rf "$OUT/prof_$name"
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
    local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_$name.csv" > /dev/null && break
  done
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"
  grep -i "gemm\|sk_reduce\|rmsnorm_rows\|attention_mfma\|silu\|rope_scatter\|layernorm\|retile" "$OUT/${R}_$name.csv" | cut -c1-170
}
prof wt_kernel_stats python "$REPO/bench.py" $B
