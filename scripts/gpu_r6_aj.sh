#!/usr/bin/env bash
# round 6, lease AJ — SiLU*mul as the epilogue of the prefill's gate/up GEMM (pair-interleaved fragment-major weights): switch test, prefill
# tests, prefill times off / on, kernel trace
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06aj}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "prefill or prefix or fork or headline or fp8 or v2_prefill or full_size_incremental" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head -20
timeout 600 python tools/bench_prefill.py --sets "swiglu_fused=0;swiglu_fused=1;swiglu_fused=0;swiglu_fused=1" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$OUT/${R}_swiglu.txt"
timeout 600 python tools/bench_prefill.py --model detikzify-cl-7b --weight-format fp8 --sets "swiglu_fused=0;swiglu_fused=1" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a "$OUT/${R}_swiglu.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_v"; timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_v" -o trace -- python "$REPO/tools/bench_prefill.py" --sets "prefill_sk=1" --reps 3 --rows 16 > "$OUT/prof_v.log" 2>&1
db=$(ls "$OUT"/prof_v/*/*.db "$OUT"/prof_v/*.db 2>/dev/null | head -1); [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_kernel_stats.csv" > /dev/null; rm -rf "$OUT/prof_v" "$OUT/prof_v.log"
grep -i "gemm_g3\|silu\|sk_reduce\|retile" "$OUT/${R}_kernel_stats.csv" | cut -c1-150
