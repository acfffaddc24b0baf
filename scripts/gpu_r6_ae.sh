#!/usr/bin/env bash
# round 6, lease AE — row strides of the prefill GEMMs' activations: natural (8 KiB at d = 4096) against + 128 B; switch test; kernel trace
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06ae}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "prefill_kernel_switches or prefill_logits or prefix_and_image" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head
timeout 600 python tools/bench_prefill.py --sets "prefill_ld_pad=0;prefill_ld_pad=64;prefill_ld_pad=0;prefill_ld_pad=64" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$OUT/${R}_ld_pad.txt"
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
  grep -i "gemm_g3\|sk_reduce\|sk_rope\|silu\|attention_mfma<128" "$OUT/${R}_$name.csv" | cut -c1-170
}
for pad in 0 64; do DTK_OPTIONS="prefill_ld_pad=$pad" prof pad${pad} python "$REPO/tools/bench_prefill.py" --sets "prefill_sk=1" --reps 3 --rows 16; done
