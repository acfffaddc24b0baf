#!/usr/bin/env bash
# round 3, lease C: the persistent-layer probe after the LDS-DMA offset fix (correctness vs the launch chain + where its time goes),
# per-kernel profile of the batched ViT under both GEMMs.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
( cd tools/probe && timeout 120 ./engine2_probe 8 20 ) > "$OUT/r3c_engine2_probe.txt" 2>&1
echo "engine probe exit $?"; cat "$OUT/r3c_engine2_probe.txt"
cd /tmp && export TMPDIR=/tmp
for impl in 0 2; do
  DTK_OPTIONS="gemm_impl=$impl" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_vit$impl" -o trace -- python "$REPO/tools/bench_vit.py" --only 8 > "$OUT/prof_vit$impl.log" 2>&1
  db=$(ls "$OUT"/prof_vit$impl/*/*.db "$OUT"/prof_vit$impl/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r3c_vit8_gemm${impl}_kernel_stats.csv" >/dev/null && grep -v "fill_synth\|rocclr" "$OUT/r3c_vit8_gemm${impl}_kernel_stats.csv" | head -12
  rm -rf "$OUT/prof_vit$impl"
done
