#!/usr/bin/env bash
# round 4, lease K (the kernel it measured lost and was removed in the next commit: `git show 985e8bf -- detikzify_amd/csrc/kernels_batch_decode.hip`;
# profiles/r04_attn_share_experiment.txt): group-shared attention (k_attn_share_b, option attn_share): the identity / tolerance test, then the 64-slot step with
# the option off / on — at the image-prefix context (32 steps) and over 256 steps of growing private contexts — ds-7b bf16 and cl-7b fp8.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -s --tb=short -k "group_shared_attention or shared_prefix_on_matrix_cores" 2>&1 | grep -vE "amdgpu.ids|^$" | cut -c1-600 | tail -25 | tee "$OUT/r04k_tests.txt"
step() { echo "-- $1 | $2 | $3 steps: $(DTK_OPTIONS="$2" timeout 300 python tools/bench_batch.py --batch 64 --fork --steps $3 $1 2>&1 | tail -1)"; }
FP8="--model detikzify-cl-7b --weight-format fp8"
{
  for o in "attn_share=0" "attn_share=1"; do step "" "$o" 32; step "" "$o" 256; step "$FP8" "$o" 32; step "$FP8" "$o" 256; done
  step "" "attn_share=1,tail_threads=128" 32; step "" "attn_share=1,tail_threads=128" 256
} | tee "$OUT/r04k_attn_share.txt"
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS=attn_share=1 timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_k" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16 > "$OUT/prof_k.log" 2>&1
db=$(ls "$OUT"/prof_k/*/*.db "$OUT"/prof_k/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_batch64_attn_share_kernel_stats.csv" > /dev/null
rm -rf "$OUT/prof_k"; grep -E "attn" "$OUT/r04_batch64_attn_share_kernel_stats.csv" | cut -c1-150
