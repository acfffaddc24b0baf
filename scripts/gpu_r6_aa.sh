#!/usr/bin/env bash
# round 6, lease AA — the vision tower's fc2 as a sliced-K role (SK blocks + reduction up to 2048 rows, one launch with the slices folded in
# registers above): op tests, every ViT test, ms per image off / on
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06aa}
timeout 900 python -m pytest tests -m gpu -q -k "op_gemm or vit or selfsim or prefill_logits or reference_models_own or v2_prefill or headline or image" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head -30
for e in 0 1; do echo "== vit_sk=$e"; DTK_OPTIONS="vit_sk=$e" timeout 600 python tools/bench_vit.py 2>&1 | grep -i "auto\|batched"; done | tee "$OUT/${R}_vit.txt"
