#!/usr/bin/env bash
# round 6, lease AC — after the SL threshold (768 rows): the prefill / long-context / op tests, long prompts again
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06ac}
timeout 1200 python -m pytest tests -m gpu -q -k "op_gemm or prefill or long_context or prefix or fork or headline or vit_batch" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head -30
timeout 600 python tools/bench_prefill.py --rows 16 600 1100 1900 --sets "prefill_sk=0;prefill_sk=1" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$OUT/${R}_long_prompts.txt"
