#!/usr/bin/env bash
# round 6, lease AK — a sliced gate/up role (d = 2048 models) reduces straight into SiLU*mul: switch test, ds-1.3b prefill off / on, its tests
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06ak}
timeout 900 python -m pytest tests -m gpu -q -k "prefill_kernel_switches or ds13b or prefill_logits or long_context_properties" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest.txt" | head -20
timeout 600 python tools/bench_prefill.py --model detikzify-ds-1.3b --sets "swiglu_fused=0;swiglu_fused=1;swiglu_fused=0;swiglu_fused=1" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$OUT/${R}_swiglu_ds13b.txt"
