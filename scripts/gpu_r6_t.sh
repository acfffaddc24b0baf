#!/usr/bin/env bash
# round 6, lease T — sliced-K prefill GEMMs, second pass: 1024-thread reduce, slice caps and tiles; the op test's failure in full; the
# model-level tests
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06t}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sliced_k" 2>&1 | grep -v "^PASSED" | tail -60 | cut -c1-250 > "$OUT/${R}_pytest_op.txt"; grep -n "^E \|passed\|failed" "$OUT/${R}_pytest_op.txt" | head -30
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "prefill or prefix or fork or headline or ds13b or full_size_incremental or reference_models_own or greedy_decode" 2>&1 | tail -30 | cut -c1-250 > "$OUT/${R}_pytest_model.txt"; grep -n "^E \|passed\|failed\|^FAILED" "$OUT/${R}_pytest_model.txt" | head -30
timeout 600 python tools/bench_prefill.py 2>&1 | grep -v Warning | tee "$OUT/${R}_prefill_ds7b.txt"
timeout 600 python tools/bench_prefill.py --model detikzify-ds-1.3b --sets "prefill_sk=0;prefill_sk=1;prefill_sk=4" 2>&1 | grep -v Warning | tee "$OUT/${R}_prefill_ds13b.txt"
