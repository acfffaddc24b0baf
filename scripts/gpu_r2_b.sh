#!/usr/bin/env bash
# round 2, call B: parity of the batched-path kernels (prefix on MFMA, tail kernel, wide tiles), decode re-sweep, batched sweep
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q --tb=short -p no:cacheprovider \
  -k "not headline and not full_size and not ds13b" > "$OUT/r2b_pytest.log" 2>&1
echo "pytest exit $?"; tail -8 "$OUT/r2b_pytest.log"
timeout 500 python tools/tune_decode.py --model detikzify-ds-7b --quick --out "$OUT/tune_decode_ds7b_b.json" > "$OUT/tune_decode_ds7b_b.log" 2>&1
echo "tune ds7b exit $?"; grep -E "^->|default configuration|final configuration" "$OUT/tune_decode_ds7b_b.log"
timeout 300 python tools/tune_decode.py --model detikzify-ds-1.3b --quick --out "$OUT/tune_decode_ds13b_b.json" > "$OUT/tune_decode_ds13b_b.log" 2>&1
echo "tune ds1.3b exit $?"; grep -E "^->|default configuration|final configuration" "$OUT/tune_decode_ds13b_b.log"
timeout 600 python tools/tune_batch.py --model detikzify-ds-7b --batch 64 --out "$OUT/tune_batch_ds7b.json" > "$OUT/tune_batch_ds7b.log" 2>&1
echo "tune batch exit $?"; grep -E "^---|ms/step" "$OUT/tune_batch_ds7b.log"
