#!/usr/bin/env bash
# round 2, call A: parity of the new decode kernels / options, then the step-level tuning sweeps
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q --tb=short -x -p no:cacheprovider \
  -k "not headline and not full_size and not ds13b" > "$OUT/r2a_pytest.log" 2>&1
echo "pytest exit $?"; tail -5 "$OUT/r2a_pytest.log"
timeout 600 python tools/tune_decode.py --model detikzify-ds-7b --out "$OUT/tune_decode_ds7b.json" > "$OUT/tune_decode_ds7b.log" 2>&1
echo "tune ds7b exit $?"; grep -E "^->|default configuration|final configuration|role-level" "$OUT/tune_decode_ds7b.log"
timeout 400 python tools/tune_decode.py --model detikzify-ds-1.3b --out "$OUT/tune_decode_ds13b.json" > "$OUT/tune_decode_ds13b.log" 2>&1
echo "tune ds1.3b exit $?"; grep -E "^->|default configuration|final configuration|role-level" "$OUT/tune_decode_ds13b.log"
