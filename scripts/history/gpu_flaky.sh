#!/usr/bin/env bash
set -uo pipefail
export PYTHONUNBUFFERED=1
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "not full_size and not ds13b and not fp8" 2>&1 | grep -E "^FAILED|^E  |passed|failed|Error" | head -12
done
