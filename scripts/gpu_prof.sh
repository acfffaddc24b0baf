#!/usr/bin/env bash
# GPU session: full parity suite, variant sweep, default bench (+cpu baseline), rocprofv3 kernel
# trace and a separate PMC pass (FETCH_SIZE) for the roofline `traffic` figure.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
BM=${BENCH_MODEL:-detikzify-ds-7b}
echo "== pytest (full)"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^FAILED|rel_l2.*ds-" "$OUT/pytest_gpu.log"
echo "== tune"
timeout 600 python tools/tune_gemv.py --model $BM --out "$OUT/tune_gemv.json" > "$OUT/tune_gemv.log" 2>&1; tail -60 "$OUT/tune_gemv.log"
echo "== combine A/B"
for mode in kernel inkernel consumer; do
  DTK_ATTN_COMBINE=$mode timeout 300 python bench.py --model $BM --steps 1 --warmup 1 --no-cpu-baseline --probe-tokens 8 > "$OUT/bench_$mode.log" 2> "$OUT/bench_$mode.err"
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$mode.log").read().strip().splitlines()[-1])
    print("$mode", "tok/s", round(d["value"],1), "decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1))
except Exception as e: print("$mode parse fail", e)
PY
done
echo "== bench (default)"
timeout 900 python bench.py --model $BM > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench exit $?"; tail -c 2500 "$OUT/bench.log"
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$REPO/bench.py" --model $BM --steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --probe-tokens 4 > "$OUT/prof_bench.log" 2>&1
echo "rocprof exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_pmc" -o pmc -- python "$REPO/bench.py" --model $BM --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --probe-tokens 2 > "$OUT/prof_pmc.log" 2>&1
echo "rocprof pmc exit $?"
cd "$REPO"; ls -la "$OUT/prof" "$OUT/prof_pmc" 2>/dev/null | head
