#!/usr/bin/env bash
# Round-1 evidence run: full parity suite, bench (with cpu baseline), A/B of the attention variant,
# rocprofv3 kernel trace + PMC pass, batched-decode profile.  Everything lands in gpurun_out/.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
BM=${BENCH_MODEL:-detikzify-ds-7b}
timeout 1800 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^FAILED" "$OUT/pytest_gpu.log"
for fm in 0 1024; do
  DTK_ATTN_FULL_MAX=$fm timeout 300 python bench.py --model $BM --steps 1 --warmup 1 --no-cpu-baseline --probe-tokens 8 --batch 0 > "$OUT/bench_fm$fm.log" 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/bench_fm$fm.log").read().strip().splitlines()[-1]); print("attn_full_max=$fm decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
done
timeout 1200 python bench.py --model $BM > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench exit $?"; tail -c 3000 "$OUT/bench.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$REPO/bench.py" --model $BM --steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --probe-tokens 4 --batch 0 > "$OUT/prof_bench.log" 2>&1
echo "rocprof exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_pmc" -o pmc -- python "$REPO/bench.py" --model $BM --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --probe-tokens 2 --batch 0 > "$OUT/prof_pmc.log" 2>&1
echo "rocprof pmc exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_batch" -o trace -- python "$REPO/tools/bench_batch.py" --batch 16 --steps 32 > "$OUT/prof_batch.log" 2>&1
tail -1 "$OUT/prof_batch.log"
cd "$REPO"
