#!/usr/bin/env bash
# Round evidence run: full parity suite, benches (ds-7b bf16 default incl. cpu baseline + batched phase,
# ds-1.3b, cl-7b fp8), 2-rank control-flow run (gloo, one GPU), rocprofv3 kernel traces + PMC pass.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^FAILED" "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 120 tools/probe/engine_probe 8 20 > "$OUT/engine_probe.txt" 2>&1; tail -3 "$OUT/engine_probe.txt"
timeout 1200 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench exit $?"; tail -c 3200 "$OUT/bench.log"
timeout 600 python bench.py --model detikzify-ds-1.3b --no-cpu-baseline --batch 0 > "$OUT/bench_13b.log" 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_13b.log").read().strip().splitlines()[-1]); print("ds-1.3b: tok/s", round(d["value"],1), "decode", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
timeout 600 python bench.py --model detikzify-cl-7b --weight-format fp8 --no-cpu-baseline --batch 0 > "$OUT/bench_cl7b_fp8.log" 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_cl7b_fp8.log").read().strip().splitlines()[-1]); print("cl-7b fp8: tok/s", round(d["value"],1), "decode", round(d["decode_tokens_per_sec_per_gpu"],1), "frac", round(d["decode_step"]["frac_of_hbm_peak"],3))
PY
timeout 600 python bench.py --model detikzify-cl-7b --weight-format fp8 --no-cpu-baseline --steps 1 --batch 64 > "$OUT/bench_cl7b_fp8_b64.log" 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_cl7b_fp8_b64.log").read().strip().splitlines()[-1]); b=d["batched_rollouts"]; print("cl-7b fp8 B=64: rollouts/s", round(b["rollouts_per_sec"],2), "tok/s", round(b["tokens_per_sec"]))
PY
timeout 600 python bench.py --model detikzify-v2-8b --no-cpu-baseline --steps 2 --batch 64 > "$OUT/bench_v2_8b.log" 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_v2_8b.log").read().strip().splitlines()[-1]); b=d["batched_rollouts"]; print("v2-8b: tok/s", round(d["value"],1), "decode", round(d["decode_tokens_per_sec_per_gpu"],1), "frac", round(d["decode_step"]["frac_of_hbm_peak"],3), "| B=64 rollouts/s", round(b["rollouts_per_sec"],2))
PY
timeout 600 python bench.py --sample --no-cpu-baseline --batch 0 --steps 2 > "$OUT/bench_sample.log" 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_sample.log").read().strip().splitlines()[-1]); print("ds-7b sampling: tok/s", round(d["value"],1), "decode", round(d["decode_tokens_per_sec_per_gpu"],1))
PY
DTK_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --model detikzify-ds-1.3b --steps 2 --warmup 1 --no-cpu-baseline --batch 0 > "$OUT/bench_2rank_gloo.log" 2> "$OUT/bench_2rank_gloo.err"; echo "2-rank exit $?"; tail -c 600 "$OUT/bench_2rank_gloo.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --probe-tokens 4 --batch 0 > "$OUT/prof_bench.log" 2>&1
echo "rocprof exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_pmc" -o pmc -- python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --probe-tokens 2 --batch 0 > "$OUT/prof_pmc.log" 2>&1
echo "rocprof pmc exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_batch" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_batch.log" 2>&1
grep "ms/step" "$OUT/prof_batch.log"
cd "$REPO"
