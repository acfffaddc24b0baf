#!/usr/bin/env bash
# round 6, lease S — sliced-K prefill GEMMs (k_gemm_g3<.., SK> + k_sk_reduce): op-level family test, the prefill / fork / oracle tests they
# touch, prefill time off / on, kernel trace of a prefill
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06s}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "op_gemm or prefill or prefix or fork or headline or ds13b or full_size_incremental or reference_models_own or greedy_decode" -rA 2>&1 | grep -v "^PASSED\|^SKIPPED" | tail -40 | cut -c1-220 > "$OUT/${R}_pytest.txt"; tail -15 "$OUT/${R}_pytest.txt"
B="--steps 2 --warmup 1 --new-tokens 32 --no-cpu-baseline --skip-batched --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-config5 --no-rank-shapes --probe-tokens 4"
for sk in 0 1; do
  DTK_OPTIONS="prefill_sk=$sk" timeout 600 python bench.py $B > "$OUT/${R}_bench_sk$sk.json" 2>"$OUT/${R}_bench_sk$sk.err"
  python - "$OUT/${R}_bench_sk$sk.json" $sk <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("prefill_sk", sys.argv[2], "prefill_ms", d.get("prefill_ms"), "vit_ms", d.get("vit_ms"), "secondary", json.dumps(d.get("secondary_rooflines"))[:600])
PY
done 2>&1 | tee "$OUT/${R}_prefill.txt"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...; rocprofv3 on this image sometimes dies with a segmentation fault before the program starts: three tries
  local name=$1; shift 1
  for try in 1 2 3; do
    rm -rf "$OUT/prof_$name"
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
    local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_$name.csv" > /dev/null && break
  done
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"
  grep -i "gemm\|sk_reduce\|rmsnorm_rows\|attention_mfma\|silu\|rope_scatter\|layernorm" "$OUT/${R}_$name.csv" | cut -c1-170
}
for sk in 0 1; do DTK_OPTIONS="prefill_sk=$sk" prof sk${sk}_kernel_stats python "$REPO/bench.py" $B; done
