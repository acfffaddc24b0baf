#!/usr/bin/env bash
# round 6, lease Q — the GPU box gives the container a CFS quota of 16 CPUs (cpu.max 1600000 100000) on a 256-thread host while torch sizes its
# intra-op pool at 128: config 5 with the default pool and with 16 / 8 threads, the cgroup's throttle counters around each run.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06q}
C5="--no-cpu-baseline --skip-batched --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-rank-shapes --steps 1 --warmup 0 --probe-tokens 4"
stat() { grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
for t in default 16 8 default 16; do
  echo "== OMP_NUM_THREADS=$t"; stat
  if [ "$t" = default ]; then timeout 900 python bench.py $C5 > "$OUT/${R}_c5_$t.json" 2>/dev/null; else OMP_NUM_THREADS=$t MKL_NUM_THREADS=$t timeout 900 python bench.py $C5 > "$OUT/${R}_c5_$t.json" 2>/dev/null; fi
  stat
  python - "$OUT/${R}_c5_$t.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c5 = d["mcts"]["config5"]
for k in ("fixed_length", "ragged", "fixed_length_fp8_matrix_cores_opt_in"):
    v = c5.get(k) or {}; e = v.get("engine") or {}
    print(" ", k, "rollouts/s", round(v.get("rollouts_per_sec") or 0, 2), "seconds", round(v.get("seconds") or 0, 2), {kk: e.get(kk) for kk in ("steps", "wait_s", "prefill_s", "drain_s", "idle_between_steps_s")})
PY
done 2>&1 | tee "$OUT/${R}_threads.txt"
