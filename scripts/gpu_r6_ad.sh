#!/usr/bin/env bash
# round 6, lease AD — the driver's N > 1 bench command on the final source, N ranks sharing the one GPU over gloo (control flow only: the
# stripes, the gather, the max-over-ranks timing; throughput means nothing here)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06fin}
for n in 2 4; do
  DTK_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29610 + n)) bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${R}_bench_${n}ranks_gloo_1gpu.json" 2> "$OUT/${R}_bench_${n}ranks_gloo_1gpu.err"; echo "$n ranks: exit $?"
  python - "$OUT/${R}_bench_${n}ranks_gloo_1gpu.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
m = d.get("mcts") or {}
print("n_gpus", d["n_gpus"], "value", round(d["value"], 1), "scaling", d["scaling"], "ranks", len(d.get("ranks") or []), "| c4", ((m.get("config4") or {}).get("fixed_length") or {}).get("rollouts_per_sec"), "merged", ((m.get("config4") or {}).get("fixed_length") or {}).get("merged_on_rank0"),
      "| c5", ((m.get("config5") or {}).get("fixed_length") or {}).get("rollouts_per_sec"), "merged", ((m.get("config5") or {}).get("fixed_length") or {}).get("merged_on_rank0"))
PY
done
