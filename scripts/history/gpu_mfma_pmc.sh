#!/usr/bin/env bash
# MFMA-busy counters for the MFMA-bound rows (ViT + prefill GEMMs, flash attention, batched decode GEMV)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|GRBM_GUI_ACTIVE\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST_VALU" | sort -u > "$OUT/pmc_avail.txt"
cat "$OUT/pmc_avail.txt" | tr '\n' ' '; echo
CTRS=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE; do grep -qx "$c" "$OUT/pmc_avail.txt" && CTRS="$CTRS $c"; done
echo "using:$CTRS"
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/prof_mfma" -o pmc -- python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --probe-tokens 2 --batch 0 > "$OUT/prof_mfma.log" 2>&1
echo "rocprof mfma exit $?"
cd "$REPO"; python - <<'PY'
import sqlite3, collections
cur = sqlite3.connect("gpurun_out/prof_mfma/pmc_results.db").cursor()
rows = collections.defaultdict(dict)
for k, c, n, avg, dur in cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name"):
    rows[k][c] = avg; rows[k]["n"] = n; rows[k]["us"] = dur / 1e3
for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"])[:12]:
    print(k[:70], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
