#!/usr/bin/env python
"""profiles/mfma_busy.json — what bench.py's `secondary_rooflines.mfma_busy` quotes: per GEMM / attention kernel of the ViT + prefill
stage, the matrix-core busy fraction from a rocprofv3 --pmc pass of its own (SQ_VALU_MFMA_BUSY_CYCLES over 4 x SQ_BUSY_CU_CYCLES: the
MFMA-busy cycles of a CU's four SIMDs against the cycles the CU had a wave resident), with the average duration beside it.

    python tools/make_mfma_busy_json.py profiles/r06_pmc_mfma.csv detikzify-ds-7b"""
import csv
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
pmc_csv, model = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(pmc_csv)))
by = {}
for r in rows:
    by.setdefault(r["kernel"], {})[r["counter"]] = r
out = {"model": model, "source": f"profiles/{os.path.basename(pmc_csv)} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES in its own pass)",
       "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)", "kernels": {}}
for k, c in by.items():
    if not any(t in k for t in ("k_gemm_g3", "k_gemm_mfma", "k_gemm_glds", "k_attention_mfma")):
        continue
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and float(c["SQ_BUSY_CU_CYCLES"]["avg_value"]) > 0:
        out["kernels"][k.replace("void ", "")] = {
            "mfma_busy": round(float(c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_value"]) / (4.0 * float(c["SQ_BUSY_CU_CYCLES"]["avg_value"])), 4),
            "dispatches": int(c["SQ_BUSY_CU_CYCLES"]["dispatches"]), "avg_us": round(float(c["SQ_BUSY_CU_CYCLES"]["avg_us"]), 2)}
if not out["kernels"]:
    sys.exit("no GEMM / attention kernel with both counters in " + pmc_csv)
(ROOT / "profiles" / "mfma_busy.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
