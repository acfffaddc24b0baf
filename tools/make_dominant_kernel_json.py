#!/usr/bin/env python
"""profiles/dominant_kernel.json — what bench.py's `roofline.traffic` quotes: the HBM bytes per launch of the dominant decode
kernel (RMSNorm + gate/up GEMV + SiLU*mul) from the rocprofv3 --pmc FETCH_SIZE summary, its rocprofv3 kernel-trace duration, and
the sha256 of the kernel source they were measured on.  bench.py reports the figure only while that hash matches.

    python tools/make_dominant_kernel_json.py profiles/r02_kernel_stats.csv profiles/r02_pmc_fetch.csv detikzify-ds-7b"""
import os
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
stats_csv, pmc_csv, model = sys.argv[1], sys.argv[2], sys.argv[3]
PREFIX = "void k_gemv<1, 3,"      # <PRO_RMSNORM, EPI_SWIGLU, ...>
out = {"model": model, "weight_format": "bf16", "kernel_prefix": PREFIX,
       "source": "detikzify_amd/csrc/kernels_decode.hip",
       "source_sha256": hashlib.sha256((ROOT / "detikzify_amd/csrc/kernels_decode.hip").read_bytes()).hexdigest()}
for row in csv.DictReader(open(stats_csv)):
    if row["kernel"].startswith(PREFIX):
        out["kernel"] = row["kernel"]
        out["rocprofv3_avg_us"] = float(row["avg_us"])
        out["rocprofv3_calls"] = int(row["calls"])
        break
for row in csv.DictReader(open(pmc_csv)):
    if row["counter"] == "FETCH_SIZE" and row["kernel"].startswith(PREFIX):
        out["hbm_read_bytes_per_launch"] = float(row["hbm_read_bytes_per_launch_x2"])
        out["traffic_source"] = f"profiles/{os.path.basename(pmc_csv)} (rocprofv3 --pmc FETCH_SIZE in its own pass, x 1024 B, x2 gfx950 correction: guides/MI355X_MICROARCH.md §HBM)"
        break
missing = [k for k in ("rocprofv3_avg_us", "hbm_read_bytes_per_launch") if k not in out]
if missing:
    sys.exit(f"kernel {PREFIX} not found in the summaries: {missing}")
(ROOT / "profiles" / "dominant_kernel.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
