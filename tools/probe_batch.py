"""Batched gate/up GEMV with and without its x-fragment loads, for 16 and 32 slots (dtk_bench_gemv role 5)."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.model import load
for slots in (16, 32):
    model, _ = load("detikzify-ds-7b", synthetic=1234, batch_slots=slots)
    for mode, name in ((0, "correct"), (1, "no x loads")):
        us = C.c_float()
        model._check(model.lib.dtk_bench_gemv(model._ctx, 5, mode, 4, C.byref(us)), "bench")
        print(f"slots {slots}: gate/up batched mode {mode} ({name}): {us.value:.2f} us  {2*model.config.ffn*model.config.hidden*2/us.value/1e3:.0f} GB/s")
    del model
