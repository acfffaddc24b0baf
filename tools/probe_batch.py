"""Differential timing of the batched gate/up GEMV (dtk_bench_gemv role 5): what the kernel costs without its x-fragment
loads (mode bit 1), without its MFMAs (bit 2), without its cross-wave reduction + epilogue (bit 4), at 16 / 32 / 64 slots.
(Some of the variants timed here — k_gemm_b, k_gemv_bk, k_gemm_dma, the experiment modes of k_gemv_b — are only built with
DTK_EXPERIMENTS=1 ./build.sh; a default build reports "built without DTK_EXPERIMENTS" for them.)"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.model import load
NAMES = {0: "full kernel", 1: "no x loads", 2: "no MFMA", 4: "no reduction / epilogue", 3: "no x, no MFMA", 6: "no MFMA, no epilogue",
         5: "no x, no epilogue", 7: "weights stream only", 8: "<= 128 VGPRs: 2 blocks per CU", 16: "2 stages of 2 k-steps in flight",
         17: "2 stages of 4 k-steps in flight", 24: "2 stages of 2, <= 128 VGPRs", 32: "x traffic / 4 (one fragment feeds all tiles)",
         40: "x traffic / 4, <= 128 VGPRs", 48: "x traffic / 4, 2 stages of 2", 56: "x traffic / 4, 2 stages of 2, <= 128 VGPRs",
         64: "non-temporal x loads", 128: "k_gemv_bx: x once per CU through LDS phases (auto)", 129: "k_gemv_bx, 2 units per block",
         130: "k_gemv_bx, 3 units per block", 131: "k_gemv_bx, 4 units per block"}
for slots in (64,):
    model, _ = load("detikzify-ds-7b", synthetic=1234, batch_slots=slots)
    for mode in ((0, 128, 129, 130, 131) if '--bx' in sys.argv else (0, 1, 7, 8, 16, 17, 24, 32, 40, 48, 56, 64)):
        us = C.c_float()
        best = 1e9
        for _ in range(2):
            model._check(model.lib.dtk_bench_gemv(model._ctx, 5, mode, 4, C.byref(us)), "bench")
            best = min(best, us.value)
        print(f"slots {slots}: gate/up mode {mode} ({NAMES[mode]:46s}): {best:6.2f} us  {2*model.config.ffn*model.config.hidden*2/best/1e3:5.0f} GB/s", flush=True)
    del model

# the LDS-DMA kernel (role 6): shape 1 = 2 K splits x 4 row groups, shape 2 = 4 x 2; mode bits 1 no x DMA, 2 no MFMA, 4 no barriers, 8 no weight loads
model, _ = load("detikzify-ds-7b", synthetic=1234, batch_slots=64) if "--no-lds" not in sys.argv else (None, None)
N6 = {0: "full kernel", 1: "no x DMA", 2: "no MFMA", 4: "no barriers / waits", 5: "no DMA, no barriers", 8: "no weight loads", 9: "no weights, no DMA",
      13: "MFMA + LDS reads only", 7: "weights only (no DMA, MFMA, barriers)"}
for shape in (() if '--no-lds' in sys.argv else (1, 2)):
    for mode in (0, 1, 2, 4, 5, 7, 8, 9, 13):
        us = C.c_float()
        best = 1e9
        for _ in range(2):
            model._check(model.lib.dtk_bench_gemv(model._ctx, 6, shape * 16 + mode, 4, C.byref(us)), "bench")
            best = min(best, us.value)
        print(f"slots 64: k_gemm_b shape {shape} mode {mode:2d} ({N6[mode]:38s}): {best:6.2f} us  {2*model.config.ffn*model.config.hidden*2/best/1e3:5.0f} GB/s", flush=True)
