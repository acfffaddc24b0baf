"""What do the fused prologues / epilogues of the decode GEMVs cost?  Each role at its default variant vs the same weights
through PRO_COPY + EPI_STORE and PRO_RMSNORM + EPI_STORE (dtk_bench_gemv variant flags 0x200 / 0x400)."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.model import load
model, _ = load("detikzify-ds-7b", synthetic=1234)
c = model.config
shapes = {0: ("qkv", 3 * c.hidden * c.hidden), 1: ("o_proj", c.hidden * c.hidden), 2: ("gate_up", 2 * c.ffn * c.hidden), 3: ("down", c.hidden * c.ffn)}
def t(role, variant):
    us = C.c_float()
    model._check(model.lib.dtk_bench_gemv(model._ctx, role, variant, 6, C.byref(us)), "bench")
    return us.value
for role, (name, n) in shapes.items():
    full = t(role, 0)
    u = 21 if role in (1, 3) else 20
    copy_store = t(role, 0x200 | u)
    norm_store = t(role, 0x400 | 20) if role in (0, 2) else float("nan")
    print(f"{name:8s} full {full:6.2f} us   copy+store {copy_store:6.2f}   rmsnorm+store {norm_store:6.2f}   ({2*n/1e6:.1f} MB, full {2*n/full/1e6:.2f} TB/s)")
