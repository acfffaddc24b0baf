"""Rounding helpers shared by the oracle modules (test infrastructure, see oracle/__init__.py)."""
import torch


def rb(x: torch.Tensor, precision: str = "bf16") -> torch.Tensor:
    """Round an fp32 tensor through bf16 (RNE, = torch .to(bfloat16)); identity in fp32 mode."""
    if precision == "fp32":
        return x
    return x.to(torch.bfloat16).to(torch.float32)


def linear(x, w, b=None, precision="bf16"):
    """nn.Linear on bf16 tensors: fp32 accumulate (+ bias), one rounding of the output.
    Weights stored as torch.bfloat16 take the native bf16 GEMM (fp32 accumulate, bf16 output —
    the same rounding point, half the memory: what HF does on a CPU in bf16)."""
    if w.dtype == torch.bfloat16:
        y = torch.nn.functional.linear(x.to(torch.bfloat16), w, None if b is None else b.to(torch.bfloat16))
        return y.float()
    y = x @ w.t()
    if b is not None:
        y = y + b
    return rb(y, precision)


def bits_to_f32(bits):
    """uint16 numpy array of bf16 bit patterns -> fp32 torch tensor."""
    import numpy as np
    u = bits.astype(np.uint32) << 16
    return torch.from_numpy(u.view(np.float32).copy())


def f32_to_bits(x: torch.Tensor):
    """fp32 tensor -> bf16 bit patterns (uint16 numpy), RNE."""
    import numpy as np
    return x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16).copy()
