"""What the compiler made of the kernels whose correctness depends on it (CPU: reads the gfx950 code object's metadata, no GPU).

k_gemv_br issues its weight loads by hand (inline asm) and counts `vmcnt` itself, because the compiler's own waits collapse its
register ring.  The price: a ring register the compiler SPILLS would be stored to scratch before its load has landed, and the load
would later land in a register that holds something else — the K = 2048 instantiations did exactly that (254 VGPRs + 68 bytes of
scratch: GPU fault) and are no longer built.  The launcher re-checks the loaded code object at run time (hipFuncGetAttributes); this
test is the same check at build time, on the object build.sh / __graft_entry__.build() leaves behind."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")


def _kernel_metadata(obj: Path, tmp: Path):
    fat, co = tmp / "fat.bin", tmp / "dev.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True, capture_output=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                    f"--output={co}", "--unbundle"], check=True, capture_output=True)
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "name" and val.startswith("_Z") and not val.endswith(".kd"):
            cur = kernels.setdefault(val, {})
        elif cur is not None and key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "agpr_count"):
            cur[key] = int(val)
    return kernels


@pytest.mark.skipif(not (LLVM / "clang-offload-bundler").exists() or shutil.which("hipcc") is None and not (LLVM / "llvm-readelf").exists(),
                    reason="needs the ROCm LLVM tools")
def test_kernels_with_hand_issued_loads_do_not_spill(tmp_path):
    obj = ROOT / "build" / "kernels_batch_gemm.o"
    if not obj.exists():
        subprocess.run(["bash", str(ROOT / "build.sh")], check=True, capture_output=True, cwd=str(ROOT))
    kernels = _kernel_metadata(obj, tmp_path)
    br = {k: v for k, v in kernels.items() if "k_gemv_br" in k}
    assert len(br) >= 9 and len(kernels) > 100, (len(br), len(kernels))         # 3 roles x 3 block shapes x {bf16, fp8}
    for name, meta in br.items():
        assert meta.get("private_segment_fixed_size") == 0 and meta.get("vgpr_spill_count") == 0, (name, meta)
        assert meta["vgpr_count"] <= 256, (name, meta)
    assert not any("k_gemv_brILi" in k and re.search(r"k_gemv_brILi\dELi\dELi2E", k) for k in kernels), "the K = 2048 instantiations spill: not to be built"
