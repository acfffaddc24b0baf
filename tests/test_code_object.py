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
    # a kernel is one YAML map item with alphabetically ordered keys (.agpr_count opens it, .name sits in the middle): collect the item,
    # file it under its name when the next one starts
    kernels, item = {}, None
    wanted = ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "agpr_count")

    def close(item):
        name = item.get("name") if item else None
        if name and name.startswith("_Z") and not name.endswith(".kd"):
            kernels[name] = {k: int(v) for k, v in item.items() if k in wanted}

    for line in notes.splitlines():
        m = re.match(r"\s*(-?)\s*\.(\w+):\s*(\S+)\s*$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if dash and key in ("agpr_count", "args"):
            close(item)
            item = {}
        if item is not None and key not in item:
            item[key] = val
    close(item)
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
        deep = name.endswith("ELb1ELi8EEv9GemvBArgs")          # fp8, register ring of 8 phases: one wave per SIMD, up to 512 unified registers
        assert meta["vgpr_count"] <= (512 if deep else 256), (name, meta)
    assert any(k.endswith("ELb1ELi4EEv9GemvBArgs") for k in br), "the fp8 ring-of-4 instantiations are the shipped fp8 kernels"
    assert not any("k_gemv_brILi" in k and re.search(r"k_gemv_brILi\dELi\dELi2E", k) for k in kernels), "the K = 2048 instantiations spill: not to be built"


def test_hand_load_checker_sees_a_register_read_before_its_wait():
    """tools/check_hand_loads.py on hand-made instruction streams: a ring register read (or copied to an AGPR, or stored) before the
    wait that retires its load is reported; the same stream with the wait in place, a loop whose back edge carries a pending load
    into a reader, and a wait that leaves exactly the newer loads outstanding are judged correctly."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import check_hand_loads as ch

    def run(lines):
        return ch.check_kernel("k", [(4 * i, mn, ops) for i, (mn, ops) in enumerate(lines)])[0]
    load = lambda r, a: ("global_load_dwordx4", f"v[{r}:{r + 3}], v[{a}:{a + 1}], off nt")
    assert run([load(0, 100), ("s_waitcnt", "vmcnt(0)"), ("v_mov_b32_e32", "v9, v1"), ("s_endpgm", "")]) == []
    assert len(run([load(0, 100), ("v_mov_b32_e32", "v9, v1"), ("s_waitcnt", "vmcnt(0)"), ("s_endpgm", "")])) == 1
    assert len(run([load(0, 100), ("v_accvgpr_write_b32", "a7, v3"), ("s_endpgm", "")])) == 1
    assert len(run([load(0, 100), ("scratch_store_dword", "off, v2, s0"), ("s_endpgm", "")])) == 1
    # two loads, vmcnt(1) retires the older one only
    two = [load(0, 100), load(4, 100), ("s_waitcnt", "vmcnt(1)")]
    assert run(two + [("v_add_u32_e32", "v9, v0, v9"), ("s_endpgm", "")]) == []
    assert len(run(two + [("v_add_u32_e32", "v9, v4, v9"), ("s_endpgm", "")])) == 1
    # a compiler-issued load (no nt) takes a counter slot but protects nothing: an extra operation only makes a hand count wait
    # for more; a count that is one too HIGH retires nothing
    assert run([load(0, 100), ("global_load_dword", "v20, v[100:101], off"), ("s_waitcnt", "vmcnt(1)"), ("v_mov_b32_e32", "v9, v0"), ("s_endpgm", "")]) == []
    assert len(run([load(0, 100), load(4, 100), ("s_waitcnt", "vmcnt(2)"), ("v_mov_b32_e32", "v9, v0"), ("s_endpgm", "")])) == 1
    # loop: the load at the bottom is still pending when the back edge returns to the reader at the top
    loop = [("v_mov_b32_e32", "v9, v0"), load(0, 100), ("s_cbranch_scc1", "65533"), ("s_waitcnt", "vmcnt(0)"), ("s_endpgm", "")]
    assert {v[0] for v in run(loop)} == {0, 4}       # the reader, and the re-issue into a register whose previous load is still in flight


@pytest.mark.skipif(not (LLVM / "clang-offload-bundler").exists(), reason="needs the ROCm LLVM tools")
def test_no_instruction_touches_a_hand_loaded_register_before_its_wait(tmp_path):
    """k_gemv_br's correctness rests on where the compiler puts its own moves relative to the hand-issued
    loads and waits.  This walks the shipped code object: every k_gemv_br instantiation, every path, 0 violations."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import check_hand_loads as ch
    obj = ROOT / "build" / "kernels_batch_gemm.o"
    if not obj.exists():
        subprocess.run(["bash", str(ROOT / "build.sh")], check=True, capture_output=True, cwd=str(ROOT))
    kernels = ch.kernels_of(ch.disassemble(obj, tmp_path), "k_gemv_br")
    assert len(kernels) >= 9
    for name, insts in kernels.items():
        violations, stats = ch.check_kernel(name, insts)
        assert stats["hand_loads"] >= 32 and stats["vmcnt_waits"] >= 32, (name, stats)
        assert not violations, (name, violations[:4])


@pytest.mark.skipif(not (LLVM / "clang-offload-bundler").exists(), reason="needs the ROCm LLVM tools")
def test_bus_kernels_wait_for_every_hand_issued_load_and_do_not_spill(tmp_path):
    """k_gemv_bus (round 6, csrc/kernels_batch_ks.hip) issues EVERY operand load by hand — the 64 slots' x fragments and the weight
    tiles, a ring of 2-4 k-step groups per wave — and counts vmcnt itself.  The shipped code object, instantiation by instantiation
    (qkv / gate-up x 1-3 units x bf16 / fp8 x K = 4096 / 2048): no instruction touches a register of a hand-issued load before a wait
    has retired it, on any path, and nothing spills (a spilled ring register would be stored before its load has landed; the
    launcher also refuses an instantiation with scratch at run time)."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import check_hand_loads as ch
    obj = ROOT / "build" / "kernels_batch_ks.o"
    if not obj.exists():
        subprocess.run(["bash", str(ROOT / "build.sh")], check=True, capture_output=True, cwd=str(ROOT))
    meta = {k: v for k, v in _kernel_metadata(obj, tmp_path).items() if "k_gemv_bus" in k}
    assert len(meta) == 16, sorted(meta)
    for name, m in meta.items():
        assert m.get("private_segment_fixed_size") == 0 and m.get("vgpr_spill_count") == 0 and m.get("sgpr_spill_count") == 0, (name, m)
        assert m.get("vgpr_count", 0) <= 256, (name, m)          # two waves per SIMD: 8 waves = the 8 K slices on one CU
    kernels = ch.kernels_of(ch.disassemble(obj, tmp_path), "k_gemv_bus")
    assert len(kernels) == 16
    for name, insts in kernels.items():
        violations, stats = ch.check_kernel(name, insts)
        assert stats["hand_loads"] >= 48 and stats["vmcnt_waits"] >= 9 and stats["points_given_up"] == 0, (name, stats)
        assert not violations, (name, violations[:4])


@pytest.mark.skipif(not (LLVM / "clang-offload-bundler").exists(), reason="needs the ROCm LLVM tools")
def test_fp8_matrix_core_kernels_issue_the_scaled_fp8_mfma_and_nothing_spills(tmp_path):
    """BASELINE config 5 names "CDNA4 fp8 MFMA": the shipped code object of csrc/kernels_batch_mx.hip, instantiation by instantiation —
    every k_gemv_mxu / k_gemv_mxk issues v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8 on the matrix cores, E8M0 block scales) and not a
    single fp8 -> bf16 widening or bf16 MFMA (what the round-1..3 fp8 kernels spent their instruction stream on), no instantiation has
    scratch or spilled registers (the five-wave / four-tile RoPE instantiation that would is not built), and a unit kernel's block
    stays within one wave per SIMD's worth of registers where it has four waves."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import check_hand_loads as ch
    obj = ROOT / "build" / "kernels_batch_mx.o"
    if not obj.exists():
        subprocess.run(["bash", str(ROOT / "build.sh")], check=True, capture_output=True, cwd=str(ROOT))
    meta = {k: v for k, v in _kernel_metadata(obj, tmp_path).items() if "k_gemv_mx" in k}
    assert len(meta) >= 60, len(meta)                      # 3 roles x waves x tiles of the unit kernel + the K-slice kernel's shapes
    for name, m in meta.items():
        assert m.get("private_segment_fixed_size") == 0 and m.get("vgpr_spill_count") == 0 and m.get("sgpr_spill_count") == 0, (name, m)
    assert not any(re.search(r"k_gemv_mxuILi2ELi4ELi4E", k) for k in meta), "q/k/v with five waves at four tiles spills: not to be built"
    dis = ch.disassemble(obj, tmp_path)
    for prefix in ("k_gemv_mxu", "k_gemv_mxk"):
        kernels = ch.kernels_of(dis, prefix)
        assert kernels
        for name, insts in kernels.items():
            mn = [i[1] for i in insts]
            scaled = sum(1 for x in mn if x.startswith("v_mfma_scale_f32_16x16x128_f8f6f4"))
            other_mfma = sum(1 for x in mn if x.startswith("v_mfma") and not x.startswith("v_mfma_scale_f32_16x16x128_f8f6f4"))
            widen = sum(1 for x in mn if x.startswith("v_cvt_scalef32_pk_bf16_fp8") or x.startswith("v_cvt_pk_f32_fp8"))
            assert scaled >= 4 and other_mfma == 0 and widen == 0, (name, scaled, other_mfma, widen)
