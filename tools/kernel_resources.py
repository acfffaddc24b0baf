#!/usr/bin/env python
"""Registers, scratch and static LDS of every gfx950 kernel in build/*.o, read from the code objects' metadata (no GPU).
    python tools/kernel_resources.py [--all]      # default: kernels that spill, use AGPRs, or sit at the VGPR cap"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
        "group_segment_fixed_size", "max_flat_workgroup_size")


def kernels_of(obj: Path, tmp: Path):
    fat, co = tmp / "fat.bin", tmp / "dev.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True, capture_output=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                    f"--output={co}", "--unbundle"], check=True, capture_output=True)
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    # the metadata lists every kernel as one YAML map item ("  - .agpr_count: ..." opens it) with its keys in alphabetical order, so
    # .agpr_count and .group_segment_fixed_size come BEFORE .name: collect an item's keys first, file them under its name at the end
    out, item = {}, None

    def close(item):
        name = item.get("name") if item else None
        if name and name.startswith("_Z") and not name.endswith(".kd"):
            out[name] = {k: int(v) for k, v in item.items() if k in KEYS}

    for line in notes.splitlines():
        m = re.match(r"(\s*)(-?)\s*\.(\w+):\s*(\S+)\s*$", line)
        if not m:
            continue
        indent, dash, key, val = m.groups()
        if dash and key in ("agpr_count", "args"):      # first key of a kernel's map
            close(item)
            item = {}
        if item is not None and key not in item:
            item[key] = val
    close(item)
    return out


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or (str(LLVM / "llvm-cxxfilt") if (LLVM / "llvm-cxxfilt").exists() else None)
    if tool is None or not names:
        return list(names)
    r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else list(names)


def main():
    show_all = "--all" in sys.argv
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for obj in sorted((ROOT / "build").glob("*.o")):
            try:
                ks = kernels_of(obj, Path(d))
            except subprocess.CalledProcessError:
                continue
            names = list(ks)
            for name, pretty in zip(names, demangle(names)):
                m = ks[name]
                interesting = m.get("private_segment_fixed_size", 0) or m.get("agpr_count", 0) or m.get("vgpr_count", 0) >= 250
                if show_all or interesting:
                    rows.append((obj.stem, re.sub(r"\(.*\)$", "", pretty)[:100], m))
    print(f"{'file':22s} {'vgpr':>4s} {'agpr':>4s} {'scratch B':>9s} {'spilled v/s':>11s} {'LDS B':>6s}  kernel")
    for stem, pretty, m in sorted(rows, key=lambda r: (-r[2].get("private_segment_fixed_size", 0), -r[2].get("agpr_count", 0), r[0], r[1])):
        print(f"{stem:22s} {m.get('vgpr_count', 0):4d} {m.get('agpr_count', 0):4d} {m.get('private_segment_fixed_size', 0):9d} "
              f"{m.get('vgpr_spill_count', 0):5d}/{m.get('sgpr_spill_count', 0):<5d} {m.get('group_segment_fixed_size', 0):6d}  {pretty}")
    print(f"{len(rows)} kernels listed")


if __name__ == "__main__":
    main()
