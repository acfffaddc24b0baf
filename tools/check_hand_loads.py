#!/usr/bin/env python
"""Static check of the kernels that issue global loads by hand and count `vmcnt` themselves (k_gemv_br): walks the gfx950
disassembly of every such kernel and verifies that no instruction touches a register of a hand-issued `global_load_dwordx4`
before an `s_waitcnt vmcnt(N)` has retired that load.  The compiler treats the asm's "=v" output as ready at once, so a `v_mov`
/ AGPR copy / spill of a ring register placed between the load and its wait would read garbage SILENTLY —
the run-time guard only sees scratch usage.  This reads what the compiler actually emitted.

Model: gfx9-family `vmcnt` counts every VMEM operation in issue order and retires them in order; `s_waitcnt vmcnt(N)` returns
when at most N are outstanding, i.e. all but the N newest have landed.  The walk is an abstract interpretation over the kernel's
control flow (both sides of every conditional branch, loop bodies entered with the state their back edge carries, each program
point revisited until no new state appears, bounded); a VMEM operation without a VGPR destination (stores, LDS-DMA, atomics
without return) occupies a queue slot but protects no register.

    python tools/check_hand_loads.py [build/kernels_batch_gemm.o] [--kernel k_gemv_br]      # exit status 1 on a violation"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
MAX_VISITS = 6          # distinct states per program point before the walk gives up on that point (reported)


def disassemble(obj: Path, tmp: Path) -> str:
    fat, co = tmp / "fat.bin", tmp / "dev.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True, capture_output=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                    f"--output={co}", "--unbundle"], check=True, capture_output=True)
    return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout


def kernels_of(text: str, needle: str):
    """{mangled name: [(address, mnemonic, operand text)]}"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), []) if needle in m.group(1) else None
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def regs(operands: str):
    """set of ('v' | 'a', index) named in an operand string"""
    out = set()
    for m in REG.finditer(operands):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def is_vmem(mn: str) -> bool:
    return mn.startswith(("global_", "buffer_", "flat_", "scratch_"))


def check_kernel(name: str, insts):
    all_loads = "k_gemv_bc" in name or "k_gemv_bus" in name     # (k_gemv_bus: the x fragments too are hand-issued, without `nt`)
    """-> (violations, stats)"""
    index = {addr: i for i, (addr, _, _) in enumerate(insts)}
    violations, hand_loads, waits = [], 0, 0
    seen = {}                       # pc -> set of states
    work = [(0, ())]                # state: tuple of frozensets (registers a pending VMEM op will write; empty = none), oldest first
    gave_up = set()
    while work:
        pc, q = work.pop()
        while 0 <= pc < len(insts):
            st = seen.setdefault(pc, set())
            if q in st:
                break
            if len(st) >= MAX_VISITS:
                gave_up.add(pc)
                break
            st.add(q)
            addr, mn, ops = insts[pc]
            touched = regs(ops)
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    waits += 1
                    n = int(m.group(1))
                    q = q[len(q) - n:] if n < len(q) else q
                    if n == 0:
                        q = ()
            elif is_vmem(mn):
                # only the hand-issued loads protect registers: `global_load_dwordx4 v[..], v[..], off nt` (br_load_nt's asm); what
                # the compiler issues itself (scales, RoPE operands, stores, the loader wave's LDS-DMA) it also waits for itself
                dst, srcs = frozenset(), touched
                # k_gemv_bc's hand-issued loads (bc_load: the x fragments, L2 hits) carry no `nt`: there EVERY 16-byte global load is
                # treated as protected — the compiler's own ones are waited for by the compiler, so they cannot raise a false alarm
                if mn == "global_load_dwordx4" and (re.search(r"\bnt\b", ops) or all_loads):
                    first = ops.split(",")[0]
                    dst = frozenset(regs(first))
                    hand_loads += 1
                    srcs = regs(ops[len(first):])
                pending = set().union(*q) if q else set()
                bad = (srcs | dst) & pending
                if bad:
                    violations.append((addr, mn, ops, sorted(bad)))
                q = q + (dst,)
                while len(q) > 63:      # the walk took a path the program cannot take (e.g. a loader loop that never waits):
                    if q[0]:            # harmless unless a protected load falls off the counter's range
                        violations.append((addr, mn, ops, "a hand-issued load is more than 63 VMEM operations old: vmcnt cannot cover it"))
                    q = q[1:]
            else:
                pending = set().union(*q) if q else set()
                bad = touched & pending
                if bad:
                    violations.append((addr, mn, ops, sorted(bad)))
            if mn == "s_endpgm":
                break
            if mn == "s_branch" or mn.startswith("s_cbranch"):
                m = re.match(r"^\s*(-?\d+)", ops.split(",")[-1].strip()) if ops else None
                target = None
                if m:
                    off = int(m.group(1))
                    if off >= 0x8000:
                        off -= 0x10000
                    target = index.get(addr + 4 + 4 * off)
                if target is not None:
                    if mn == "s_branch":
                        pc = target
                        continue
                    work.append((target, q))
            pc += 1
    uniq, out = set(), []
    for v in violations:                # a program point revisited with another state reports the same finding again
        key = (v[0], str(v[3]))
        if key not in uniq:
            uniq.add(key)
            out.append(v)
    violations = out
    return violations, dict(instructions=len(insts), hand_loads=hand_loads, vmcnt_waits=waits, points_given_up=len(gave_up))


def main(argv):
    obj = Path(next((a for a in argv[1:] if not a.startswith("--")), ROOT / "build" / "kernels_batch_gemm.o"))
    needle = argv[argv.index("--kernel") + 1] if "--kernel" in argv else "k_gemv_br"
    with tempfile.TemporaryDirectory() as d:
        text = disassemble(obj, Path(d))
    bad = 0
    for name, insts in sorted(kernels_of(text, needle).items()):
        v, stats = check_kernel(name, insts)
        print(f"{name}: {stats}" + (f"  {len(v)} VIOLATIONS" if v else "  ok"))
        for item in v[:8]:
            print("   ", item)
        bad += len(v)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
