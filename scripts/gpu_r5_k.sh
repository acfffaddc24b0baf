#!/usr/bin/env bash
# round 5, lease K — no GPU work, follow-up of lease J: 8 emulated ranks x 64 trees ran at 20-21 rollouts/s each where one rank alone
# reaches 44.8, with 11 of 256 cores busy.  Does giving every rank its own CPUs (what dist.pin_to_gpu_numa_node does in a real
# multi-GPU run) change that?  And the one-rank baselines of the N = 8 shapes.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
{
lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread|Core" | sed 's/  */ /g'
timeout 300 python tools/host_emulation.py --trees 8 --expansions 4 --new-tokens 256 --step-ms 2.3
timeout 300 python tools/host_emulation.py --procs 8 --pin --trees 8 --expansions 4 --new-tokens 256 --step-ms 2.3
timeout 300 python tools/host_emulation.py --procs 4 --trees 64 --expansions 2 --new-tokens 256 --step-ms 4.5
timeout 300 python tools/host_emulation.py --procs 8 --pin --trees 64 --expansions 2 --new-tokens 256 --step-ms 4.5
timeout 300 python tools/host_emulation.py --procs 8 --trees 64 --expansions 2 --new-tokens 256 --step-ms 4.5
} 2>&1 | grep -v amdgpu.ids | tee "$OUT/r05_host_emulation_gpu_box_pinned.txt"
