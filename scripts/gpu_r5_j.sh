#!/usr/bin/env bash
# round 5, lease J — no GPU work: the HOST side of 8 ranks on the GPU box's own host (256 threads), the shipped Python stack over the
# scripted device with the GPU emulated by sleeps of the measured step times (tools/host_emulation.py; round 3 could only run this on
# the 8-core build container).  Shapes: one rank with 64 trees; 8 ranks of BASELINE config 5 at N = 8 (8 trees per rank, 16-slot
# step 2.3 ms), of config 4 at N = 8 (2 trees per rank, multi-vector step 2.9 ms), and 8 ranks x 64 trees (the heaviest host load).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
{
nproc; python -c "import os; print('cpus in affinity mask:', len(os.sched_getaffinity(0)))"
timeout 300 python tools/host_emulation.py --trees 64 --expansions 2 --new-tokens 256 --step-ms 4.5
timeout 300 python tools/host_emulation.py --procs 8 --trees 8 --expansions 4 --new-tokens 256 --step-ms 2.3
timeout 300 python tools/host_emulation.py --procs 8 --trees 2 --expansions 2 --new-tokens 256 --step-ms 2.9
timeout 300 python tools/host_emulation.py --procs 8 --trees 64 --expansions 2 --new-tokens 256 --step-ms 4.5
timeout 300 python tools/host_emulation.py --procs 2 --trees 64 --expansions 2 --new-tokens 256 --step-ms 4.5
} 2>&1 | grep -v amdgpu.ids | tee "$OUT/r05_host_emulation_gpu_box.txt"
