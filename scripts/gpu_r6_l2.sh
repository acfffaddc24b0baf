set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
for pr in 4 5 6 7; do for role in 1 2; do
echo "== probe $pr role $role"
DTK_BUS_PROBE=$pr STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=32 timeout 300 $SB "gemv_bus=$role" 2>&1 | sed -E 's/; logits hash.*//'
done; done | tee $OUT/r06l2_bus_probe.txt
