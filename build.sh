#!/usr/bin/env bash
# Builds libdtk_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
SRC=detikzify_amd/csrc
OUT=detikzify_amd/lib
mkdir -p "$OUT" build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-variable"

if [ "$(cat build/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f build/*.o; mkdir -p build; echo "$FLAGS" > build/.flags; fi
pids=()
for f in kernels_decode kernels_decode_mv kernels_batch_decode kernels_batch_gemm kernels_batch_ks kernels_batch_mx kernels_batched kernels_sample_mb dtk_api; do
  if [ ! -f build/$f.o ] || [ $SRC/$f.hip -nt build/$f.o ] || [ $SRC/common.h -nt build/$f.o ] || [ $SRC/kernels.h -nt build/$f.o ] || [ $SRC/gemv_inl.h -nt build/$f.o ] || [ $SRC/batch_epi.h -nt build/$f.o ] || [ $SRC/mx_quant.h -nt build/$f.o ] || [ include/dtk.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $SRC/$f.hip -o build/$f.o &
    pids+=($!)
  fi
done
# the run loop of a batch (host code only; same flags so the exports stay include/dtk.h's)
if [ ! -f build/dtk_engine.o ] || [ $SRC/dtk_engine.cpp -nt build/dtk_engine.o ] || [ include/dtk.h -nt build/dtk_engine.o ]; then
  $HIPCC $FLAGS -x c++ -pthread -c $SRC/dtk_engine.cpp -o build/dtk_engine.o &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libdtk_hip.so build/kernels_decode.o build/kernels_decode_mv.o build/kernels_batch_decode.o build/kernels_batch_gemm.o build/kernels_batch_ks.o build/kernels_batch_mx.o build/kernels_batched.o build/kernels_sample_mb.o build/dtk_api.o build/dtk_engine.o -pthread
echo "built $OUT/libdtk_hip.so"
