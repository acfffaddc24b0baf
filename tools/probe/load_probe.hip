// load_probe.hip — what one CU can pull through its vector-memory path, by footprint (L2-resident vs HBM-streaming), access
// pattern (1 KiB contiguous per wave-load vs 16 rows x 64 B vs 8 rows x 128 B), loads in flight per wave, and waves per CU.
// One block per CU (grid 256) unless stated; every wave issues U 16-byte loads per lane, waits, folds them into a checksum, repeats.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/load_probe tools/probe/load_probe.hip && tools/probe/load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U, int PATTERN, bool NT>
__global__ __launch_bounds__(1024) void k_load(const u32x4* __restrict__ buf, size_t n16, int iters, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  // every wave walks its own stream of 1 KiB units: unit index = ((it * gridDim.x + block) * waves + wave) * U + u  (mod footprint)
  const size_t units = n16 / 64;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t unit = (((size_t)it * gridDim.x + blockIdx.x) * waves + wave) * U + u;
      size_t idx;
      if (PATTERN == 0) idx = (unit % units) * 64 + lane;                       // 1 KiB contiguous
      else if (PATTERN == 1) {                                                   // 16 rows x 64 B: rows 8 KiB apart (a K = 4096 bf16 matrix)
        const size_t tile = unit % (units / 8);                                  // a tile = 16 rows x 512 B; 8 units per tile
        const size_t base = (tile / 1) * 16 * 512 ;                              // 16 rows of 8 KiB ... keep it simple: rows 512 chunks apart
        idx = ((base + (size_t)(lane & 15) * 512 + (unit % 8) * 4 + (lane >> 4)) % n16);
      } else {                                                                   // 8 rows x 128 B, rows 8 KiB apart
        const size_t tile = unit % (units / 4);
        idx = ((tile * 8 * 512 + (size_t)(lane >> 3) * 512 + (unit % 4) * 8 + (lane & 7)) % n16);
      }
      v[u] = NT ? __builtin_nontemporal_load(buf + idx) : buf[idx];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int U, int PATTERN, bool NT>
static double run(const u32x4* buf, size_t bytes, int blocks, int waves, unsigned* out) {
  const size_t n16 = bytes / 16;
  const size_t per_iter = (size_t)blocks * waves * U * 1024;
  int iters = (int)((size_t)(1ull << 31) / per_iter);       // ~2 GiB of loads per launch
  if (iters < 4) iters = 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_load<U, PATTERN, NT>), dim3(blocks), dim3(waves * 64), 0, 0, buf, n16, iters / 4 + 1, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_load<U, PATTERN, NT>), dim3(blocks), dim3(waves * 64), 0, 0, buf, n16, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return (double)per_iter * iters / (ms * 1e-3) / 1e9;      // GB/s, whole chip
}

int main() {
  const size_t big = (size_t)4 << 30;
  u32x4* buf; unsigned* out;
  hipMalloc(&buf, big); hipMalloc(&out, 64);
  hipMemset(buf, 1, big);
  printf("# GB/s whole chip (per CU = / 256 when 256 blocks); U = 16-byte loads per lane in flight per wave\n");
  const struct { const char* name; size_t bytes; } foot[] = {{"L2-resident 2 MiB", (size_t)2 << 20}, {"MALL-resident 96 MiB", (size_t)96 << 20}, {"HBM 4 GiB", big}};
  for (auto& f : foot) {
    for (int waves : {4, 8, 16}) {
      printf("%-22s %2d waves/CU  contiguous:  U=2 %7.0f  U=4 %7.0f  U=8 %7.0f  U=16 %7.0f   | nt U=8 %7.0f | 16x64B U=8 %7.0f | 8x128B U=8 %7.0f\n", f.name, waves,
             run<2, 0, false>(buf, f.bytes, 256, waves, out), run<4, 0, false>(buf, f.bytes, 256, waves, out), run<8, 0, false>(buf, f.bytes, 256, waves, out),
             run<16, 0, false>(buf, f.bytes, 256, waves, out), run<8, 0, true>(buf, f.bytes, 256, waves, out), run<8, 1, false>(buf, f.bytes, 256, waves, out),
             run<8, 2, false>(buf, f.bytes, 256, waves, out));
      fflush(stdout);
    }
  }
  // fewer CUs busy: what ONE CU can pull when the chip is otherwise idle, and 128 / 172 of 256
  for (int blocks : {1, 64, 128, 172}) {
    printf("HBM 4 GiB, %3d blocks x 8 waves, U=8: %7.0f GB/s = %6.1f per CU   | L2-resident: %7.0f = %6.1f per CU\n", blocks,
           run<8, 0, false>(buf, big, blocks, 8, out), run<8, 0, false>(buf, big, blocks, 8, out) / blocks,
           run<8, 0, false>(buf, (size_t)2 << 20, blocks, 8, out), run<8, 0, false>(buf, (size_t)2 << 20, blocks, 8, out) / blocks);
    fflush(stdout);
  }
  return 0;
}
