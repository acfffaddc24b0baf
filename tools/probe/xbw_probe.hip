// xbw_probe.hip — how fast can EVERY CU at once pull the 64 slots' x fragments (L2 hits) into LDS, alone and next to an HBM weight
// stream?  Decides the shape of the batched unit kernels (round 6): if 512 KiB of x per CU costs ~5 us, an output-stationary kernel
// (full K per wave, x streamed once per CU) is enough; if it costs ~13 us (the 40 GB/s per CU the round-2..6 kernels saw), only a K
// split (x slice stationary in LDS, fp32 partials) removes it.
//   hipcc --offload-arch=gfx950 -O3 -o xbw_probe xbw_probe.hip && ./xbw_probe
// One block per CU, WAVES waves.  Per chunk of 64 KiB: every wave loads its share (64 KiB / WAVES) with 16-byte loads into registers,
// parks it in LDS (ds_write_b128), raw barrier, and (optionally) reads all of it back as 4 fragments per k-step (the MFMA operand
// reads).  The weight stream: every wave reads `wbytes / (blocks * WAVES)` of its own with DEPTH non-temporal loads in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int WAVES, bool XLOAD, bool WLOAD, bool LDSREAD>
__global__ __launch_bounds__(WAVES * 64) void k_probe(const unsigned char* X, int xchunks, const unsigned char* W, size_t wbytes_per_wave, unsigned* sink) {
  constexpr int FPW = 64 / WAVES;                   // 1 KiB fragments per wave and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x 64 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  const unsigned char* w = W + ((size_t)(blockIdx.x * WAVES + wave)) * wbytes_per_wave + lane * 16;
  const int wtiles = max(1, (int)(wbytes_per_wave >> 10));
  const int wper = xchunks > 0 ? (wtiles + xchunks - 1) / xchunks : wtiles;    // weight tiles per x chunk
  u32x4 ring[16];
  int wi = 0;
  if (WLOAD) {
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + (size_t)min(wi + i, wtiles - 1) * 1024));
  }
  u32x4 xs[FPW];
  if (XLOAD) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) xs[i] = *reinterpret_cast<const u32x4*>(X + ((size_t)(wave * FPW + i)) * 1024 + lane * 16);
  }
  for (int c = 0; c < xchunks; ++c) {
    unsigned char* buf = smem + (c & 1) * 65536;
    if (XLOAD) {
#pragma unroll
      for (int i = 0; i < FPW; ++i) *reinterpret_cast<u32x4*>(buf + (wave * FPW + i) * 1024 + lane * 16) = xs[i];
      if (c + 1 < xchunks) {
#pragma unroll
        for (int i = 0; i < FPW; ++i) xs[i] = *reinterpret_cast<const u32x4*>(X + ((size_t)(c + 1) * 64 + wave * FPW + i) * 1024 + lane * 16);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (LDSREAD) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { const u32x4 v = *reinterpret_cast<const u32x4*>(buf + (nt * 16 + j) * 1024 + lane * 16); acc ^= v[0] ^ v[3]; }
    }
    if (WLOAD) {
      for (int t = 0; t < wper; t += 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc ^= ring[i][0] ^ ring[i][1] ^ ring[i][2] ^ ring[i][3];
          ring[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + (size_t)min(wi + 16 + i, wtiles - 1) * 1024));
        }
        wi += 16;
      }
    }
  }
  if (WLOAD && xchunks == 0) {
    for (int t = 0; t < wtiles; t += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc ^= ring[i][0] ^ ring[i][1] ^ ring[i][2] ^ ring[i][3];
        ring[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + (size_t)min(wi + 16 + i, wtiles - 1) * 1024));
      }
      wi += 16;
    }
  }
  if (WLOAD) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= ring[i][0];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t wtotal = (size_t)3 << 30;                    // distinct weight bytes per rep so that nothing is cache resident
  unsigned char *X, *W; unsigned* sink;
  CK(hipMalloc(&X, 1 << 20)); CK(hipMemset(X, 1, 1 << 20));
  CK(hipMalloc(&W, wtotal)); CK(hipMemset(W, 2, wtotal));
  CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256, reps = 12;
  auto run = [&](const char* name, auto kern, int waves, int xchunks, size_t wbytes) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const size_t per_wave = (wbytes / ((size_t)blocks * waves)) & ~(size_t)1023;
    auto go = [&](int r) { const size_t off = wbytes ? ((size_t)r * (wbytes + (1 << 20))) % (wtotal - wbytes - (1 << 20)) : 0; hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 131072, s, X, xchunks, W + (off & ~(size_t)1023), per_wave, sink); };
    go(0); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < reps; ++r) go(r + 1); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-74s %7.2f us  x %5.1f GB/s per CU  weights %5.2f TB/s\n", name, us, xchunks * 65536.0 / us / 1e3, (double)per_wave * blocks * waves / us / 1e6);
  };
  printf("-- 256 blocks (one per CU); x = 64 KiB chunks, L2-resident (the same 512 KiB for every block); back-to-back launches, so each time includes one boundary (~1.5 us)\n");
  run("8 waves: x 8 chunks (512 KiB / CU) -> LDS", k_probe<8, true, false, false>, 8, 8, 0);
  run("8 waves: x 8 chunks -> LDS + fragment reads", k_probe<8, true, false, true>, 8, 8, 0);
  run("16 waves: x 8 chunks -> LDS", k_probe<16, true, false, false>, 16, 8, 0);
  run("4 waves: x 8 chunks -> LDS", k_probe<4, true, false, false>, 4, 8, 0);
  run("8 waves: x 2 chunks (128 KiB / CU) -> LDS", k_probe<8, true, false, false>, 8, 2, 0);
  run("8 waves: x 1 chunk (64 KiB / CU) -> LDS", k_probe<8, true, false, false>, 8, 1, 0);
  run("8 waves: no x, 8 barriers only", k_probe<8, false, false, false>, 8, 8, 0);
  for (size_t mb : {17, 50, 90, 180}) {
    char nm[128];
    snprintf(nm, sizeof nm, "8 waves: weights only, %zu MB", mb); run(nm, k_probe<8, false, true, false>, 8, 0, mb << 20);
    snprintf(nm, sizeof nm, "8 waves: weights %zu MB + x 8 chunks + fragment reads", mb); run(nm, k_probe<8, true, true, true>, 8, 8, mb << 20);
    snprintf(nm, sizeof nm, "8 waves: weights %zu MB + x 2 chunks + fragment reads", mb); run(nm, k_probe<8, true, true, true>, 8, 2, mb << 20);
    snprintf(nm, sizeof nm, "16 waves: weights %zu MB + x 8 chunks + fragment reads", mb); run(nm, k_probe<16, true, true, true>, 16, 8, mb << 20);
  }
  return 0;
}
