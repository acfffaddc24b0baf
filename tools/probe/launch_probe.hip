// launch_probe.hip — what does ONE dependent kernel boundary cost inside a hipGraph replay on this machine?
// Chains of N kernels (each depends on the previous one through the stream order) with different grid sizes / work.
//   hipcc --offload-arch=gfx950 -O3 -o launch_probe launch_probe.hip && ./launch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void k_empty(int* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1; }
__global__ void k_touch(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
// streams `bytes` bytes once (16 B per lane, grid-stride), tiny reduction so the loads are not dead
__global__ __launch_bounds__(256) void k_stream(const u32x4* src, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_nontemporal_load(src + i);
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// GEMV-shaped access: each WAVE streams whole rows of `row16` 16-byte chunks (rows_per_block rows per block, one per wave),
// U loads in flight per lane, no x, xor instead of dot products: isolates the access pattern / grid shape of k_gemv
template <int U>
__global__ __launch_bounds__(256) void k_rows(const u32x4* src, int nrows, int row16, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= nrows) return;
  const u32x4* r = src + (size_t)row * row16;
  unsigned acc = 0;
  for (int c0 = 0; c0 < row16; c0 += 64 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int c = c0 + u * 64 + lane; v[u] = c < row16 ? __builtin_nontemporal_load(r + c) : (u32x4){0u, 0u, 0u, 0u}; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  for (int off = 32; off > 0; off >>= 1) acc ^= __shfl_xor(acc, off, 64);
  if (acc == 0x12345678u && lane == 0) sink[0] = acc;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  int* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  const size_t big = (size_t)2 << 30;   // 2 GiB of distinct data so nothing is cache resident
  unsigned char* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
  unsigned* sink = reinterpret_cast<unsigned*>(p) + 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 200, reps = 20;
  auto run = [&](const char* name, auto launch_one, double bytes_per_kernel) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) launch_one(i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * N);
    if (bytes_per_kernel > 0) printf("%-58s %7.2f us/kernel  (%6.2f us over bytes/7.7 TB/s, %5.2f TB/s)\n", name, us, us - bytes_per_kernel / 7.7e6, bytes_per_kernel / us / 1e6);
    else printf("%-58s %7.2f us/kernel\n", name, us);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  };
  run("empty kernel, 1 block x 64", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }, 0);
  run("empty kernel, 256 blocks x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); }, 0);
  run("empty kernel, 5504 blocks x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(5504), dim3(256), 0, s, p); }, 0);
  run("1-thread read-modify-write chain, 1 block", [&](int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, p); }, 0);
  for (size_t mb : {8, 33, 90, 180}) {
    for (int grid : {1024, 4096}) {
      const size_t bytes = mb << 20;
      char nm[96]; snprintf(nm, sizeof nm, "stream %3zu MiB per kernel, %4d blocks x 256", mb, grid);
      run(nm, [&](int i) { const size_t off = ((size_t)i * bytes) % (big - bytes); hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, reinterpret_cast<const u32x4*>(buf + (off & ~(size_t)255)), bytes / 16, sink); }, (double)bytes);
    }
  }
  struct Shape { const char* name; int N, K; } shapes[] = {{"o_proj 4096 x 4096", 4096, 4096}, {"down 4096 x 11008", 4096, 11008},
                                                            {"qkv 12288 x 4096", 12288, 4096}, {"gate/up 22016 x 4096", 22016, 4096}};
  for (auto& sh : shapes) {
    const size_t bytes = (size_t)sh.N * sh.K * 2;
    char nm[96];
    snprintf(nm, sizeof nm, "row-per-wave U=8  %s", sh.name);
    run(nm, [&](int i) { const size_t off = ((size_t)i * bytes) % (big - bytes); hipLaunchKernelGGL((k_rows<8>), dim3((sh.N + 3) / 4), dim3(256), 0, s, reinterpret_cast<const u32x4*>(buf + (off & ~(size_t)255)), sh.N, sh.K / 8, sink); }, (double)bytes);
    snprintf(nm, sizeof nm, "row-per-wave U=4  %s", sh.name);
    run(nm, [&](int i) { const size_t off = ((size_t)i * bytes) % (big - bytes); hipLaunchKernelGGL((k_rows<4>), dim3((sh.N + 3) / 4), dim3(256), 0, s, reinterpret_cast<const u32x4*>(buf + (off & ~(size_t)255)), sh.N, sh.K / 8, sink); }, (double)bytes);
    snprintf(nm, sizeof nm, "grid-stride 4096 blocks, same bytes  %s", sh.name);
    run(nm, [&](int i) { const size_t off = ((size_t)i * bytes) % (big - bytes); hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s, reinterpret_cast<const u32x4*>(buf + (off & ~(size_t)255)), bytes / 16, sink); }, (double)bytes);
  }
  return 0;
}
