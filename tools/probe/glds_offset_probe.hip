// glds_offset_probe.hip — where does `global_load_lds_dwordx4 v, off offset:N` put its data?  (LDS address = M0 + 16 * lane, or
// M0 + N + 16 * lane?)  One wave loads one 1 KiB piece with offset:1024 from a buffer whose every dword holds its own index.
//   hipcc --offload-arch=gfx950 -O3 -o glds_offset_probe glds_offset_probe.hip && ./glds_offset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(const unsigned* g, unsigned* out) {
  extern __shared__ unsigned smem[];
  for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = 0xdeadbeefu;       // 4 KiB of LDS
  __syncthreads();
  unsigned keep; const unsigned dst = 0u;
  const unsigned* src = g + threadIdx.x * 4;                               // lane l: bytes 16 l .. 16 l + 15 of the buffer (+ offset)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = smem[i];
}
int main() {
  unsigned *g, *out, h[1024], hg[2048];
  for (int i = 0; i < 2048; ++i) hg[i] = i;
  CK(hipMalloc(&g, sizeof hg)); CK(hipMalloc(&out, sizeof h)); CK(hipMemcpy(g, hg, sizeof hg, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, out);
  CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  int first = -1; for (int i = 0; i < 1024; ++i) if (h[i] != 0xdeadbeefu) { first = i; break; }
  if (first < 0) { printf("nothing landed in the first 4 KiB of LDS\n"); return 0; }
  printf("offset:1024 with M0 = 0: data landed at LDS byte %d and holds global dword %u (global byte %u)\n", first * 4, h[first], h[first] * 4);
  printf("=> the instruction offset %s the LDS destination (and %s the global source)\n", first * 4 == 1024 ? "MOVES" : "does not move", h[first] * 4 == 1024 ? "moves" : "does not move");
  return 0;
}
