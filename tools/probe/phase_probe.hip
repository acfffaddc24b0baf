// phase_probe.hip — what ONE compute wave of the loader-wave batched GEMVs (k_gemv_bl) costs per phase of 4 k-steps, without any
// DMA, flags or other kernels' traffic: 24 ds_read_b128 (8 A + 16 B fragments) + 32 v_mfma_f32_16x16x32_bf16 on 8 accumulators,
// in the order the shipped kernel uses.  Variants leave the reads or the MFMAs out, issue all reads first, or split the unit over
// two waves by column tiles (16 reads + 16 MFMAs each).  1 block per CU, W compute waves per block (one per SIMD up to 4).
//   hipcc --offload-arch=gfx950 -O3 -o phase_probe tools/probe/phase_probe.hip && ./phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
// explicit LDS reads (a volatile access through a generic pointer compiles to FLAT loads + vmcnt(0)); the compiler does not count
// them, so every use is preceded by lds_wait()
__device__ __forceinline__ u32x4 lds_read(unsigned byte_off) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_off) : "memory"); return v; }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: per k-step 6 reads, wait, 8 MFMAs (no overlap at all: the upper bound of the shipped order), 1: reads only, 2: MFMAs only, 3: all 24 reads first, 4: half unit (2 of 4 column tiles)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_phase(float* out, int phases) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;   // small bf16 values
  __syncthreads();
  constexpr int NT = MODE == 4 ? 2 : 4;
  f32x4 c[2][NT];
  for (int t = 0; t < 2; ++t) for (int nt = 0; nt < NT; ++nt) c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned xb = lane * 16;                                      // LDS byte offsets (the dynamic segment starts at 0)
  const unsigned wb = 16 * 1024 + (wave * 8 * 1024) % (24 * 1024) + lane * 16;
  u32x4 keep = {0u, 0u, 0u, 0u};
  for (int p = 0; p < phases; ++p) {
    if (MODE == 3) {
      u32x4 xr[4][NT], wr[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xr[j][nt] = lds_read(xb + (nt * 4 + j) * 1024);
#pragma unroll
        for (int t = 0; t < 2; ++t) wr[t][j] = lds_read(wb + (t * 4 + j) * 1024);
      }
      lds_wait();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            c[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wr[t][j]), __builtin_bit_cast(bf16x8_t, xr[j][nt]), c[t][nt], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 xf[NT], af[2];
        if (MODE != 2) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) xf[nt] = lds_read(xb + (nt * 4 + j) * 1024);
#pragma unroll
          for (int t = 0; t < 2; ++t) af[t] = lds_read(wb + (t * 4 + j) * 1024);
          lds_wait();
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) xf[nt] = keep + (unsigned)j;
          af[0] = keep; af[1] = keep + 1u;
        }
        if (MODE == 1) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) keep ^= xf[nt];
          keep ^= af[0] ^ af[1];
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              c[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[t]), __builtin_bit_cast(bf16x8_t, xf[nt]), c[t][nt], 0, 0, 0);
        }
      }
    }
  }
  float s = (float)keep[0];
  for (int t = 0; t < 2; ++t) for (int nt = 0; nt < NT; ++nt) s += c[t][nt][0] + c[t][nt][3];
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static int run(const char* what, int waves, float* out) {
  const int phases = 20000, lds = 40 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_phase<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_phase<MODE>), dim3(256), dim3(64 * waves), lds, 0, out, 200);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((k_phase<MODE>), dim3(256), dim3(64 * waves), lds, 0, out, phases);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-72s %d waves/CU: %7.3f us per phase\n", what, waves, ms * 1e3 / phases);
  return 0;
}

int main() {
  float* out; CHECK(hipMalloc(&out, 256 * 256 * 4));
  for (int waves : {1, 3, 4}) {
    if (run<0>("24 ds_read_b128 + 32 MFMA, k-step by k-step, reads waited for", waves, out)) return 1;
    if (run<3>("24 ds_read_b128 first, then 32 MFMA", waves, out)) return 1;
    if (run<1>("24 ds_read_b128 only", waves, out)) return 1;
    if (run<2>("32 MFMA only", waves, out)) return 1;
    if (run<4>("half unit: 16 ds_read_b128 + 16 MFMA (2 of 4 column tiles)", waves, out)) return 1;
  }
  return 0;
}
