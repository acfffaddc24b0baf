// engine_probe.hip — does a persistent "layer engine" (one launch per token, grid barriers between the
// dependent GEMV phases, the first weight segment of the NEXT phase already in flight while a block waits
// at the barrier) beat one-kernel-per-phase launches replayed from a hipGraph?  Stand-alone measurement
// tool (no torch): ds-7b shaped phases  qkv 12288x4096 -> [2 tiny phases standing in for attention +
// combine] -> o 4096x4096 -> gate/up 22016x4096 -> down 4096x11008, L layers of distinct weights.
//   hipcc --offload-arch=gfx950 -O3 -o engine_probe engine_probe.hip && ./engine_probe [layers] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ uint16_t f2bf(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(unsigned short, h); }

struct Phase { const uint16_t* W; const uint16_t* x; uint16_t* y; int N, K; };
constexpr int MAXP = 8;
struct Layer { Phase p[MAXP]; int np; };

constexpr int THREADS = 512, WAVES = THREADS / 64;
constexpr int SEG = 8;              // 16-byte chunks per lane per segment -> 64*8*8 = 4096 elements
constexpr int XMAX = 11008;

// one segment of one row: lane's chunks c = seg*512 + i*64 + lane (i < SEG), guarded by nchunks
__device__ __forceinline__ void seg_load(u32x4 (&r)[SEG], const uint16_t* row, int seg, int nchunks, int lane) {
#pragma unroll
  for (int i = 0; i < SEG; ++i) {
    const int c = seg * (64 * SEG) + i * 64 + lane;
    if (c < nchunks) r[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row) + c);
    else r[i] = (u32x4){0u, 0u, 0u, 0u};
  }
}
__device__ __forceinline__ float seg_dot(const u32x4 (&r)[SEG], const u32x4* xs, int seg, int nchunks, int lane, float acc) {
#pragma unroll
  for (int i = 0; i < SEG; ++i) {
    const int c = seg * (64 * SEG) + i * 64 + lane;
    if (c < nchunks) {
      const u32x4 xv = xs[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = dot2(r[i][e], xv[e], acc);
    }
  }
  return acc;
}

// the GEMV phase body shared by both variants.  `pre` = the first segment of this wave's first row, already
// loaded (engine: issued before the barrier).  gw / W = this wave's global index / total waves.
__device__ __forceinline__ void phase_body(const Phase& ph, const u32x4* xs, int gw, int W, int lane, u32x4 (&cur)[SEG], bool have_pre) {
  const int nchunks = ph.K / 8;
  const int nseg = (nchunks + 64 * SEG - 1) / (64 * SEG);
  u32x4 nxt[SEG];
  int row = gw;
  if (row >= ph.N) return;
  if (!have_pre) seg_load(cur, ph.W + (size_t)row * ph.K, 0, nchunks, lane);
  while (row < ph.N) {
    float acc = 0.f;
    for (int s = 0; s < nseg; ++s) {
      // prefetch the next segment (same row, or the first segment of the wave's next row)
      const bool last = s + 1 == nseg;
      const int nrow = last ? row + W : row;
      const int ns = last ? 0 : s + 1;
      const bool more = nrow < ph.N;
      if (more) seg_load(nxt, ph.W + (size_t)nrow * ph.K, ns, nchunks, lane);
      acc = seg_dot(cur, xs, s, nchunks, lane, acc);
      if (more) {
#pragma unroll
        for (int i = 0; i < SEG; ++i) cur[i] = nxt[i];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) ph.y[row] = f2bf(acc * 0.01f);
    row += W;
  }
}

__device__ __forceinline__ void load_x(const Phase& ph, u32x4* xs, bool coherent) {
  const int nchunks = ph.K / 8;
  for (int c = threadIdx.x; c < nchunks; c += THREADS) {
    const u32x4* p = reinterpret_cast<const u32x4*>(ph.x) + c;
    xs[c] = coherent ? __builtin_nontemporal_load(p) : *p;
  }
  __syncthreads();
}

// ---- variant A: one kernel per phase (persistent-shaped grid), replayed from a graph
__global__ __launch_bounds__(THREADS) void k_phase(Phase ph) {
  __shared__ __attribute__((aligned(16))) u32x4 xs[XMAX / 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = gridDim.x * WAVES, gw = wave * gridDim.x + blockIdx.x;
  u32x4 cur[SEG];
  load_x(ph, xs, false);
  phase_body(ph, xs, gw, W, lane, cur, false);
}
__global__ void k_tiny(uint16_t* y) { if (threadIdx.x == 0) y[blockIdx.x] = (uint16_t)(y[blockIdx.x] + 1); }

// ---- variant B: the engine
// mode 0: flat agent-scope counter, relaxed polling.  mode 1: two-level — blocks of one XCD meet on an
// XCD-local counter with L2-resident (workgroup-scope) atomics, the last arriver of each XCD bumps the global
// agent-scope counter, everybody polls the global one.  counter[0] = global, counter[16 + 16*x] = XCD x.
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, unsigned epoch, int mode, unsigned per_xcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (mode == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    } else {
      unsigned* xc = counter + 16 + 16 * xcc_id();
      const unsigned v = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
      if (v == epoch * per_xcd) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * 8u) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(THREADS) void k_engine(const Layer* layers, int nlayers, unsigned* counter, unsigned base, int prefetch, int mode, unsigned epoch0) {
  __shared__ __attribute__((aligned(16))) u32x4 xs[XMAX / 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = gridDim.x * WAVES, gw = wave * gridDim.x + blockIdx.x;
  unsigned target = base, epoch = epoch0;
  u32x4 cur[SEG];
  bool have = false;
  for (int l = 0; l < nlayers; ++l) {
    const Layer& L = layers[l];
    for (int pi = 0; pi < L.np; ++pi) {
      const Phase ph = L.p[pi];
      if (ph.N == 0) {   // stand-in for a tiny phase (attention / combine): just the dependency
        if (blockIdx.x == 0 && threadIdx.x == 0) ph.y[0] = (uint16_t)(ph.y[0] + 1);
      } else {
        load_x(ph, xs, true);
        phase_body(ph, xs, gw, W, lane, cur, have);
      }
      have = false;
      // next phase's first weights go in flight before we wait for everybody
      const Phase* np = nullptr;
      if (pi + 1 < L.np) np = &L.p[pi + 1]; else if (l + 1 < nlayers) np = &layers[l + 1].p[0];
      if (prefetch && np && np->N > 0 && gw < np->N) { seg_load(cur, np->W + (size_t)gw * np->K, 0, np->K / 8, lane); have = true; }
      target += gridDim.x; ++epoch;
      grid_barrier(counter, target, epoch, mode, gridDim.x / 8);
    }
  }
}

int main(int argc, char** argv) {
  const int nl = argc > 1 ? atoi(argv[1]) : 8;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const int d = 4096, ff = 11008;
  struct Shape { int N, K; } shapes[6] = {{3 * d, d}, {0, 0}, {0, 0}, {d, d}, {2 * ff, d}, {d, ff}};
  hipStream_t s; CK(hipStreamCreate(&s));
  uint16_t* act[2]; for (auto& a : act) { CK(hipMalloc(&a, 65536 * 2)); CK(hipMemset(a, 0x3c, 65536 * 2)); }
  std::vector<Layer> host(nl);
  size_t bytes = 0;
  for (int l = 0; l < nl; ++l) {
    host[l].np = 6;
    for (int p = 0; p < 6; ++p) {
      Phase& ph = host[l].p[p];
      ph.N = shapes[p].N; ph.K = shapes[p].K; ph.x = act[p & 1]; ph.y = act[(p + 1) & 1]; ph.W = nullptr;
      if (ph.N) {
        uint16_t* w; size_t n = (size_t)ph.N * ph.K;
        CK(hipMalloc(&w, n * 2)); CK(hipMemset(w, 0x3c, n * 2)); ph.W = w; bytes += n * 2;
      }
    }
  }
  Layer* dl; CK(hipMalloc(&dl, sizeof(Layer) * nl)); CK(hipMemcpy(dl, host.data(), sizeof(Layer) * nl, hipMemcpyHostToDevice));
  unsigned* counter; CK(hipMalloc(&counter, 1024)); CK(hipMemset(counter, 0, 1024));
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_engine, THREADS, 0));
  printf("CUs %d, engine occupancy %d blocks/CU, weights %.2f GB over %d layers\n", cus, occ, bytes / 1e9, nl);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto fn, const char* name) {
    fn(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-44s %8.1f us/pass  %6.2f us/layer  %5.2f TB/s\n", name, ms * 1e3, ms * 1e3 / nl, bytes / (ms * 1e-3) / 1e12);
    return ms;
  };
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int grid = cus * bpc;
    // A: graph of per-phase kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int l = 0; l < nl; ++l) for (int p = 0; p < 6; ++p) {
      const Phase& ph = host[l].p[p];
      if (ph.N) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(THREADS), 0, s, ph);
      else hipLaunchKernelGGL(k_tiny, dim3(32), dim3(64), 0, s, ph.y);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    char nm[96]; snprintf(nm, sizeof nm, "A graph of per-phase kernels, %d blocks", grid);
    time_it([&] { CK(hipGraphLaunch(ge, s)); }, nm);
    if (bpc > occ) continue;
    for (int mode = 0; mode <= 1; ++mode)
    for (int pf = 0; pf <= 1; ++pf) {
      snprintf(nm, sizeof nm, "B engine, %d blocks, barrier=%d prefetch=%d", grid, mode, pf);
      CK(hipMemsetAsync(counter, 0, 1024, s)); unsigned base = 0, epoch = 0;
      time_it([&] {
        hipLaunchKernelGGL(k_engine, dim3(grid), dim3(THREADS), 0, s, dl, nl, counter, base, pf, mode, epoch);
        base += (unsigned)grid * 6u * (unsigned)nl; epoch += 6u * (unsigned)nl;
      }, nm);
    }
  }
  // barrier cost alone
  {
    std::vector<Layer> e(1); e[0].np = 0;
    // reuse engine with N = 0 phases only
    std::vector<Layer> bl(nl);
    for (int l = 0; l < nl; ++l) { bl[l].np = 6; for (int p = 0; p < 6; ++p) { bl[l].p[p] = host[l].p[p]; bl[l].p[p].N = 0; } }
    Layer* dbl; CK(hipMalloc(&dbl, sizeof(Layer) * nl)); CK(hipMemcpy(dbl, bl.data(), sizeof(Layer) * nl, hipMemcpyHostToDevice));
    for (int mode = 0; mode <= 1; ++mode) {
      CK(hipMemsetAsync(counter, 0, 1024, s)); unsigned base = 0, epoch = 0;
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      auto go = [&] { hipLaunchKernelGGL(k_engine, dim3(cus), dim3(THREADS), 0, s, dbl, nl, counter, base, 0, mode, epoch); base += cus * 6u * nl; epoch += 6u * nl; };
      go(); CK(hipStreamSynchronize(s));
      CK(hipEventRecord(a, s));
      for (int i = 0; i < iters; ++i) go();
      CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("barrier alone (mode %d): %.2f us per barrier (%d blocks)\n", mode, ms * 1e3 / iters / (6.0 * nl), cus);
    }
  }
  return 0;
}
