// kernels_decode_mv.hip — the batched decode step for contexts with at most 4 decoding slots ("multi-vector" step).
//
// BASELINE config 4 (16 rollouts of one image, root-parallel over the ranks: /root/reference examples/eval.py:80-83,108-137)
// leaves 8 / 4 / 2 trees per rank at N = 2 / 4 / 8.  Two or four sequences do not fill a 16-column MFMA tile: the
// fragment-major kernels of kernels_batch_decode.hip cost 3.4 ms per step whether 2 or 16 slots decode, against 2.62 ms for the
// single-sequence graph.  k_gemv_mv is the single-sequence GEMV (kernels_decode.hip: k_gemv) carrying NB <= 4 input vectors:
// the same row-major weights, streamed ONCE with the same non-temporal 16-byte loads into the same two register stages, and
// every weight chunk is folded into NB accumulators against NB x vectors held in LDS (v_dot2c_f32_bf16: the step stays
// HBM-bound — 2.5 dot2 lanes per clock and CU per vector against 64 available).  Per slot the arithmetic is exactly k_gemv's:
// the same chunk order per lane, the same wave reduction, the same HF rounding points in the epilogues — a slot's result does
// not depend on NB or on which other slots are active, and a PRO_COPY role equals the single-sequence kernel bit for bit
// (tests/test_gpu_parity.py::test_multi_vector_gemv_is_the_single_sequence_gemv_per_vector).
//
// One step = sampler + L x [ rmsnorm+qkv+RoPE+KV | attention per (head, slot) | o_proj+residual | rmsnorm+gate/up+SiLU*mul |
// down+residual ] + rmsnorm+lm_head: five launches per layer like the single-sequence step (the MFMA step needs seven: its
// RMSNorms are kernels of their own).  Inputs / outputs live where the MFMA step keeps them (residual streams row-major,
// attention output and SwiGLU activation in B-operand fragment order, common.h xtile_off: slots 0..3 of one 8-k group are one
// contiguous 64-byte line), so attention, sampler, KV fork / resume are the kernels of kernels_batch_decode.hip unchanged.
#include "kernels.h"
#include "gemv_inl.h"

template <int NB, int NR, int U>
__device__ __forceinline__ void mv_fma(float (&acc)[NR][NB], const u32x4 (&w)[NR][U], const u32x4* xs, int K8,
                                       int g, int lane) {
  const bool full = 64 * (g * U + U) <= K8;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = lane + 64 * (g * U + u);
    const bool ok = full || c < K8;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      u32x4 xv = {0u, 0u, 0u, 0u};
      if (ok) xv = xs[b * K8 + c];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r][b] = dot8(w[r][u], xv, acc[r][b]);
    }
  }
}

// fp8 rows: a 16-byte chunk = 16 weights, widened once to 8 packed bf16 pairs (exact) and folded against 32 bytes of each x
template <int NB, int NR, int U>
__device__ __forceinline__ void mv_fma_f8(float (&acc)[NR][NB], const u32x4 (&w)[NR][U], const u32x4* xs, int K8, int KC,
                                          int g, int lane) {
  const bool full = 64 * (g * U + U) <= KC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = lane + 64 * (g * U + u);
    const bool ok = full || c < KC;
    bf16x2_t wl[NR][4], wh[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wl[r][j] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[r][u][j], 1.0f, false);
        wh[r][j] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[r][u][j], 1.0f, true);
      }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      u32x4 x0 = {0u, 0u, 0u, 0u}, x1 = {0u, 0u, 0u, 0u};
      if (ok) { x0 = xs[b * K8 + 2 * c]; x1 = xs[b * K8 + 2 * c + 1]; }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        float s = acc[r][b];                 // the order of dot16_f8 (gemv_inl.h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t xa = (j < 2) ? x0[2 * j] : x1[2 * j - 4];
          const uint32_t xb = (j < 2) ? x0[2 * j + 1] : x1[2 * j - 3];
          s = __builtin_amdgcn_fdot2_f32_bf16(wl[r][j], __builtin_bit_cast(bf16x2_t, xa), s, false);
          s = __builtin_amdgcn_fdot2_f32_bf16(wh[r][j], __builtin_bit_cast(bf16x2_t, xb), s, false);
        }
        acc[r][b] = s;
      }
    }
  }
}

// Epilogue of one (unit, slot): the roundings of gemv_epilogue (kernels_decode.hip), addressed per slot.
template <int EPI, bool F8>
__device__ __forceinline__ void mv_epilogue(const GemvMvArgs& a, int slot, int u, float a0, float a1, float pre0, float pre1, int pos) {
  if (F8) {
    if (EPI == EPI_QKV) {
      const int r0 = (u >> 6) * 128 + (u & 63);
      a0 *= a.wscale[r0];
      a1 *= a.wscale[r0 + 64];
    } else if (EPI == EPI_SWIGLU) {
      a0 *= a.wscale[u];
      a1 *= a.wscale[a.ff + u];
    } else {
      a0 *= a.wscale[u];
    }
  }
  if (EPI == EPI_STORE) {
    a.Y[(size_t)slot * a.ldy + u] = f2bf(a0);
  } else if (EPI == EPI_RESID) {
    a.Y[(size_t)slot * a.ldy + u] = f2bf(pre0 + rbf(a0));
  } else if (EPI == EPI_LOGITS) {
    a.logits[(size_t)slot * a.N + u] = rbf(a0);
  } else if (EPI == EPI_SWIGLU) {
    const float gte = rbf(a0);
    const float up = rbf(a1);
    const float sl = rbf(gte / (1.f + expf(-gte)));
    a.Y[xtile_off(slot, u, (a.ff + 31) >> 5)] = f2bf(sl * up);     // input of the down projection: fragment-major
  } else if (EPI == EPI_QKV) {
    const int hb = u >> 6, i = u & 63;
    const int sec = hb < a.H ? 0 : (hb < a.H + a.KVH ? 1 : 2);
    const int head = sec == 0 ? hb : (sec == 1 ? hb - a.H : hb - a.H - a.KVH);
    const size_t slot_kv = (size_t)slot * a.kv_slot_stride;
    const float x1 = rbf(a0);      // dim i
    const float x2 = rbf(a1);      // dim i + 64
    if (sec == 2) {
      bf16_t* dst = a.vcache + slot_kv + ((size_t)head * a.T_max + pos) * 128;
      dst[i] = f2bf(x1);
      dst[i + 64] = f2bf(x2);
    } else {
      const float c = pre0, s = pre1;
      const float o1 = rbf(rbf(x1 * c) + rbf(-x2 * s));
      const float o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
      bf16_t* dst = (sec == 0) ? (a.q_out + (size_t)slot * a.d + head * 128)
                               : (a.kcache + slot_kv + ((size_t)head * a.T_max + pos) * 128);
      dst[i] = f2bf(o1);
      dst[i + 64] = f2bf(o2);
    }
  }
}

// NB = input vectors (slots 0..NB-1); R / U / WAVES / PERSIST as in k_gemv.  Lane b < NB of a wave runs slot b's epilogue.
template <int PRO, int EPI, int NB, int R, int U, int WAVES, bool PERSIST, bool F8 = false>
__global__ __launch_bounds__(WAVES * 64) void k_gemv_mv(GemvMvArgs a) {
  constexpr bool PAIRED = (EPI == EPI_QKV) || (EPI == EPI_SWIGLU);
  constexpr int NR = PAIRED ? 2 * R : R;
  constexpr int THREADS = WAVES * 64;
  static_assert(WAVES >= 4, "the RMSNorm prologue runs on the block's first four waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);    // [NB][K8]
  const int K8 = a.K >> 3;                       // 16-byte chunks of one x vector (bf16)
  const int KC = F8 ? (a.K >> 4) : K8;           // 16-byte chunks of one weight row
  const size_t row_bytes = F8 ? (size_t)a.K : (size_t)a.K * 2;
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(F8 ? (const void*)a.W8 : (const void*)a.W);
  float* red = reinterpret_cast<float*>(smem + (size_t)NB * K8 * 16);  // [NB][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  int n_units;
  if (EPI == EPI_QKV) n_units = (a.N >> 1);
  else if (EPI == EPI_SWIGLU) n_units = a.ff;
  else n_units = a.N;
  const int chunk_stride = PERSIST ? (int)gridDim.x * WAVES : 0;
  int chunk = PERSIST ? (int)blockIdx.x + (int)gridDim.x * wave : (int)blockIdx.x * WAVES + wave;
  int unit0 = chunk * R;

  const u32x4* rows[NR];
  auto set_rows = [&](int u0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int u = u0 + j;
      if (u >= n_units) u = n_units - 1;  // clamped: inactive tails never fault
      int r0, r1 = 0;
      if (EPI == EPI_QKV) {   // unit = RoPE pair (i, i+64) of head block hb over [H q | KVH k | KVH v]
        r0 = (u >> 6) * 128 + (u & 63);
        r1 = r0 + 64;
      } else if (EPI == EPI_SWIGLU) {
        r0 = u;
        r1 = a.ff + u;
      } else {
        r0 = u;
      }
      if (PAIRED) {
        rows[2 * j] = reinterpret_cast<const u32x4*>(Wb + (size_t)r0 * row_bytes);
        rows[2 * j + 1] = reinterpret_cast<const u32x4*>(Wb + (size_t)r1 * row_bytes);
      } else {
        rows[j] = reinterpret_cast<const u32x4*>(Wb + (size_t)r0 * row_bytes);
      }
    }
  };
  set_rows(unit0);

  const int iters = (KC + 63) >> 6;
  const int G = (iters + U - 1) / U;
  u32x4 wa[NR][U], wb[NR][U];
  float acc[NR][NB];

  // first stage of weights goes in flight before the prologue touches x
  if (G > 0) gemv_load<NR, U>(wa, rows, 0, lane, KC);

  // this lane's slot (lanes 0..NB-1 run the epilogues) and the operands its epilogue needs from memory
  const bool mine = lane < NB && (a.bs ? a.bs->active[lane] != 0 : true);
  float pre0[R], pre1[R];
  int pos = 0;
  if (EPI == EPI_QKV && mine) pos = a.st[lane].pos;
  auto prefetch_epilogue = [&](int u0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      pre0[j] = 0.f; pre1[j] = 0.f;
      if (mine) {
        int u = u0 + j;
        if (u >= n_units) u = n_units - 1;
        if (EPI == EPI_RESID) pre0[j] = bf2f(a.Y[(size_t)lane * a.ldy + u]);
        if (EPI == EPI_QKV) {
          pre0[j] = bf2f(a.rope_cos[(size_t)pos * 64 + (u & 63)]);
          pre1[j] = bf2f(a.rope_sin[(size_t)pos * 64 + (u & 63)]);
        }
      }
    }
  };
  if (EPI == EPI_RESID || EPI == EPI_QKV) prefetch_epilogue(unit0);

  // ---- prologue: the NB bf16 input vectors in LDS
  if (PRO == PRO_COPY) {
    if (a.x_rowmajor) {
      for (int i = tid; i < NB * K8; i += THREADS) {
        const int b = i / K8, c = i - b * K8;
        xs[i] = reinterpret_cast<const u32x4*>(a.X + (size_t)b * a.ldx)[c];
      }
    } else {   // fragment order (xtile_off): the 16-byte pieces of slots 0..3 of one 8-k group are adjacent
      const u32x4* x4 = reinterpret_cast<const u32x4*>(a.X);
      for (int i = tid; i < NB * K8; i += THREADS) {
        const int c = i / NB, b = i - c * NB;
        xs[b * K8 + c] = x4[(size_t)(c >> 2) * 64 + (c & 3) * 16 + b];
      }
    }
  } else {     // PRO_RMSNORM: residual streams are row-major [slot][ldx]
    // The statistics are computed by the block's first 256 threads in the order of a 256-thread block WHATEVER the block
    // size is (thread t folds chunks t, t + 256, ...; 4 wave sums added in wave order): the normalised vector — hence every
    // logit — does not depend on the block shape a role runs in, so a slot's result is the same for 1, 2 and 4 vectors even
    // where those take different shapes, and equals k_gemv's 4-wave kernels.  (One memory round trip either way.)
    constexpr int NT = 256, NW = 4;
    const u32x4* w4 = reinterpret_cast<const u32x4*>(a.norm_w);
    const bool worker = tid < NT;
    if (K8 <= 2 * NT) {
      // as k_gemv: x and the norm weight in ONE memory round trip, kept in registers across the block reduction
      const int c0 = tid, c1 = tid + NT;
      const bool h0 = worker && c0 < K8, h1 = worker && c1 < K8;
      const u32x4 zz = {0u, 0u, 0u, 0u};
      u32x4 v0[NB], v1[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const u32x4* x4 = reinterpret_cast<const u32x4*>(a.X + (size_t)b * a.ldx);
        v0[b] = h0 ? x4[c0] : zz;
        v1[b] = h1 ? x4[c1] : zz;
      }
      const u32x4 g0 = h0 ? w4[c0] : zz, g1 = h1 ? w4[c1] : zz;
      if (worker) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = pk_lo(v0[b][e]), hi = pk_hi(v0[b][e]);
            ss += lo * lo;
            ss += hi * hi;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = pk_lo(v1[b][e]), hi = pk_hi(v1[b][e]);
            ss += lo * lo;
            ss += hi * hi;
          }
          ss = wave_sum(ss);
          if (lane == 0) red[b * NW + wave] = ss;
        }
      }
      __syncthreads();
      if (worker) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float tot = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) tot += red[b * NW + w];
          const float inv = rsqrtf(tot / (float)a.K + a.eps);
          u32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // HF LlamaRMSNorm: weight * (x * rsqrt(var+eps)).to(bf16)
            o0[e] = pack2(pk_lo(g0[e]) * rbf(pk_lo(v0[b][e]) * inv), pk_hi(g0[e]) * rbf(pk_hi(v0[b][e]) * inv));
            o1[e] = pack2(pk_lo(g1[e]) * rbf(pk_lo(v1[b][e]) * inv), pk_hi(g1[e]) * rbf(pk_hi(v1[b][e]) * inv));
          }
          if (h0) xs[b * K8 + c0] = o0;
          if (h1) xs[b * K8 + c1] = o1;
        }
      }
    } else {
      if (worker) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const u32x4* x4 = reinterpret_cast<const u32x4*>(a.X + (size_t)b * a.ldx);
          float ss = 0.f;
          for (int c = tid; c < K8; c += NT) {
            const u32x4 v = x4[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = pk_lo(v[e]), hi = pk_hi(v[e]);
              ss += lo * lo;
              ss += hi * hi;
            }
          }
          ss = wave_sum(ss);
          if (lane == 0) red[b * NW + wave] = ss;
        }
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const u32x4* x4 = reinterpret_cast<const u32x4*>(a.X + (size_t)b * a.ldx);
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[b * NW + w];
        const float inv = rsqrtf(tot / (float)a.K + a.eps);
        for (int c = tid; c < K8; c += THREADS) {
          const u32x4 v = x4[c];
          const u32x4 g = w4[c];
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float nlo = rbf(pk_lo(v[e]) * inv), nhi = rbf(pk_hi(v[e]) * inv);
            o[e] = pack2(pk_lo(g[e]) * nlo, pk_hi(g[e]) * nhi);
          }
          xs[b * K8 + c] = o;
        }
      }
    }
  }
  __syncthreads();

  for (;;) {
    const bool active = unit0 < n_units;  // wave-uniform
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    // ---- main loop: two register stages (wa holds stage 0 on entry)
    for (int g = 0; g < G; g += 2) {
      if (g + 1 < G) gemv_load<NR, U>(wb, rows, g + 1, lane, KC);
      if (F8) mv_fma_f8<NB, NR, U>(acc, wa, xs, K8, KC, g, lane); else mv_fma<NB, NR, U>(acc, wa, xs, K8, g, lane);
      if (g + 1 < G) {
        if (g + 2 < G) gemv_load<NR, U>(wa, rows, g + 2, lane, KC);
        if (F8) mv_fma_f8<NB, NR, U>(acc, wb, xs, K8, KC, g + 1, lane); else mv_fma<NB, NR, U>(acc, wb, xs, K8, g + 1, lane);
      }
    }
    const int cur = unit0;
    float q0[R], q1[R];          // this chunk's epilogue operands (the next chunk's are fetched below)
#pragma unroll
    for (int j = 0; j < R; ++j) { q0[j] = pre0[j]; q1[j] = pre1[j]; }
    if (PERSIST) {  // next chunk's first stage goes in flight before this chunk's reduction
      chunk += chunk_stride;
      unit0 = chunk * R;
      if (unit0 < n_units) {
        set_rows(unit0);
        gemv_load<NR, U>(wa, rows, 0, lane, KC);
        if (EPI == EPI_RESID || EPI == EPI_QKV) prefetch_epilogue(unit0);
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);

    // ---- epilogue: lane b owns slot b (every lane holds every sum after the butterfly)
    if (active && mine) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int u = cur + j;
        if (u >= n_units) break;
        float s0 = PAIRED ? acc[2 * j][0] : acc[j][0];
        float s1 = PAIRED ? acc[2 * j + 1][0] : 0.f;
#pragma unroll
        for (int b = 1; b < NB; ++b) {
          if (lane == b) {
            s0 = PAIRED ? acc[2 * j][b] : acc[j][b];
            if (PAIRED) s1 = acc[2 * j + 1][b];
          }
        }
        mv_epilogue<EPI, F8>(a, lane, u, s0, s1, q0[j], q1[j], pos);
      }
    }
    if (!PERSIST || unit0 >= n_units) break;
  }
}

static int mv_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int PRO, int EPI, int NB, int R, int U, int WAVES, bool PERSIST, bool F8>
static void launch_mv_t(const GemvMvArgs& a, hipStream_t s, int blocks_per_cu) {
  const int n_units = (EPI == EPI_QKV) ? (a.N >> 1) : (EPI == EPI_SWIGLU ? a.ff : a.N);
  const int per_block = WAVES * R;
  int grid = (n_units + per_block - 1) / per_block;
  if (PERSIST) {
    const int cap = mv_num_cus() * (blocks_per_cu > 0 ? blocks_per_cu : 2);
    if (grid > cap) grid = cap;
  }
  const int lds = (int)((size_t)NB * (a.K >> 3) * 16 + (size_t)NB * WAVES * 4 + 64);
  auto fn = k_gemv_mv<PRO, EPI, NB, R, U, WAVES, PERSIST, F8>;
  if (lds > 64 * 1024) {      // down projection with 4 vectors (ff = 11008: 88 KB of x): one block per CU
    static int raised[64] = {};          // per device (the attribute is a per-device property of the function)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (raised[dev] < lds) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); raised[dev] = lds; }
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(WAVES * 64), lds, s, a);
}

// Shapes per role.  shape 0 = the single-sequence default of the role (one unit per wave, the grid covers N: every block
// re-stages the NB vectors, NB x the prologue traffic of k_gemv); 1 = persistent, 8 waves, 2 blocks per CU; 2 = persistent,
// 4 waves, 4 blocks per CU; 3 = persistent, 16 waves, 1 block per CU (x staged once per CU).  The measured choice per role
// and NB is mv_default_shape() below (tools/tune_mv.py, profiles/r04_tune_mv.txt).
static int g_mv_shape[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // indexed like launch_gemv's table: epi, 5 = o_proj
void set_gemv_mv_shape(int role, int shape) { if (role >= 0 && role < 8) g_mv_shape[role] = shape; }

static int mv_default_shape(int epi, bool o_proj, int nb, int K) {
  if (nb <= 1) return 0;                                   // one vector: the single-sequence shapes
  const size_t x_bytes = (size_t)nb * K * 2;
  if (x_bytes > 64 * 1024) return 3;                       // 88 KB of x: one block per CU is all that fits
  // measured, ds-7b, ms per step (profiles/r04_tune_mv_ds7b.txt; every shape gives the same bits):
  //   2 vectors: shape 0 wins every role (2.867-2.869) except down, where one 16-wave block per CU does (2.841)
  //   4 vectors: qkv 3.268 / 3.194 / 3.233 / 3.204 -> 1; o_proj 3.186 / 3.193 / 3.224 / 3.178 -> 3; gate/up 3.227 / 3.190 /
  //              3.183 / 3.115 -> 3; lm_head 3.207 / 3.191 / 3.190 / 3.186 -> 3
  if (nb == 2) return (epi == EPI_RESID && !o_proj) ? 3 : 0;
  if (epi == EPI_QKV) return 1;
  return 3;
}

// fp8 rows hold 16 weights per 16-byte chunk (two x chunks each): half the chunks per stage keep the same K span in flight;
// the one-block-per-CU shape runs 8 waves for fp8 (16 waves leave 128 VGPRs per lane: the widened pairs of 4 vectors spill)
#define MV(PRO, EPI, NBV, R, U, W, P, BPC)                                                                     \
  do {                                                                                                          \
    if (a.W8) launch_mv_t<PRO, EPI, NBV, R, ((U) >= 2 ? (U) / 2 : 1), ((W) == 16 ? 8 : (W)), P, true>(a, s, BPC);  \
    else launch_mv_t<PRO, EPI, NBV, R, U, W, P, false>(a, s, BPC);                                              \
    return;                                                                                                     \
  } while (0)

template <int NB>
static void launch_mv_nb(int pro, int epi, int shape, const GemvMvArgs& a, hipStream_t s) {
  constexpr int UR = NB == 1 ? 8 : 4;     // chunks per stage of the N = d roles: 8 x-chunk reads per vector and stage spill at NB >= 2
  if (pro == PRO_RMSNORM && epi == EPI_QKV) {
    switch (shape) {
      default: MV(PRO_RMSNORM, EPI_QKV, NB, 1, 2, 4, false, 0);
      case 1: MV(PRO_RMSNORM, EPI_QKV, NB, 1, 2, 8, true, 2);
      case 2: MV(PRO_RMSNORM, EPI_QKV, NB, 1, 2, 4, true, 4);
      case 3: MV(PRO_RMSNORM, EPI_QKV, NB, 1, 2, 16, true, 1);
    }
  }
  if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) {
    switch (shape) {
      default: MV(PRO_RMSNORM, EPI_SWIGLU, NB, 1, 2, 4, false, 0);
      case 1: MV(PRO_RMSNORM, EPI_SWIGLU, NB, 1, 2, 8, true, 2);
      case 2: MV(PRO_RMSNORM, EPI_SWIGLU, NB, 1, 2, 4, true, 4);
      case 3: MV(PRO_RMSNORM, EPI_SWIGLU, NB, 1, 2, 16, true, 1);
    }
  }
  if (pro == PRO_RMSNORM && epi == EPI_LOGITS) {
    switch (shape) {
      default: MV(PRO_RMSNORM, EPI_LOGITS, NB, 1, 4, 4, false, 0);
      case 1: MV(PRO_RMSNORM, EPI_LOGITS, NB, 1, 4, 8, true, 2);
      case 2: MV(PRO_RMSNORM, EPI_LOGITS, NB, 1, 4, 4, true, 4);
      case 3: MV(PRO_RMSNORM, EPI_LOGITS, NB, 1, 2, 16, true, 1);
    }
  }
  if (pro == PRO_COPY && epi == EPI_RESID) {
    switch (shape) {
      default: MV(PRO_COPY, EPI_RESID, NB, 1, UR, 8, false, 0);
      case 1: MV(PRO_COPY, EPI_RESID, NB, 1, UR, 8, true, 2);
      case 2: MV(PRO_COPY, EPI_RESID, NB, 1, UR, 4, true, 4);
      case 3: MV(PRO_COPY, EPI_RESID, NB, 1, 4, 16, true, 1);
    }
  }
  if (pro == PRO_RMSNORM) MV(PRO_RMSNORM, EPI_STORE, NB, 1, 2, 4, false, 0);   // op-level tests
  MV(PRO_COPY, EPI_STORE, NB, 1, 2, 4, false, 0);
}
#undef MV

void launch_gemv_mv(int pro, int epi, int nb, const GemvMvArgs& a, hipStream_t s) {
  const bool o_proj = epi == EPI_RESID && a.K == a.d;
  int shape = g_mv_shape[epi & 7];
  if (o_proj && g_mv_shape[5] >= 0) shape = g_mv_shape[5];
  if (shape < 0) shape = mv_default_shape(epi, o_proj, nb, a.K);
  if ((size_t)nb * a.K * 2 > 64 * 1024 && shape != 3) shape = 3;     // more than one such block does not fit a CU's LDS
  if (nb <= 1) launch_mv_nb<1>(pro, epi, shape, a, s);
  else if (nb == 2) launch_mv_nb<2>(pro, epi, shape, a, s);
  else launch_mv_nb<4>(pro, epi, shape, a, s);
}
