// kernels_decode.hip — the tokens/sec kernel set: one decoded token = 5 kernels per
// LLaMA layer + lm_head + sampler, all HBM-bound weight streaming (SURVEY §8 rows
// a·D-step, a·H, a·S).  Numerics follow HF modeling_llama.py in bf16: every tensor HF
// materialises is rounded to bf16 at the same point, dot products accumulate in fp32.
//
// GEMV family (k_gemv): y = W . x with W [N][K] bf16 streamed exactly once with
// non-temporal 16-byte loads straight into VGPRs (no LDS round trip: each weight byte is
// used by one wave once), x staged in LDS (bf16) by a fused prologue:
//   PRO_RMSNORM  x = rmsnorm(residual)            (input_layernorm / post_attention / final norm)
//   PRO_ATTN     x = combine of the split-K attention partials (flash-decode reduction)
//   PRO_COPY     x = activation vector
// and a fused epilogue on the wave-reduced sums:
//   EPI_QKV      RoPE (rotate-half) on q,k pairs, q -> scratch, k,v -> KV cache at pos
//   EPI_SWIGLU   silu(gate)*up
//   EPI_RESID    residual += y
//   EPI_LOGITS   fp32 logits
// Each wave owns NR weight rows for the full K, lanes stride K in 16-byte chunks
// (64 lanes x 16 B = 1 KiB per row per load instruction), two register stages so the
// next stage's loads are in flight while the current one is consumed by v_dot2c_f32_bf16.
#include <cstdlib>
#include <cstring>
#include "kernels.h"

template <int NR, int U>
__device__ __forceinline__ void gemv_load(u32x4 (&w)[NR][U], const u32x4* (&rows)[NR],
                                          int g, int lane, int K8) {
  if (64 * (g * U + U) <= K8) {  // wave-uniform: the whole group is inside the row -> no predication
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = lane + 64 * (g * U + u);
#pragma unroll
      for (int r = 0; r < NR; ++r) w[r][u] = ld_nt(rows[r] + c);
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = lane + 64 * (g * U + u);
    const bool ok = c < K8;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32x4 z = {0u, 0u, 0u, 0u};
      w[r][u] = ok ? ld_nt(rows[r] + c) : z;
    }
  }
}

template <int NR, int U>
__device__ __forceinline__ void gemv_fma(float (&acc)[NR], const u32x4 (&w)[NR][U],
                                         const u32x4* xs, int g, int lane, int K8) {
  const bool full = 64 * (g * U + U) <= K8;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = lane + 64 * (g * U + u);
    u32x4 xv = {0u, 0u, 0u, 0u};
    if (full || c < K8) xv = xs[c];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = dot8(w[r][u], xv, acc[r]);
  }
}

// fp8 (OCP e4m3) weights: a 16-byte chunk holds 16 weights of one row; they are widened to bf16 pairs
// (exact) and fed to the same v_dot2c_f32_bf16 against 32 bytes of x.  The per-row power-of-two scale is
// applied to the fp32 sum in the epilogue (exact), so the result equals the bf16 kernel on the
// de-quantised ("effective") weights bit for bit at equal accumulation order.
__device__ __forceinline__ float dot16_f8(const u32x4& w, const u32x4& x0, const u32x4& x1, float c) {
  // v_cvt_scalef32_pk_bf16_fp8: two e4m3 bytes -> packed bf16 pair in ONE instruction (exact, scale 1)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16x2_t lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j], 1.0f, false);
    const bf16x2_t hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j], 1.0f, true);
    const uint32_t xa = (j < 2) ? x0[2 * j] : x1[2 * j - 4];
    const uint32_t xb = (j < 2) ? x0[2 * j + 1] : x1[2 * j - 3];
    c = __builtin_amdgcn_fdot2_f32_bf16(lo, __builtin_bit_cast(bf16x2_t, xa), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(hi, __builtin_bit_cast(bf16x2_t, xb), c, false);
  }
  return c;
}

template <int NR, int U>
__device__ __forceinline__ void gemv_fma_f8(float (&acc)[NR], const u32x4 (&w)[NR][U],
                                            const u32x4* xs, int g, int lane, int KC) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = lane + 64 * (g * U + u);
    u32x4 x0 = {0u, 0u, 0u, 0u}, x1 = {0u, 0u, 0u, 0u};
    if (64 * (g * U + U) <= KC || c < KC) { x0 = xs[2 * c]; x1 = xs[2 * c + 1]; }
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = dot16_f8(w[r][u], x0, x1, acc[r]);
  }
}

// Epilogue of one output unit (one lane per wave): `a0` is the fp32 dot product of the unit's row (paired epilogues:
// a0 / a1 = the two rows of the pair: RoPE partners i, i+64 or gate, up).  Same HF rounding points as the reference.
template <int EPI, bool F8>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, int u, float a0, float a1) {
    if (F8) {  // per-output-channel power-of-two scale (exact)
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
      a.y[u] = f2bf(a0);
    } else if (EPI == EPI_RESID) {
      // HF: hidden = residual + proj(x); proj output is a bf16 tensor
      a.y[u] = f2bf(bf2f(a.y[u]) + rbf(a0));
    } else if (EPI == EPI_LOGITS) {
      a.logits[u] = rbf(a0);  // lm_head output is bf16, then .float()
    } else if (EPI == EPI_SWIGLU) {
      const float gte = rbf(a0);
      const float up = rbf(a1);
      const float sl = rbf(gte / (1.f + expf(-gte)));
      a.y[u] = f2bf(sl * up);
    } else if (EPI == EPI_QKV) {
      const int hb = u >> 6, i = u & 63;
      const int sec = hb < a.H ? 0 : (hb < a.H + a.KVH ? 1 : 2);
      const int head = sec == 0 ? hb : (sec == 1 ? hb - a.H : hb - a.H - a.KVH);
      const int pos = a.st->pos;
      const float x1 = rbf(a0);      // dim i
      const float x2 = rbf(a1);  // dim i + 64
      if (sec == 2) {
        bf16_t* dst = a.vcache + ((size_t)head * a.T_max + pos) * 128;
        dst[i] = f2bf(x1);
        dst[i + 64] = f2bf(x2);
      } else {
        // HF apply_rotary_pos_emb: q*cos + rotate_half(q)*sin, every product a bf16 tensor
        const float c = bf2f(a.rope_cos[(size_t)pos * 64 + i]);
        const float s = bf2f(a.rope_sin[(size_t)pos * 64 + i]);
        const float o1 = rbf(rbf(x1 * c) + rbf(-x2 * s));
        const float o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
        bf16_t* dst = (sec == 0) ? (a.q_out + head * 128)
                                 : (a.kcache + ((size_t)head * a.T_max + pos) * 128);
        dst[i] = f2bf(o1);
        dst[i + 64] = f2bf(o2);
      }
    }
}

// R = output units per wave-chunk; paired epilogues (QKV, SWIGLU) stream 2 rows per unit.
// WAVES = waves per block.  PERSIST: grid-stride over chunks (chunk c -> block c % grid,
// wave (c / grid) % WAVES) so a grid sized to the machine covers any N with <= 1 chunk of
// imbalance per wave; otherwise one chunk per wave and the grid covers N.
template <int PRO, int EPI, int R, int U, int WAVES, bool PERSIST, bool F8 = false>
__global__ __launch_bounds__(WAVES * 64) void k_gemv(GemvArgs a) {
  constexpr bool PAIRED = (EPI == EPI_QKV) || (EPI == EPI_SWIGLU);
  constexpr int NR = PAIRED ? 2 * R : R;
  constexpr int THREADS = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);
  const int K8 = a.K >> 3;                     // 16-byte chunks of x (bf16)
  const int KC = F8 ? (a.K >> 4) : K8;         // 16-byte chunks of one weight row
  const size_t row_bytes = F8 ? (size_t)a.K : (size_t)a.K * 2;
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(F8 ? (const void*)a.W8 : (const void*)a.W);
  float* red = reinterpret_cast<float*>(smem + (size_t)K8 * 16);  // WAVES floats of scratch

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
  float acc[NR];

  // first stage of weights goes in flight before the prologue touches x
  gemv_load<NR, U>(wa, rows, 0, lane, KC);

  // ---- prologue: build the bf16 input vector in LDS
  if (PRO == PRO_COPY) {
    const u32x4* x4 = reinterpret_cast<const u32x4*>(a.x);
    for (int c = tid; c < K8; c += THREADS) xs[c] = x4[c];
  } else if (PRO == PRO_RMSNORM) {
    const u32x4* x4 = reinterpret_cast<const u32x4*>(a.x);
    const u32x4* w4 = reinterpret_cast<const u32x4*>(a.norm_w);
    float ss = 0.f;
    for (int c = tid; c < K8; c += THREADS) {
      const u32x4 v = x4[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = pk_lo(v[e]), hi = pk_hi(v[e]);
        ss += lo * lo;
        ss += hi * hi;
      }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) tot += red[w];
    const float inv = rsqrtf(tot / (float)a.K + a.eps);
    for (int c = tid; c < K8; c += THREADS) {
      const u32x4 v = x4[c];
      const u32x4 g = w4[c];
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // HF LlamaRMSNorm: weight * (x * rsqrt(var+eps)).to(bf16)
        const float nlo = rbf(pk_lo(v[e]) * inv), nhi = rbf(pk_hi(v[e]) * inv);
        o[e] = pack2(pk_lo(g[e]) * nlo, pk_hi(g[e]) * nhi);
      }
      xs[c] = o;
    }
  } else {  // PRO_ATTN: reduce the S split-K partials of every head (flash-decode combine)
    const int S = a.S;
    for (int c = tid; c < K8; c += THREADS) {
      const int head = c >> 4;       // 16 chunks of 8 dims per 128-dim head
      const int d0 = (c & 15) * 8;
      float M = -1e30f;
      for (int s = 0; s < S; ++s) M = fmaxf(M, a.pm[head * S + s]);
      float L = 0.f;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
      for (int s = 0; s < S; ++s) {
        const float w = __expf(a.pm[head * S + s] - M);
        L += w * a.pl[head * S + s];
        const f32x4* po4 =
            reinterpret_cast<const f32x4*>(a.po + ((size_t)(head * S + s)) * 128 + d0);
        const f32x4 p0 = po4[0], p1 = po4[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] += w * p0[e];
          o[4 + e] += w * p1[e];
        }
      }
      const float invL = 1.f / L;
      u32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = pack2(o[2 * e] * invL, o[2 * e + 1] * invL);
      xs[c] = ov;
    }
  }
  __syncthreads();

  for (;;) {
    const bool active = unit0 < n_units;  // wave-uniform
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
    // ---- main loop: two register stages (wa holds stage 0 on entry)
    for (int g = 0; g < G; g += 2) {
      if (g + 1 < G) gemv_load<NR, U>(wb, rows, g + 1, lane, KC);
      if (F8) gemv_fma_f8<NR, U>(acc, wa, xs, g, lane, KC); else gemv_fma<NR, U>(acc, wa, xs, g, lane, KC);
      if (g + 1 < G) {
        if (g + 2 < G) gemv_load<NR, U>(wa, rows, g + 2, lane, KC);
        if (F8) gemv_fma_f8<NR, U>(acc, wb, xs, g + 1, lane, KC); else gemv_fma<NR, U>(acc, wb, xs, g + 1, lane, KC);
      }
    }
    const int cur = unit0;
    if (PERSIST) {  // next chunk's first stage goes in flight before this chunk's reduction
      chunk += chunk_stride;
      unit0 = chunk * R;
      if (unit0 < n_units) {
        set_rows(unit0);
        gemv_load<NR, U>(wa, rows, 0, lane, KC);
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);

    // ---- epilogue (one lane per wave; a handful of scalars)
    if (active && lane == 0) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int u = cur + j;
        if (u >= n_units) break;
        gemv_epilogue<EPI, F8>(a, u, PAIRED ? acc[2 * j] : acc[j], PAIRED ? acc[2 * j + 1] : 0.f);
      }
    }
    if (!PERSIST || unit0 >= n_units) break;
  }
}

// number of CUs of the current device (cached) — persistent grids are sized from it
static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int PRO, int EPI, int R, int U, int WAVES, bool PERSIST>
static void launch_gemv_t(const GemvArgs& a, hipStream_t s, int blocks_per_cu) {
  int n_units = (EPI == EPI_QKV) ? (a.N >> 1) : (EPI == EPI_SWIGLU ? a.ff : a.N);
  const int per_block = WAVES * R;
  int grid = (n_units + per_block - 1) / per_block;
  if (PERSIST) {
    const int cap = num_cus() * (blocks_per_cu > 0 ? blocks_per_cu : 2);
    if (grid > cap) grid = cap;
  }
  const size_t lds = (size_t)(a.K >> 3) * 16 + 64;
  hipLaunchKernelGGL((k_gemv<PRO, EPI, R, U, WAVES, PERSIST>), dim3(grid), dim3(WAVES * 64), lds, s, a);
}

// Tuning table: variant -> instantiation.  Variant 0 is the product default for each role; the
// others exist for the in-situ microbenchmark (dtk_bench_gemv) that picked the default.
#define GV(PRO, EPI, R, U, W, P, BPC) return launch_gemv_t<PRO, EPI, R, U, W, P>(a, s, BPC)
void launch_gemv_variant(int pro, int epi, int variant, const GemvArgs& a, hipStream_t s) {
  if (pro == PRO_RMSNORM && epi == EPI_QKV) {
    switch (variant) {
      default: GV(PRO_RMSNORM, EPI_QKV, 1, 2, 4, false, 0);
      case 8: GV(PRO_RMSNORM, EPI_QKV, 2, 2, 4, false, 0);
      case 9: GV(PRO_RMSNORM, EPI_QKV, 1, 1, 4, false, 0);
      case 10: GV(PRO_RMSNORM, EPI_QKV, 1, 2, 8, false, 0);
      case 11: GV(PRO_RMSNORM, EPI_QKV, 1, 2, 2, false, 0);
      case 1: GV(PRO_RMSNORM, EPI_QKV, 1, 4, 4, false, 0);
      case 2: GV(PRO_RMSNORM, EPI_QKV, 2, 2, 8, false, 0);
      case 3: GV(PRO_RMSNORM, EPI_QKV, 2, 2, 4, true, 4);
      case 4: GV(PRO_RMSNORM, EPI_QKV, 2, 2, 8, true, 2);
      case 5: GV(PRO_RMSNORM, EPI_QKV, 1, 4, 8, true, 2);
      case 6: GV(PRO_RMSNORM, EPI_QKV, 4, 1, 4, false, 0);
      case 7: GV(PRO_RMSNORM, EPI_QKV, 1, 2, 4, false, 0);
    }
  }
  if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) {
    switch (variant) {
      default: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 4, false, 0);
      case 8: GV(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 4, false, 0);
      case 9: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 1, 4, false, 0);
      case 10: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 8, false, 0);
      case 11: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 2, false, 0);
      case 1: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 4, 4, false, 0);
      case 2: GV(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 8, false, 0);
      case 3: GV(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 4, true, 4);
      case 4: GV(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 8, true, 2);
      case 5: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 4, 8, true, 2);
      case 6: GV(PRO_RMSNORM, EPI_SWIGLU, 4, 1, 4, false, 0);
      case 7: GV(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 4, false, 0);
    }
  }
  if (pro == PRO_COPY && epi == EPI_RESID) {
    switch (variant) {
      default: GV(PRO_COPY, EPI_RESID, 1, 8, 4, false, 0);
      case 8: GV(PRO_COPY, EPI_RESID, 2, 4, 4, false, 0);
      case 9: GV(PRO_COPY, EPI_RESID, 1, 2, 4, false, 0);
      case 10: GV(PRO_COPY, EPI_RESID, 1, 8, 8, false, 0);
      case 11: GV(PRO_COPY, EPI_RESID, 1, 8, 2, false, 0);
      case 12: GV(PRO_COPY, EPI_RESID, 1, 4, 2, false, 0);
      case 1: GV(PRO_COPY, EPI_RESID, 1, 4, 4, false, 0);
      case 2: GV(PRO_COPY, EPI_RESID, 4, 2, 4, false, 0);
      case 3: GV(PRO_COPY, EPI_RESID, 2, 4, 8, false, 0);
      case 4: GV(PRO_COPY, EPI_RESID, 2, 4, 4, true, 4);
      case 5: GV(PRO_COPY, EPI_RESID, 2, 4, 8, true, 2);
      case 6: GV(PRO_COPY, EPI_RESID, 1, 8, 4, false, 0);
      case 7: GV(PRO_COPY, EPI_RESID, 2, 2, 4, false, 0);
    }
  }
  if (pro == PRO_ATTN && epi == EPI_RESID) GV(PRO_ATTN, EPI_RESID, 2, 4, 4, false, 0);
  if (pro == PRO_RMSNORM && epi == EPI_LOGITS) {
    switch (variant) {
      default: GV(PRO_RMSNORM, EPI_LOGITS, 2, 4, 4, false, 0);
      case 1: GV(PRO_RMSNORM, EPI_LOGITS, 4, 2, 8, true, 2);
      case 2: GV(PRO_RMSNORM, EPI_LOGITS, 4, 2, 4, false, 0);
      case 3: GV(PRO_RMSNORM, EPI_LOGITS, 1, 4, 4, false, 0);
      case 4: GV(PRO_RMSNORM, EPI_LOGITS, 1, 8, 4, false, 0);
    }
  }
  if (pro == PRO_RMSNORM && epi == EPI_STORE) {
    if (variant == 20) GV(PRO_RMSNORM, EPI_STORE, 1, 2, 4, false, 0);   // probe: the streaming core + norm prologue only
    GV(PRO_RMSNORM, EPI_STORE, 4, 2, 4, false, 0);
  }
  if (variant == 20) GV(PRO_COPY, EPI_STORE, 1, 2, 4, false, 0);         // probes (dtk_bench_gemv | 0x200): streaming core,
  if (variant == 21) GV(PRO_COPY, EPI_STORE, 1, 8, 4, false, 0);         // plain x copy, plain store
  GV(PRO_COPY, EPI_STORE, 4, 2, 4, false, 0);
}
#undef GV

template <int PRO, int EPI, int R, int U, int WAVES, bool PERSIST = false, int BPC = 4>
static void launch_gemv_f8_t(const GemvArgs& a, hipStream_t s) {
  int n_units = (EPI == EPI_QKV) ? (a.N >> 1) : (EPI == EPI_SWIGLU ? a.ff : a.N);
  const int per_block = WAVES * R;
  int grid = (n_units + per_block - 1) / per_block;
  if (PERSIST && grid > num_cus() * BPC) grid = num_cus() * BPC;
  const size_t lds = (size_t)(a.K >> 3) * 16 + 64;
  hipLaunchKernelGGL((k_gemv<PRO, EPI, R, U, WAVES, PERSIST, true>), dim3(grid), dim3(WAVES * 64), lds, s, a);
}
// fp8-weight decode GEMVs (requires K % 16 == 0); same roles as the bf16 defaults.  An fp8 row is half
// the bytes, so more rows per wave (R=2) amortise the prologue / reduction; DTK_F8_VARIANT sweeps the choice.
static int f8_variant() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DTK_F8_VARIANT"); v = e ? atoi(e) : 8; }
  return v;
}
#define F8(PRO, EPI, R, U, W) return launch_gemv_f8_t<PRO, EPI, R, U, W>(a, s)
#define F8P(PRO, EPI, R, U, W, BPC) return launch_gemv_f8_t<PRO, EPI, R, U, W, true, BPC>(a, s)
void launch_gemv_f8(int pro, int epi, const GemvArgs& a, hipStream_t s) {
  const int v = f8_variant();
  if (pro == PRO_RMSNORM && (epi == EPI_QKV || epi == EPI_SWIGLU)) {
    if (epi == EPI_QKV) {
      switch (v) { default: F8(PRO_RMSNORM, EPI_QKV, 2, 2, 4); case 0: F8(PRO_RMSNORM, EPI_QKV, 1, 2, 4);
                   case 2: F8(PRO_RMSNORM, EPI_QKV, 2, 4, 4); case 3: F8(PRO_RMSNORM, EPI_QKV, 4, 2, 4);
                   case 4: F8(PRO_RMSNORM, EPI_QKV, 2, 2, 8); case 5: F8(PRO_RMSNORM, EPI_QKV, 1, 4, 4);
                   case 6: F8P(PRO_RMSNORM, EPI_QKV, 1, 2, 4, 4); case 7: F8P(PRO_RMSNORM, EPI_QKV, 2, 2, 4, 4);
                   case 8: F8P(PRO_RMSNORM, EPI_QKV, 1, 2, 8, 2); case 9: F8P(PRO_RMSNORM, EPI_QKV, 1, 4, 4, 4); }
    }
    switch (v) { default: F8(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 4); case 0: F8(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 4);
                 case 2: F8(PRO_RMSNORM, EPI_SWIGLU, 2, 4, 4); case 3: F8(PRO_RMSNORM, EPI_SWIGLU, 4, 2, 4);
                 case 4: F8(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 8); case 5: F8(PRO_RMSNORM, EPI_SWIGLU, 1, 4, 4);
                 case 6: F8P(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 4, 4); case 7: F8P(PRO_RMSNORM, EPI_SWIGLU, 2, 2, 4, 4);
                 case 8: F8P(PRO_RMSNORM, EPI_SWIGLU, 1, 2, 8, 2); case 9: F8P(PRO_RMSNORM, EPI_SWIGLU, 1, 4, 4, 4); }
  }
  if (pro == PRO_RMSNORM && epi == EPI_LOGITS) {
    switch (v) { default: F8(PRO_RMSNORM, EPI_LOGITS, 4, 2, 4); case 0: F8(PRO_RMSNORM, EPI_LOGITS, 2, 2, 4);
                 case 2: F8(PRO_RMSNORM, EPI_LOGITS, 4, 4, 4); case 3: F8(PRO_RMSNORM, EPI_LOGITS, 8, 2, 4); }
  }
  if (pro == PRO_ATTN && epi == EPI_RESID) F8(PRO_ATTN, EPI_RESID, 2, 2, 4);
  switch (v) { default: F8(PRO_COPY, EPI_RESID, 2, 4, 4); case 0: F8(PRO_COPY, EPI_RESID, 1, 4, 4);
               case 2: F8(PRO_COPY, EPI_RESID, 2, 8, 4); case 3: F8(PRO_COPY, EPI_RESID, 4, 4, 4);
               case 4: F8(PRO_COPY, EPI_RESID, 2, 4, 8); case 5: F8(PRO_COPY, EPI_RESID, 1, 8, 4);
               case 6: F8P(PRO_COPY, EPI_RESID, 1, 4, 4, 4); case 7: F8P(PRO_COPY, EPI_RESID, 2, 4, 4, 4);
               case 8: F8P(PRO_COPY, EPI_RESID, 1, 4, 8, 2); case 9: F8P(PRO_COPY, EPI_RESID, 1, 8, 4, 4); }
}
#undef F8
#undef F8P

static int g_variant[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // per-epilogue default variant (tuned)
void set_gemv_default_variant(int epi, int variant) { if (epi >= 0 && epi < 8) g_variant[epi] = variant; }

void launch_gemv(int pro, int epi, const GemvArgs& a, hipStream_t s) {
  if (a.W8) return launch_gemv_f8(pro, epi, a, s);
  int v = g_variant[epi & 7];
  // d <= 2048 (ds-1.3b / tl-1.1b): the per-layer kernels stream only 8-45 MB each, so fewer, fatter waves win
  // (tools/tune_gemv.py on ds-1.3b: gate/up persistent 8-wave 9.35 us vs 10.44; decode 1049 -> 1100 tok/s)
  if (v == 0 && a.d > 0 && a.d <= 2048) {
    if (epi == EPI_SWIGLU) v = 5;        // (R 1, U 4, 8 waves, persistent, 2 blocks per CU)
    else if (epi == EPI_QKV) v = 9;      // (R 1, U 1, 4 waves)
    else if (epi == EPI_RESID) v = 10;   // (R 1, U 8, 8 waves)
  }
  launch_gemv_variant(pro, epi, v, a, s);
}

// ------------------------------------------------------------------------------------------
// Decode attention, split-K ("flash-decode"): grid (H, S).  Block (h, s) scans keys
// [s*chunk, (s+1)*chunk) of head h: 16 lanes share one 256-byte K/V row (16 B each), so a
// wave covers 4 rows per load instruction and a block 16 rows; every 16-lane group keeps
// its own online-softmax stream (m, l, o[8 dims]); streams merge through shuffles + LDS.
// Scores and probabilities stay fp32 (fused-attention semantics: one bf16 rounding of the
// head output, done by the consumer's PRO_ATTN prologue).
__global__ __launch_bounds__(256) void k_attn_decode(AttnDecArgs a) {
  const int h = blockIdx.x, sp = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15;   // which 16-byte piece of the row
  const int grp = lane >> 4;   // which of the wave's 4 rows
  const int n = a.st->pos + 1; // keys 0..pos (this step's k/v were appended by the QKV kernel)
  int chunk = (n + a.S - 1) / a.S;
  chunk = (chunk + 15) & ~15;
  const int j_begin = sp * chunk;
  const int j_end = min(n, j_begin + chunk);

  // q piece of this lane: dims sub*8 .. sub*8+7 as packed bf16
  const u32x4 qv = reinterpret_cast<const u32x4*>(a.q + h * 128)[sub];
  const int kvh = h / a.G;   // GQA: G query heads share one kv head
  const bf16_t* kbase = a.kcache + (size_t)kvh * a.T_max * 128;
  const bf16_t* vbase = a.vcache + (size_t)kvh * a.T_max * 128;

  float m = -1e30f, l = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;

  // 64 rows per block iteration: every 16-lane group has 4 K rows + 4 V rows (8 x 16 B per lane)
  // in flight before the first dot product, so a ~500-token context is one memory round trip
  for (int j0 = j_begin; j0 < j_end; j0 += 64) {
    u32x4 kv[4], vv[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + i * 16 + wave * 4 + grp;
      ok[i] = j < j_end;
      const int jj = ok[i] ? j : j_begin;  // clamp: always a valid row
      kv[i] = reinterpret_cast<const u32x4*>(kbase + (size_t)jj * 128)[sub];
      vv[i] = reinterpret_cast<const u32x4*>(vbase + (size_t)jj * 128)[sub];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = dot8(qv, kv[i], 0.f);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      s *= a.scale;
      if (ok[i]) {
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn);
        const float p = __expf(s - mn);
        l = l * corr + p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[2 * e] = o[2 * e] * corr + p * pk_lo(vv[i][e]);
          o[2 * e + 1] = o[2 * e + 1] * corr + p * pk_hi(vv[i][e]);
        }
        m = mn;
      }
    }
  }
  // merge the 4 row-groups of the wave (lanes with equal sub)
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64);
    const float l2 = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, m2);
    const float c1 = __expf(m - mn), c2 = __expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o2 = __shfl_xor(o[e], off, 64);
      o[e] = o[e] * c1 + o2 * c2;
    }
    m = mn;
  }
  // merge the 4 waves through LDS
  __shared__ float sm_m[4][16], sm_l[4][16], sm_o[4][16][8];
  if (grp == 0) {
    sm_m[wave][sub] = m;
    sm_l[wave][sub] = l;
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_o[wave][sub][e] = o[e];
  }
  __syncthreads();
  if (tid < 16) {
    float M = sm_m[0][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, sm_m[w][tid]);
    float L = 0.f;
    float oo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) oo[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float c = __expf(sm_m[w][tid] - M);
      L += c * sm_l[w][tid];
#pragma unroll
      for (int e = 0; e < 8; ++e) oo[e] += c * sm_o[w][tid][e];
    }
    const size_t slot = (size_t)h * a.S + sp;
    if (a.combine != 1) {  // partials for k_attn_combine / the consumer-side combine (k_gemv<PRO_ATTN>)
      float* dst = a.po + slot * 128 + tid * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = oo[e];
      if (tid == 0) {
        a.pm[slot] = M;
        a.pl[slot] = L;
      }
      return;
    }
    // ---- in-kernel combine by the last-arriving split of this head (placement independent):
    // partials are published with 8-byte agent-scope (write-through, sc1) stores, drained, then
    // one relaxed agent-scope ticket; the block that draws S-1 reads them back with agent-scope
    // loads (guide: "8-B agent atomics both sides"), normalises, rounds once to bf16.
    typedef unsigned long long u64;
    u64* part = reinterpret_cast<u64*>(a.po) + slot * 65;  // 64 x {o,o} + {m,l}
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u64 v = (u64)__float_as_uint(oo[2 * e]) | ((u64)__float_as_uint(oo[2 * e + 1]) << 32);
      __hip_atomic_store(part + tid * 4 + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
      const u64 v = (u64)__float_as_uint(M) | ((u64)__float_as_uint(L) << 32);
      __hip_atomic_store(part + 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing lane is in this wave
    unsigned ticket = 0;
    if (tid == 0) ticket = __hip_atomic_fetch_add(a.counters + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0, 64);
    if (ticket != (unsigned)(a.S - 1)) return;
    const u64* base = reinterpret_cast<u64*>(a.po) + (size_t)h * a.S * 65;
    float Mg = -1e30f;
    for (int s = 0; s < a.S; ++s) {
      const u64 v = __hip_atomic_load(base + (size_t)s * 65 + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Mg = fmaxf(Mg, __uint_as_float((unsigned)v));
    }
    float Lg = 0.f;
    float og[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) og[e] = 0.f;
    for (int s = 0; s < a.S; ++s) {
      const u64 mlv = __hip_atomic_load(base + (size_t)s * 65 + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float w = __expf(__uint_as_float((unsigned)mlv) - Mg);
      Lg += w * __uint_as_float((unsigned)(mlv >> 32));
      const u64* src = base + (size_t)s * 65 + tid * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u64 v = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        og[2 * e] += w * __uint_as_float((unsigned)v);
        og[2 * e + 1] += w * __uint_as_float((unsigned)(v >> 32));
      }
    }
    const float invL = 1.f / Lg;
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack2(og[2 * e] * invL, og[2 * e + 1] * invL);
    reinterpret_cast<u32x4*>(a.out + h * 128)[tid] = ov;
    if (tid == 0) __hip_atomic_store(a.counters + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Reduction of the S split-K partials of every head: one 128-thread block per head, thread = dim.
// A kernel boundary (not an in-launch hand-off) publishes the partials: measured cheaper than both
// the consumer-side combine in o_proj's prologue (133 KB re-read by every block) and a
// last-arriver combine inside k_attn_decode (agent-scope stores + ticket + serialized loads).
__global__ __launch_bounds__(128) void k_attn_combine(AttnDecArgs a) {
  const int h = blockIdx.x, t = threadIdx.x;
  const int S = a.S;
  // all loads first (one memory round trip), then the arithmetic
  float pm[16], pl[16], po[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const bool ok = s < S;
    const int ss = ok ? s : 0;
    pm[s] = ok ? a.pm[h * S + ss] : -1e30f;
    pl[s] = ok ? a.pl[h * S + ss] : 0.f;
    po[s] = ok ? a.po[((size_t)(h * S + ss)) * 128 + t] : 0.f;
  }
  float M = -1e30f;
#pragma unroll
  for (int s = 0; s < 16; ++s) M = fmaxf(M, pm[s]);
  float L = 0.f, o = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float w = __expf(pm[s] - M);
    L += w * pl[s];
    o += w * po[s];
  }
  a.out[h * 128 + t] = f2bf(o / L);
}

// Short-context variant: ONE 1024-thread block per head scans all keys (16 waves x 4 row-groups, 4 rows
// each in flight = 256 rows per iteration), merges its 64 online-softmax streams through shuffles + LDS
// and writes the normalised bf16 head output itself: no split-K partials, no combine kernel (2 launches
// and ~4 us per layer less below ~1k tokens of context; above that the split-K grid wins on bandwidth).
__global__ __launch_bounds__(1024) void k_attn_decode_head(AttnDecArgs a) {
  const int h = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, grp = lane >> 4;
  const int n = a.st->pos + 1;
  const u32x4 qv = reinterpret_cast<const u32x4*>(a.q + h * 128)[sub];
  const int kvh = h / a.G;   // GQA: G query heads share one kv head
  const bf16_t* kbase = a.kcache + (size_t)kvh * a.T_max * 128;
  const bf16_t* vbase = a.vcache + (size_t)kvh * a.T_max * 128;
  float m = -1e30f, l = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j0 = 0; j0 < n; j0 += 256) {
    u32x4 kv[4], vv[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + i * 64 + wave * 4 + grp;
      ok[i] = j < n;
      const int jj = ok[i] ? j : 0;
      kv[i] = reinterpret_cast<const u32x4*>(kbase + (size_t)jj * 128)[sub];
      vv[i] = reinterpret_cast<const u32x4*>(vbase + (size_t)jj * 128)[sub];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = dot8(qv, kv[i], 0.f);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      s *= a.scale;
      if (ok[i]) {
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn);
        const float p = __expf(s - mn);
        l = l * corr + p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[2 * e] = o[2 * e] * corr + p * pk_lo(vv[i][e]);
          o[2 * e + 1] = o[2 * e + 1] * corr + p * pk_hi(vv[i][e]);
        }
        m = mn;
      }
    }
  }
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64);
    const float l2 = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, m2);
    const float c1 = __expf(m - mn), c2 = __expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o2 = __shfl_xor(o[e], off, 64);
      o[e] = o[e] * c1 + o2 * c2;
    }
    m = mn;
  }
  __shared__ float sm_m[16][16], sm_l[16][16], sm_o[16][16][8];
  if (grp == 0) {
    sm_m[wave][sub] = m;
    sm_l[wave][sub] = l;
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_o[wave][sub][e] = o[e];
  }
  __syncthreads();
  if (tid < 16) {
    float M = sm_m[0][tid];
#pragma unroll
    for (int w = 1; w < 16; ++w) M = fmaxf(M, sm_m[w][tid]);
    float L = 0.f;
    float oo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) oo[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const float c = __expf(sm_m[w][tid] - M);
      L += c * sm_l[w][tid];
#pragma unroll
      for (int e = 0; e < 8; ++e) oo[e] += c * sm_o[w][tid][e];
    }
    const float invL = 1.f / L;
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack2(oo[2 * e] * invL, oo[2 * e + 1] * invL);
    reinterpret_cast<u32x4*>(a.out + h * 128)[tid] = ov;
  }
}

void launch_attn_decode(const AttnDecArgs& a, hipStream_t s) {
  if (a.combine == 3) {  // one block per head, output written directly
    hipLaunchKernelGGL(k_attn_decode_head, dim3(a.H), dim3(1024), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(k_attn_decode, dim3(a.H, a.S), dim3(256), 0, s, a);
  if (a.combine == 2) hipLaunchKernelGGL(k_attn_combine, dim3(a.H), dim3(128), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// Sampler: HF logits processors + argmax | multinomial for ONE sequence, one 1024-thread
// block (the logits are 126-513 KB and L2 resident).  Mirrors generation/utils.py _sample:
// NoBadWords(-inf) -> SuppressTokensAtBegin(-inf on the first generated token) ->
// Temperature -> TopK -> TopP -> softmax -> draw.  Top-k/top-p thresholds come from a
// 4x8-bit radix descent over order-preserving keys with fixed-point (integer) probability
// mass, so the kept set and the inverse-CDF draw are deterministic and reproducible by
// oracle/sampling.py.  The block then advances DecState and gathers the token embedding
// into the residual stream (first op of the next forward).
#define SAMPLE_THREADS 1024

__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ bool is_banned(const SamplingDev* sp, int i, bool first) {
  for (int k = 0; k < sp->n_bad; ++k)
    if (sp->bad_ids[k] == i) return true;
  for (int k = 0; k < sp->n_always; ++k)
    if (sp->always_ids[k] == i) return true;
  if (first)
    for (int k = 0; k < sp->n_begin; ++k)
      if (sp->begin_ids[k] == i) return true;
  return false;
}

__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample(SampleArgs a) {
  if (a.bs) {  // batched decode: one block per slot, same code on that slot's buffers
    const int slot = blockIdx.x;
    if (!a.bs->active[slot]) return;
    a.logits += (size_t)slot * a.logits_stride;
    a.sp += slot;
    a.st += slot;
    a.x += (size_t)slot * a.d;
    a.tok_ring += (size_t)((unsigned)a.bs->step % (unsigned)a.ring) * DTK_MAX_BATCH + slot;
    a.ring = 1;        // the ring index was applied above
    a.step_override = -1;
  }
  __shared__ float s_f[16];
  __shared__ int s_i[16];
  __shared__ unsigned long long s_q[16];
  __shared__ unsigned long long h_mass[256];
  __shared__ unsigned int h_cnt[256];
  __shared__ unsigned long long sc_above_q;
  __shared__ unsigned int sc_above_c;
  __shared__ unsigned int sc_prefix;
  __shared__ unsigned int sc_bin;
  __shared__ unsigned long long scan[SAMPLE_THREADS];
  __shared__ int s_token;
  __shared__ float s_max;
  __shared__ unsigned long long s_total;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = a.V;
  // the sampling configuration is read once into LDS (is_banned runs for every vocabulary entry)
  __shared__ SamplingDev s_sp;
  if (tid < (int)(sizeof(SamplingDev) / 4)) reinterpret_cast<uint32_t*>(&s_sp)[tid] = reinterpret_cast<const uint32_t*>(a.sp)[tid];
  __syncthreads();
  const SamplingDev* sp = &s_sp;
  const uint32_t draw = (a.step_override >= 0) ? (uint32_t)a.step_override : a.st->draw;
  const bool first = (draw == 0);
  const bool sampling = sp->do_sample != 0;
  const float invT = sampling ? 1.f / sp->temperature : 1.f;

  // ---- pass 1: masked (scaled) max and first argmax.  16-byte loads, four per thread in flight
  // (the greedy path is one memory round trip + the block reduction)
  float best = -INFINITY;
  int besti = 0x7fffffff;
  const bool vec = ((V & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.logits) & 15) == 0);
  if (vec) {
    const int V4 = V >> 2;
    const f32x4* l4 = reinterpret_cast<const f32x4*>(a.logits);
    for (int base = tid; base < V4; base += 4 * SAMPLE_THREADS) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i4 = base + u * SAMPLE_THREADS;
        v[u] = (i4 < V4) ? l4[i4] : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i4 = base + u * SAMPLE_THREADS;
        if (i4 >= V4) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i4 * 4 + e;
          float z = v[u][e] * invT;
          if (is_banned(sp, i, first)) z = -INFINITY;
          if (z > best || (z == best && i < besti)) { best = z; besti = i; }
        }
      }
    }
  } else {
    for (int i = tid; i < V; i += SAMPLE_THREADS) {
      float z = a.logits[i] * invT;
      if (is_banned(sp, i, first)) z = -INFINITY;
      if (z > best || (z == best && i < besti)) { best = z; besti = i; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(besti, off, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) { s_f[wave] = best; s_i[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    float b = s_f[0]; int bi = s_i[0];
    for (int w = 1; w < 16; ++w)
      if (s_f[w] > b || (s_f[w] == b && s_i[w] < bi)) { b = s_f[w]; bi = s_i[w]; }
    s_max = b; s_token = bi;
  }
  __syncthreads();
  const float zmax = s_max;

  if (sampling) {
    // fixed-point mass q_i = floor(exp(z_i - zmax) * 2^31)  (fits 32 bits; exact integer sums)
    auto mass = [&](int i, float& z) -> unsigned long long {
      z = a.logits[i] * invT;
      if (is_banned(sp, i, first)) z = -INFINITY;
      const float e = expf(z - zmax);
      return (unsigned long long)((double)e * 2147483648.0);
    };
    // ---- radix descent for the keep-threshold key.  Top-k then top-p, as HF orders the
    // warpers: top-k keeps keys >= (k-th largest); top-p then works on the softmax of the
    // survivors: keep token iff the mass strictly above it is < top_p * total.
    uint32_t thr_k = 0;  // keep keys >= thr_k
    if (sp->top_k > 0 && sp->top_k < V) {
      uint32_t prefix = 0; unsigned int above = 0;
      for (int level = 3; level >= 0; --level) {
        const int shift = level * 8;
        for (int b = tid; b < 256; b += SAMPLE_THREADS) h_cnt[b] = 0;
        __syncthreads();
        for (int i = tid; i < V; i += SAMPLE_THREADS) {
          float z; (void)mass(i, z);
          const uint32_t key = fkey(z);
          const bool match = (level == 3) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
          if (match) atomicAdd(&h_cnt[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned int ab = above; int bsel = 0;
          for (int b = 255; b >= 0; --b) {
            if (h_cnt[b] == 0) continue;
            if (ab + h_cnt[b] >= (unsigned)sp->top_k) { bsel = b; break; }
            ab += h_cnt[b];
          }
          sc_above_c = ab; sc_bin = bsel;
        }
        __syncthreads();
        above = sc_above_c;
        prefix |= (sc_bin << shift);
        __syncthreads();
      }
      thr_k = prefix;
    }
    // total mass of top-k survivors
    unsigned long long loc = 0;
    for (int i = tid; i < V; i += SAMPLE_THREADS) {
      float z; const unsigned long long q = mass(i, z);
      if (fkey(z) >= thr_k) loc += q;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loc += __shfl_xor(loc, off, 64);
    if (lane == 0) s_q[wave] = loc;
    __syncthreads();
    if (tid == 0) { unsigned long long t = 0; for (int w = 0; w < 16; ++w) t += s_q[w]; s_total = t; }
    __syncthreads();
    const unsigned long long total_k = s_total;

    uint32_t thr = thr_k;
    if (sp->top_p < 1.0f) {
      const unsigned long long pq = (unsigned long long)((double)sp->top_p * (double)total_k);
      uint32_t prefix = 0; unsigned long long above = 0;
      for (int level = 3; level >= 0; --level) {
        const int shift = level * 8;
        for (int b = tid; b < 256; b += SAMPLE_THREADS) { h_mass[b] = 0; h_cnt[b] = 0; }
        __syncthreads();
        for (int i = tid; i < V; i += SAMPLE_THREADS) {
          float z; const unsigned long long q = mass(i, z);
          const uint32_t key = fkey(z);
          if (key < thr_k) continue;
          const bool match = (level == 3) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
          if (match) {
            atomicAdd(&h_mass[(key >> shift) & 255], q);
            atomicAdd(&h_cnt[(key >> shift) & 255], 1u);
          }
        }
        __syncthreads();
        if (tid == 0) {
          // lowest non-empty bin whose strictly-above mass is still < pq
          unsigned long long ab = above; int bsel = -1; unsigned long long ab_sel = above;
          for (int b = 255; b >= 0; --b) {
            if (h_cnt[b] == 0) continue;
            if (ab < pq || bsel < 0) { bsel = b; ab_sel = ab; } else break;
            ab += h_mass[b];
          }
          sc_above_q = ab_sel; sc_bin = (unsigned)bsel;
        }
        __syncthreads();
        above = sc_above_q;
        prefix |= (sc_bin << shift);
        __syncthreads();
      }
      thr = prefix > thr_k ? prefix : thr_k;
    }
    // ---- kept mass + inverse-CDF draw in index order
    const int per = (V + SAMPLE_THREADS - 1) / SAMPLE_THREADS;
    const int i0 = tid * per, i1 = min(V, i0 + per);
    unsigned long long mine = 0;
    for (int i = i0; i < i1; ++i) {
      float z; const unsigned long long q = mass(i, z);
      if (fkey(z) >= thr) mine += q;
    }
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < SAMPLE_THREADS; off <<= 1) {
      unsigned long long v = 0;
      if (tid >= off) v = scan[tid - off];
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const unsigned long long kept = scan[SAMPLE_THREADS - 1];
    const uint64_t r = splitmix64(sp->seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(draw + 1))) >> 32;
    const unsigned long long target = __umul64hi(kept, r << 32);  // floor(kept * r / 2^32)
    const unsigned long long excl = scan[tid] - mine;
    if (mine > 0 && target >= excl && target < excl + mine) {
      unsigned long long run = excl;
      for (int i = i0; i < i1; ++i) {
        float z; const unsigned long long q = mass(i, z);
        if (fkey(z) >= thr) {
          if (target < run + q) { s_token = i; break; }
          run += q;
        }
      }
    }
    if (a.probs_out) {
      for (int i = tid; i < V; i += SAMPLE_THREADS) {
        float z; const unsigned long long q = mass(i, z);
        a.probs_out[i] = (fkey(z) >= thr) ? (float)((double)q / (double)kept) : 0.f;
      }
    }
    __syncthreads();
  } else if (a.probs_out) {
    for (int i = tid; i < V; i += SAMPLE_THREADS) a.probs_out[i] = (i == s_token) ? 1.f : 0.f;
  }

  const int tok = s_token;
  if (tid == 0) {
    a.tok_ring[a.bs ? 0u : draw % (uint32_t)a.ring] = (int64_t)tok;
    if (a.advance) {
      a.st->token = tok;
      a.st->pos = a.st->next_pos;
      a.st->next_pos = a.st->next_pos + 1;
      a.st->draw = draw + 1;
    }
  }
  if (a.advance) {
    // embedding gather: first op of the forward that follows
    const u32x4* src = reinterpret_cast<const u32x4*>(a.embed + (size_t)tok * a.d);
    u32x4* dst = reinterpret_cast<u32x4*>(a.x);
    for (int c = tid; c < (a.d >> 3); c += SAMPLE_THREADS) dst[c] = src[c];
  }
}

// ------------------------------------------------------------------------------------------
// Register-resident sampler for vocabularies up to 32 768 (every v1 model): thread t owns the 32
// CONSECUTIVE logits [32t, 32t+32) — one 128-byte line, read once with 16-byte loads — and keeps their keys
// and integer masses in registers through every pass (argmax, top-k / top-p radix descent, inverse-CDF
// scan), so the whole sampler is ONE pass over the logits.  Histogram updates are run-length
// aggregated per thread (neighbouring logits mostly share the high key byte), which removes the
// same-address LDS-atomic serialisation that dominated k_sample (190 us -> measured in profiles/).
// Same integer semantics as k_sample / oracle/sampling.py (bit-exact kept set and draws).
#define SF_PER 32
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_fast(SampleArgs a) {
  if (a.bs) {
    const int slot = blockIdx.x;
    if (!a.bs->active[slot]) return;
    a.logits += (size_t)slot * a.logits_stride;
    a.sp += slot;
    a.st += slot;
    a.x += (size_t)slot * a.d;
    a.tok_ring += (size_t)((unsigned)a.bs->step % (unsigned)a.ring) * DTK_MAX_BATCH + slot;
    a.ring = 1;
    a.step_override = -1;
  }
  __shared__ SamplingDev s_sp;
  __shared__ float s_f[16];
  __shared__ int s_i[16];
  __shared__ unsigned long long s_q[16];
  __shared__ unsigned long long h_mass[256];
  __shared__ unsigned int h_cnt[256];
  __shared__ unsigned long long sc_above_q;
  __shared__ unsigned int sc_above_c, sc_bin;
  __shared__ unsigned long long scan[SAMPLE_THREADS];
  __shared__ int s_token;
  __shared__ float s_max;
  __shared__ unsigned long long s_total;
  __shared__ unsigned long long s_bw[8];
  __shared__ int s_bi[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = a.V;
  if (tid < (int)(sizeof(SamplingDev) / 4)) reinterpret_cast<uint32_t*>(&s_sp)[tid] = reinterpret_cast<const uint32_t*>(a.sp)[tid];
  __syncthreads();
  const SamplingDev* sp = &s_sp;
  const uint32_t draw = (a.step_override >= 0) ? (uint32_t)a.step_override : a.st->draw;
  const bool first = (draw == 0);
  const bool sampling = sp->do_sample != 0;
  const float invT = sampling ? 1.f / sp->temperature : 1.f;
  const int i0 = tid * SF_PER;

  // ---- the one pass over the logits
  float z[SF_PER];
  const bool vec = ((reinterpret_cast<uintptr_t>(a.logits) & 15) == 0) && (i0 + SF_PER <= V);
  if (vec) {
    const f32x4* l4 = reinterpret_cast<const f32x4*>(a.logits + i0);
#pragma unroll
    for (int k = 0; k < SF_PER / 4; ++k) {
      const f32x4 v = l4[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) z[4 * k + e] = v[e] * invT;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) z[k] = (i0 + k < V) ? a.logits[i0 + k] * invT : -INFINITY;
  }
  // bans: a handful of ids, each owned by exactly one thread (static register index via the unrolled compare)
  {
    const int nb = sp->n_bad, na = sp->n_always, ng = first ? sp->n_begin : 0;
    for (int j = 0; j < nb + na + ng; ++j) {
      const int id = j < nb ? sp->bad_ids[j] : (j < nb + na ? sp->always_ids[j - nb] : sp->begin_ids[j - nb - na]);
      const int rel = id - i0;
      if (rel >= 0 && rel < SF_PER) {
#pragma unroll
        for (int k = 0; k < SF_PER; ++k) if (k == rel) z[k] = -INFINITY;
      }
    }
  }
  float best = -INFINITY;
  int besti = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < SF_PER; ++k)
    if (i0 + k < V && (z[k] > best || (z[k] == best && i0 + k < besti))) { best = z[k]; besti = i0 + k; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(besti, off, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) { s_f[wave] = best; s_i[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    float b = s_f[0]; int bi = s_i[0];
    for (int w = 1; w < 16; ++w)
      if (s_f[w] > b || (s_f[w] == b && s_i[w] < bi)) { b = s_f[w]; bi = s_i[w]; }
    s_max = b; s_token = bi;
  }
  __syncthreads();
  const float zmax = s_max;

  if (sampling) {
    uint32_t q[SF_PER];   // floor(exp(z - zmax) * 2^31) <= 2^31
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) {
      const bool in = i0 + k < V;
      const float zz = z[k];
      q[k] = in ? (uint32_t)((double)expf(zz - zmax) * 2147483648.0) : 0u;
      z[k] = __uint_as_float(in ? fkey(zz) : 0u);   // the slot now holds the order-preserving key
    }
#define key(k) __float_as_uint(z[k])
    // radix descent helper state: run-length aggregated histogram update
    uint32_t thr_k = 0;
    if (sp->top_k > 0 && sp->top_k < V) {
      uint32_t prefix = 0; unsigned int above = 0;
      for (int level = 3; level >= 0; --level) {
        const int shift = level * 8;
        for (int b = tid; b < 256; b += SAMPLE_THREADS) h_cnt[b] = 0;
        __syncthreads();
        uint32_t cur = 0xffffffffu, cc = 0;
#pragma unroll
        for (int k = 0; k < SF_PER; ++k) {
          if (i0 + k >= V) continue;
          const bool match = (level == 3) || ((key(k) >> (shift + 8)) == (prefix >> (shift + 8)));
          if (!match) continue;
          const uint32_t bin = (key(k) >> shift) & 255u;
          if (bin != cur) { if (cc) atomicAdd(&h_cnt[cur], cc); cur = bin; cc = 0; }
          ++cc;
        }
        if (cc) atomicAdd(&h_cnt[cur], cc);
        __syncthreads();
        if (tid == 0) {
          unsigned int ab = above; int bsel = 0;
          for (int b = 255; b >= 0; --b) {
            if (h_cnt[b] == 0) continue;
            if (ab + h_cnt[b] >= (unsigned)sp->top_k) { bsel = b; break; }
            ab += h_cnt[b];
          }
          sc_above_c = ab; sc_bin = bsel;
        }
        __syncthreads();
        above = sc_above_c;
        prefix |= (sc_bin << shift);
        __syncthreads();
      }
      thr_k = prefix;
    }
    unsigned long long loc = 0;
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) if (i0 + k < V && key(k) >= thr_k) loc += q[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loc += __shfl_xor(loc, off, 64);
    if (lane == 0) s_q[wave] = loc;
    __syncthreads();
    if (tid == 0) { unsigned long long t = 0; for (int w = 0; w < 16; ++w) t += s_q[w]; s_total = t; }
    __syncthreads();
    const unsigned long long total_k = s_total;

    uint32_t thr = thr_k;
    if (sp->top_p < 1.0f) {
      const unsigned long long pq = (unsigned long long)((double)sp->top_p * (double)total_k);
      uint32_t prefix = 0; unsigned long long above = 0;
      for (int level = 3; level >= 0; --level) {
        const int shift = level * 8;
        for (int b = tid; b < 256; b += SAMPLE_THREADS) { h_mass[b] = 0; h_cnt[b] = 0; }
        __syncthreads();
        uint32_t cur = 0xffffffffu, cc = 0;
        unsigned long long cm = 0;
#pragma unroll
        for (int k = 0; k < SF_PER; ++k) {
          if (i0 + k >= V || key(k) < thr_k) continue;
          const bool match = (level == 3) || ((key(k) >> (shift + 8)) == (prefix >> (shift + 8)));
          if (!match) continue;
          const uint32_t bin = (key(k) >> shift) & 255u;
          if (bin != cur) {
            if (cc) { atomicAdd(&h_mass[cur], cm); atomicAdd(&h_cnt[cur], cc); }
            cur = bin; cc = 0; cm = 0;
          }
          ++cc; cm += q[k];
        }
        if (cc) { atomicAdd(&h_mass[cur], cm); atomicAdd(&h_cnt[cur], cc); }
        __syncthreads();
        unsigned int bsel; unsigned long long ab_sel;
        bin_select_mass(h_mass, h_cnt, pq, above, s_bw, s_bi, bsel, ab_sel);   // parallel form of the serial 255..0 scan
        above = ab_sel;
        prefix |= (bsel << shift);
      }
      thr = prefix > thr_k ? prefix : thr_k;
    }
    // kept mass + inverse-CDF draw in index order (thread t's range is contiguous: a plain block scan)
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) if (i0 + k < V && key(k) >= thr) mine += q[k];
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < SAMPLE_THREADS; off <<= 1) {
      unsigned long long v = 0;
      if (tid >= off) v = scan[tid - off];
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const unsigned long long kept = scan[SAMPLE_THREADS - 1];
    const uint64_t r = splitmix64(sp->seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(draw + 1))) >> 32;
    const unsigned long long target = __umul64hi(kept, r << 32);
    const unsigned long long excl = scan[tid] - mine;
    if (mine > 0 && target >= excl && target < excl + mine) {
      unsigned long long run = excl;
      bool done = false;
#pragma unroll
      for (int k = 0; k < SF_PER; ++k) {
        if (!done && i0 + k < V && key(k) >= thr) {
          if (target < run + q[k]) { s_token = i0 + k; done = true; }
          run += q[k];
        }
      }
    }
    if (a.probs_out) {
#pragma unroll
      for (int k = 0; k < SF_PER; ++k)
        if (i0 + k < V) a.probs_out[i0 + k] = (key(k) >= thr) ? (float)((double)q[k] / (double)kept) : 0.f;
    }
    __syncthreads();
  } else if (a.probs_out) {
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) if (i0 + k < V) a.probs_out[i0 + k] = (i0 + k == s_token) ? 1.f : 0.f;
  }

  const int tok = s_token;
  if (tid == 0) {
    a.tok_ring[a.bs ? 0u : draw % (uint32_t)a.ring] = (int64_t)tok;
    if (a.advance) {
      a.st->token = tok;
      a.st->pos = a.st->next_pos;
      a.st->next_pos = a.st->next_pos + 1;
      a.st->draw = draw + 1;
    }
  }
  if (a.advance) {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.embed + (size_t)tok * a.d);
    u32x4* dst = reinterpret_cast<u32x4*>(a.x);
    for (int c = tid; c < (a.d >> 3); c += SAMPLE_THREADS) dst[c] = src[c];
  }
}
#undef key

static bool sample_fast_ok(const SampleArgs& a) {
  static int force_generic = -1;
  if (force_generic < 0) { const char* e = getenv("DTK_SAMPLER"); force_generic = (e && !strcmp(e, "generic")) ? 1 : 0; }
  return !force_generic && a.V <= SF_PER * SAMPLE_THREADS;
}
void launch_sample(const SampleArgs& a, hipStream_t s) {
  if (sample_fast_ok(a)) hipLaunchKernelGGL(k_sample_fast, dim3(1), dim3(SAMPLE_THREADS), 0, s, a);
  else hipLaunchKernelGGL(k_sample, dim3(1), dim3(SAMPLE_THREADS), 0, s, a);
}
void launch_sample_b(const SampleArgs& a, hipStream_t s) {
  if (sample_fast_ok(a)) { hipLaunchKernelGGL(k_sample_fast, dim3(a.nslots), dim3(SAMPLE_THREADS), 0, s, a); return; }
  hipLaunchKernelGGL(k_sample, dim3(a.nslots), dim3(SAMPLE_THREADS), 0, s, a);
}
