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
#include "gemv_inl.h"

// Epilogue of one output unit (one lane per wave): `a0` is the fp32 dot product of the unit's row (paired epilogues:
// a0 / a1 = the two rows of the pair: RoPE partners i, i+64 or gate, up).  Same HF rounding points as the reference.
// `pre0` / `pre1`: operands the epilogue needs from memory, loaded at kernel start by the lane that runs it (EPI_RESID: the
// residual value y[u]; EPI_QKV: cos / sin of (pos, i)) — a dependent load in the epilogue was ~1 us of exposed latency at
// the very end of every wave; `pos` likewise (read once at kernel start).
template <int EPI, bool F8>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, int u, float a0, float a1, float pre0, float pre1, int pos) {
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
      a.y[u] = f2bf(pre0 + rbf(a0));
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
      const float x1 = rbf(a0);      // dim i
      const float x2 = rbf(a1);  // dim i + 64
      if (sec == 2) {
        bf16_t* dst = a.vcache + ((size_t)head * a.T_max + pos) * 128;
        dst[i] = f2bf(x1);
        dst[i + 64] = f2bf(x2);
      } else {
        // HF apply_rotary_pos_emb: q*cos + rotate_half(q)*sin, every product a bf16 tensor
        const float c = pre0, s = pre1;
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
// KS > 1: split-K over KS waves of the block — the k-groups of a row (pair) are dealt round-robin to KS waves, whose partial
// sums meet in LDS in a fixed order (deterministic).  For the N = d roles (o_proj, down: only d rows) this puts KS times the
// waves, hence loads, in flight per row; the block then owns WAVES / KS row-chunks.
template <int PRO, int EPI, int R, int U, int WAVES, bool PERSIST, bool F8 = false, int KS = 1>
__global__ __launch_bounds__(WAVES * 64) void k_gemv(GemvArgs a) {
  constexpr bool PAIRED = (EPI == EPI_QKV) || (EPI == EPI_SWIGLU);
  constexpr int NR = PAIRED ? 2 * R : R;
  constexpr int THREADS = WAVES * 64;
  constexpr int RW = WAVES / KS;               // row-chunks (of R units) per block
  static_assert(WAVES % KS == 0 && (KS == 1 || !PERSIST), "split-K variants are not persistent");
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
  const int rw = KS > 1 ? wave / KS : wave;    // which row-chunk of the block
  const int ks = KS > 1 ? wave % KS : 0;       // which share of K

  int n_units;
  if (EPI == EPI_QKV) n_units = (a.N >> 1);
  else if (EPI == EPI_SWIGLU) n_units = a.ff;
  else n_units = a.N;
  const int chunk_stride = PERSIST ? (int)gridDim.x * WAVES : 0;
  int chunk = PERSIST ? (int)blockIdx.x + (int)gridDim.x * wave : (int)blockIdx.x * RW + rw;
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
  const int Gall = (iters + U - 1) / U;
  const int G = KS > 1 ? (Gall - ks + KS - 1) / KS : Gall;    // k-groups of this wave: ks, ks + KS, ...
#define GIDX(g) (KS > 1 ? ks + KS * (g) : (g))
  u32x4 wa[NR][U], wb[NR][U];
  float acc[NR];

  // first stage of weights goes in flight before the prologue touches x
  if (G > 0) gemv_load<NR, U>(wa, rows, GIDX(0), lane, KC);

  // operands of the epilogue (see gemv_epilogue): issued now, consumed after the last weight chunk
  float pre0[R], pre1[R];
  int pos = 0;
  if (EPI == EPI_QKV) pos = a.st->pos;
  auto prefetch_epilogue = [&](int u0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      pre0[j] = 0.f; pre1[j] = 0.f;
      if (lane == 0 && ks == 0) {
        int u = u0 + j;
        if (u >= n_units) u = n_units - 1;
        if (EPI == EPI_RESID) pre0[j] = bf2f(a.y[u]);
        if (EPI == EPI_QKV) {
          pre0[j] = bf2f(a.rope_cos[(size_t)pos * 64 + (u & 63)]);
          pre1[j] = bf2f(a.rope_sin[(size_t)pos * 64 + (u & 63)]);
        }
      }
    }
  };
  if (EPI == EPI_RESID || EPI == EPI_QKV) prefetch_epilogue(unit0);

  // ---- prologue: build the bf16 input vector in LDS
  if (PRO == PRO_COPY) {
    const u32x4* x4 = reinterpret_cast<const u32x4*>(a.x);
    for (int c = tid; c < K8; c += THREADS) xs[c] = x4[c];
  } else if (PRO == PRO_RMSNORM) {
    const u32x4* x4 = reinterpret_cast<const u32x4*>(a.x);
    const u32x4* w4 = reinterpret_cast<const u32x4*>(a.norm_w);
    if (K8 <= 2 * THREADS) {
      // x and the norm weight in ONE memory round trip, kept in registers across the block reduction (the two-pass form
      // re-read x after the barrier: a second dependent L1 / L2 trip in front of every RMSNorm GEMV)
      const int c0 = tid, c1 = tid + THREADS;
      const bool h0 = c0 < K8, h1 = c1 < K8;
      const u32x4 zz = {0u, 0u, 0u, 0u};
      const u32x4 v0 = h0 ? x4[c0] : zz, v1 = h1 ? x4[c1] : zz;
      const u32x4 g0 = h0 ? w4[c0] : zz, g1 = h1 ? w4[c1] : zz;
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = pk_lo(v0[e]), hi = pk_hi(v0[e]);
        ss += lo * lo;
        ss += hi * hi;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = pk_lo(v1[e]), hi = pk_hi(v1[e]);
        ss += lo * lo;
        ss += hi * hi;
      }
      ss = wave_sum(ss);
      if (lane == 0) red[wave] = ss;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += red[w];
      const float inv = rsqrtf(tot / (float)a.K + a.eps);
      u32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // HF LlamaRMSNorm: weight * (x * rsqrt(var+eps)).to(bf16)
        o0[e] = pack2(pk_lo(g0[e]) * rbf(pk_lo(v0[e]) * inv), pk_hi(g0[e]) * rbf(pk_hi(v0[e]) * inv));
        o1[e] = pack2(pk_lo(g1[e]) * rbf(pk_lo(v1[e]) * inv), pk_hi(g1[e]) * rbf(pk_hi(v1[e]) * inv));
      }
      if (h0) xs[c0] = o0;
      if (h1) xs[c1] = o1;
    } else {
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
          const float nlo = rbf(pk_lo(v[e]) * inv), nhi = rbf(pk_hi(v[e]) * inv);
          o[e] = pack2(pk_lo(g[e]) * nlo, pk_hi(g[e]) * nhi);
        }
        xs[c] = o;
      }
    }
  } else {  // PRO_ATTN: reduce the S split-K partials of every head (flash-decode combine)
    // Every load of a group of 4 splits is issued before the first use: one L2 round trip per group instead of one per
    // split (a runtime-bounded loop of load -> exp -> fma made the prologue S dependent trips long).  The sums run over
    // s = 0, 1, 2, ... in that order, as in k_attn_combine.
    const int S = a.S;
    for (int c = tid; c < K8; c += THREADS) {
      const int head = c >> 4;       // 16 chunks of 8 dims per 128-dim head
      const int d0 = (c & 15) * 8;
      const float* pmh = a.pm + head * S;
      const float* plh = a.pl + head * S;
      float M = -1e30f;
      for (int s0 = 0; s0 < S; s0 += 4) {
        float m4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m4[j] = (s0 + j < S) ? pmh[s0 + j] : -1e30f;
#pragma unroll
        for (int j = 0; j < 4; ++j) M = fmaxf(M, m4[j]);
      }
      float L = 0.f;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
      for (int s0 = 0; s0 < S; s0 += 4) {
        float m4[4], l4[4];
        f32x4 p0[4], p1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = s0 + j < S;
          const int sj = ok ? s0 + j : s0;
          m4[j] = ok ? pmh[sj] : -1e30f;
          l4[j] = ok ? plh[sj] : 0.f;
          const f32x4* po4 = reinterpret_cast<const f32x4*>(a.po + ((size_t)(head * S + sj)) * 128 + d0);
          p0[j] = po4[0];
          p1[j] = po4[1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (s0 + j < S) {
            const float w = __expf(m4[j] - M);
            L += w * l4[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] += w * p0[j][e];
              o[4 + e] += w * p1[j][e];
            }
          }
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
      if (g + 1 < G) gemv_load<NR, U>(wb, rows, GIDX(g + 1), lane, KC);
      if (F8) gemv_fma_f8<NR, U>(acc, wa, xs, GIDX(g), lane, KC); else gemv_fma<NR, U>(acc, wa, xs, GIDX(g), lane, KC);
      if (g + 1 < G) {
        if (g + 2 < G) gemv_load<NR, U>(wa, rows, GIDX(g + 2), lane, KC);
        if (F8) gemv_fma_f8<NR, U>(acc, wb, xs, GIDX(g + 1), lane, KC); else gemv_fma<NR, U>(acc, wb, xs, GIDX(g + 1), lane, KC);
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
    for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);
    if (KS > 1) {   // partial sums of the KS waves of a row-chunk meet in LDS; wave ks == 0 adds them in a fixed order
      float* part = red + WAVES;                       // [RW][NR][KS]
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) part[(rw * NR + r) * KS + ks] = acc[r];
      }
      __syncthreads();
      if (ks == 0 && lane == 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float t = part[(rw * NR + r) * KS];
#pragma unroll
          for (int k = 1; k < KS; ++k) t += part[(rw * NR + r) * KS + k];
          acc[r] = t;
        }
      }
    }

    // ---- epilogue (one lane per row-chunk; a handful of scalars)
    if (active && lane == 0 && ks == 0) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int u = cur + j;
        if (u >= n_units) break;
        gemv_epilogue<EPI, F8>(a, u, PAIRED ? acc[2 * j] : acc[j], PAIRED ? acc[2 * j + 1] : 0.f, q0[j], q1[j], pos);
      }
    }
    if (!PERSIST || unit0 >= n_units) break;
  }
#undef GIDX
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

template <int PRO, int EPI, int R, int U, int WAVES, bool PERSIST, int KS = 1>
static void launch_gemv_t(const GemvArgs& a, hipStream_t s, int blocks_per_cu) {
  int n_units = (EPI == EPI_QKV) ? (a.N >> 1) : (EPI == EPI_SWIGLU ? a.ff : a.N);
  const int per_block = (WAVES / KS) * R;
  int grid = (n_units + per_block - 1) / per_block;
  if (PERSIST) {
    const int cap = num_cus() * (blocks_per_cu > 0 ? blocks_per_cu : 2);
    if (grid > cap) grid = cap;
  }
  const size_t lds = (size_t)(a.K >> 3) * 16 + 64 + 1024;   // x (bf16) + per-wave RMSNorm partials + split-K partials
  hipLaunchKernelGGL((k_gemv<PRO, EPI, R, U, WAVES, PERSIST, false, KS>), dim3(grid), dim3(WAVES * 64), lds, s, a);
}

// Tuning table: variant -> instantiation.  Variant 0 is the product default for each role; the
// others exist for the in-situ microbenchmark (dtk_bench_gemv) that picked the default.
#define GV(PRO, EPI, R, U, W, P, BPC) return launch_gemv_t<PRO, EPI, R, U, W, P>(a, s, BPC)
#define GVK(PRO, EPI, R, U, W, KS) return launch_gemv_t<PRO, EPI, R, U, W, false, KS>(a, s, 0)
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
      // split-K over the waves of a block (KS waves share a row): twice / four times the loads in flight per row
      case 13: GVK(PRO_COPY, EPI_RESID, 1, 4, 8, 2);    // o_proj (K 4096): one k-group per wave, 4 rows per block
      case 14: GVK(PRO_COPY, EPI_RESID, 2, 4, 8, 2);    //                  8 rows per block
      case 15: GVK(PRO_COPY, EPI_RESID, 1, 6, 8, 2);    // down (K 11008): 22 k-iterations = 4 groups of 6, two per wave
      case 16: GVK(PRO_COPY, EPI_RESID, 2, 6, 8, 2);
      case 17: GVK(PRO_COPY, EPI_RESID, 1, 6, 16, 4);   // one group of 6 iterations per wave, 4 rows per 16-wave block
      case 18: GVK(PRO_COPY, EPI_RESID, 1, 2, 16, 4);   // o_proj: 2 iterations per wave
      case 19: GVK(PRO_COPY, EPI_RESID, 1, 3, 8, 2);    // down, 8 groups of 3 iterations
      case 20: GVK(PRO_COPY, EPI_RESID, 1, 4, 4, 2);    // 2 rows per 4-wave block
      case 21: GVK(PRO_COPY, EPI_RESID, 1, 6, 4, 2);
      case 22: GVK(PRO_COPY, EPI_RESID, 4, 2, 8, 2);
    }
  }
  if (pro == PRO_ATTN && epi == EPI_RESID) {   // o_proj that reduces the attention partials in its prologue: fewer, fatter blocks
    switch (variant) {                          // (every block reads ALL partials: H * S * 130 floats)
      default: GV(PRO_ATTN, EPI_RESID, 2, 4, 8, false, 0);   // 16 rows per block
      case 1: GV(PRO_ATTN, EPI_RESID, 2, 4, 4, false, 0);    // 8 rows (the round-1 shape)
      case 2: GV(PRO_ATTN, EPI_RESID, 1, 8, 8, false, 0);    // 8 rows, one per wave
      case 3: GV(PRO_ATTN, EPI_RESID, 4, 2, 8, false, 0);    // 32 rows
      case 4: GV(PRO_ATTN, EPI_RESID, 4, 2, 4, false, 0);    // 16 rows, 4 waves
      case 5: GVK(PRO_ATTN, EPI_RESID, 2, 4, 8, 2);          // 8 rows, split-K 2
      case 6: GVK(PRO_ATTN, EPI_RESID, 4, 4, 8, 2);          // 16 rows, split-K 2
      case 7: GV(PRO_ATTN, EPI_RESID, 2, 4, 16, false, 0);   // 32 rows, 16 waves
      case 8: GVK(PRO_ATTN, EPI_RESID, 2, 4, 16, 2);         // 16 rows, 16 waves, split-K 2
    }
  }
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
#undef GVK

template <int PRO, int EPI, int R, int U, int WAVES, bool PERSIST = false, int BPC = 4>
static void launch_gemv_f8_t(const GemvArgs& a, hipStream_t s) {
  int n_units = (EPI == EPI_QKV) ? (a.N >> 1) : (EPI == EPI_SWIGLU ? a.ff : a.N);
  const int per_block = WAVES * R;
  int grid = (n_units + per_block - 1) / per_block;
  if (PERSIST && grid > num_cus() * BPC) grid = num_cus() * BPC;
  const size_t lds = (size_t)(a.K >> 3) * 16 + 64 + 1024;
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

// per-role default variant (tuned): indexed by epilogue; slot 5 = o_proj (EPI_RESID with K == d) when it has its own choice
// (-1: same as EPI_RESID), slot 6 = o_proj with the PRO_ATTN prologue
static int g_variant[8] = {0, 0, 0, 0, 0, -1, 0, 0};
void set_gemv_default_variant(int epi, int variant) { if (epi >= 0 && epi < 8) g_variant[epi] = variant; }

void launch_gemv(int pro, int epi, const GemvArgs& a, hipStream_t s) {
  if (a.W8) return launch_gemv_f8(pro, epi, a, s);
  if (pro == PRO_ATTN) {   // 0 = not chosen: 16 rows per 16-wave block with split-K 2 (d > 2048: 387.0 tok/s vs 385.9), 8 rows per block (d <= 2048)
    const int v = g_variant[6] ? g_variant[6] : (a.d > 2048 ? 8 : 1);
    return launch_gemv_variant(pro, epi, v, a, s);
  }
  int v = g_variant[epi & 7];
  const bool o_proj = epi == EPI_RESID && a.K == a.d;
  if (o_proj && g_variant[5] >= 0) v = g_variant[5];
  // Measured defaults (tools/tune_decode.py, profiles/r02_tune_decode_*.log); variant 0 of a role = "not chosen by the caller".
  if (v == 0 && a.d > 0 && a.d <= 2048) {
    // d <= 2048 (ds-1.3b): every kernel streams 8-45 MB and sits on its ~4 us launch + latency floor
    if (epi == EPI_SWIGLU) v = 5;                    // (R 1, U 4, 8 waves, persistent, 2 blocks per CU): 9.8 us vs 10.3
    else if (epi == EPI_RESID) v = o_proj ? 1 : 15;  // o_proj (R 1, U 4, 4 waves) 4.08 us; down split-K 2 (R 1, U 6, 8 waves) 6.88 vs 7.41
    else if (epi == EPI_LOGITS) v = 3;               // (R 1, U 4, 4 waves) 20.0 us vs 21.9
  } else if (v == 0 && a.d > 2048) {
    if (epi == EPI_RESID) v = 10;                    // (R 1, U 8, 8 waves): o_proj 7.76 us, down 17.67; whole step 380 -> 383 tok/s
    else if (epi == EPI_LOGITS) v = 3;               // 37.9 us vs 38.9
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

// Position-independent split scheme (variant 1): the keys are cut into tiles of ROWS = 16 * WAVES rows and tile t belongs to
// split t % S, so a block knows its first tile WITHOUT the position: its K/V loads (and q) go in flight before `pos` has
// been read — one dependent memory trip less on the critical path of a latency-bound kernel — and are masked once it has.
// THREADS = 256 / 512 / 1024 trades blocks for rows per memory round trip (64 / 128 / 256); with S splits one round trip
// covers S * ROWS keys.  S == 1 writes the normalised bf16 head output itself (no partials, no combine).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_attn_decode_t(AttnDecArgs a) {
  constexpr int WAVES = THREADS / 64, ROWS = WAVES * 16;
  const int h = blockIdx.x, sp = blockIdx.y, S = a.S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, grp = lane >> 4;
  const int kvh = h / a.G;
  const bf16_t* kbase = a.kcache + (size_t)kvh * a.T_max * 128;
  const bf16_t* vbase = a.vcache + (size_t)kvh * a.T_max * 128;
  u32x4 kv[4], vv[4];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int j = t * ROWS + i * (ROWS / 4) + wave * 4 + grp;
      j = min(j, a.T_max - 1);        // always a row of the cache; rows at or beyond the context are masked below
      kv[i] = ld_nt(reinterpret_cast<const u32x4*>(kbase + (size_t)j * 128) + sub);      // a K / V row is read once per token by one block:
      vv[i] = ld_nt(reinterpret_cast<const u32x4*>(vbase + (size_t)j * 128) + sub);      // non-temporal, like the weights
    }
  };
  int t = sp;
  load_tile(t);
  const u32x4 qv = reinterpret_cast<const u32x4*>(a.q + h * 128)[sub];
  const int n = a.st->pos + 1;        // keys 0..pos (this step's k/v were appended by the QKV kernel)

  float m = -1e30f, l = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  while (t * ROWS < n) {
    u32x4 kc[4], vc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { kc[i] = kv[i]; vc[i] = vv[i]; }
    const int tcur = t;
    t += S;
    if (t * ROWS < n) load_tile(t);   // contexts beyond S * ROWS keys: next tile in flight under this one's arithmetic
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = tcur * ROWS + i * (ROWS / 4) + wave * 4 + grp;
      float s = dot8(qv, kc[i], 0.f);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      s *= a.scale;
      if (j < n) {
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn);
        const float p = __expf(s - mn);
        l = l * corr + p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[2 * e] = o[2 * e] * corr + p * pk_lo(vc[i][e]);
          o[2 * e + 1] = o[2 * e + 1] * corr + p * pk_hi(vc[i][e]);
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
  __shared__ float sm_m[WAVES][16], sm_l[WAVES][16], sm_o[WAVES][16][8];
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
    for (int w = 1; w < WAVES; ++w) M = fmaxf(M, sm_m[w][tid]);
    float L = 0.f;
    float oo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) oo[e] = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float c = __expf(sm_m[w][tid] - M);
      L += c * sm_l[w][tid];
#pragma unroll
      for (int e = 0; e < 8; ++e) oo[e] += c * sm_o[w][tid][e];
    }
    if (S == 1) {     // the whole head in this block: normalise and round once, no partials
      const float invL = 1.f / L;
      u32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = pack2(oo[2 * e] * invL, oo[2 * e + 1] * invL);
      reinterpret_cast<u32x4*>(a.out + h * 128)[tid] = ov;
      return;
    }
    const size_t slot = (size_t)h * S + sp;   // partials for k_attn_combine / the consumer-side combine (k_gemv<PRO_ATTN>)
    f32x4* dst = reinterpret_cast<f32x4*>(a.po + slot * 128 + tid * 8);
    dst[0] = (f32x4){oo[0], oo[1], oo[2], oo[3]};
    dst[1] = (f32x4){oo[4], oo[5], oo[6], oo[7]};
    if (tid == 0) {
      a.pm[slot] = M;         // a split without a key below the context keeps M = -1e30, L = 0: weight exp(-1e30 - max) = 0
      a.pl[slot] = L;
    }
  }
}

void launch_attn_decode(const AttnDecArgs& a, hipStream_t s) {
  if (a.threads) {   // tile-interleaved splits (k_attn_decode_t); a.combine: 0 consumer reduces the partials, 2 own kernel; S == 1: direct
    const dim3 grid(a.H, a.S);
    if (a.threads >= 1024) hipLaunchKernelGGL(k_attn_decode_t<1024>, grid, dim3(1024), 0, s, a);
    else if (a.threads >= 512) hipLaunchKernelGGL(k_attn_decode_t<512>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_attn_decode_t<256>, grid, dim3(256), 0, s, a);
    if (a.S > 1 && a.combine == 2) hipLaunchKernelGGL(k_attn_combine, dim3(a.H), dim3(128), 0, s, a);
    return;
  }
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
  const int forced = a.advance ? a.st->force_plus1 : 0;
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

  // a resumed slot (dtk_resume_slot) forwards the last token of its prompt instead of a sampled one: the draw counter (and
  // with it the begin-suppress rule of the first SAMPLED token) does not move
  const int tok = forced > 0 ? forced - 1 : s_token;
  if (tid == 0) {
    a.tok_ring[a.bs ? 0u : draw % (uint32_t)a.ring] = (int64_t)tok;
    if (a.advance) {
      a.st->token = tok;
      a.st->pos = a.st->next_pos;
      a.st->next_pos = a.st->next_pos + 1;
      a.st->draw = forced > 0 ? draw : draw + 1;
      if (forced > 0) a.st->force_plus1 = 0;
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
  const int forced = a.advance ? a.st->force_plus1 : 0;
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
    // integer mass floor(exp(z - zmax) * 2^31) <= 2^31 of logit k.  Recomputed from the key wherever it is needed (the key is an
    // invertible image of the logit, so the value is the same every time): holding 32 masses next to 32 keys put the kernel 33
    // VGPRs + 201 SGPRs over the 128-register budget of a 1024-thread block (profiles/r03_kernel_resources.txt: 136 B of scratch)
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) {
      const bool in = i0 + k < V;
      z[k] = __uint_as_float(in ? fkey(z[k]) : 0u);   // the slot now holds the order-preserving key
    }
#define key(k) __float_as_uint(z[k])
    auto mass_of = [&](uint32_t kb) -> uint32_t {     // only called for k with i0 + k < V
      const float zz = (kb & 0x80000000u) ? __uint_as_float(kb & 0x7fffffffu) : __uint_as_float(~kb);
      return (uint32_t)((double)expf(zz - zmax) * 2147483648.0);
    };
#define q_(k) mass_of(key(k))
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
    for (int k = 0; k < SF_PER; ++k) if (i0 + k < V && key(k) >= thr_k) loc += q_(k);
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
          ++cc; cm += q_(k);
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
    for (int k = 0; k < SF_PER; ++k) if (i0 + k < V && key(k) >= thr) mine += q_(k);
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
          const unsigned long long qk = q_(k);
          if (target < run + qk) { s_token = i0 + k; done = true; }
          run += qk;
        }
      }
    }
    if (a.probs_out) {
#pragma unroll
      for (int k = 0; k < SF_PER; ++k)
        if (i0 + k < V) a.probs_out[i0 + k] = (key(k) >= thr) ? (float)((double)q_(k) / (double)kept) : 0.f;
    }
    __syncthreads();
  } else if (a.probs_out) {
#pragma unroll
    for (int k = 0; k < SF_PER; ++k) if (i0 + k < V) a.probs_out[i0 + k] = (i0 + k == s_token) ? 1.f : 0.f;
  }

  // a resumed slot (dtk_resume_slot) forwards the last token of its prompt instead of a sampled one: the draw counter (and
  // with it the begin-suppress rule of the first SAMPLED token) does not move
  const int tok = forced > 0 ? forced - 1 : s_token;
  if (tid == 0) {
    a.tok_ring[a.bs ? 0u : draw % (uint32_t)a.ring] = (int64_t)tok;
    if (a.advance) {
      a.st->token = tok;
      a.st->pos = a.st->next_pos;
      a.st->next_pos = a.st->next_pos + 1;
      a.st->draw = forced > 0 ? draw : draw + 1;
      if (forced > 0) a.st->force_plus1 = 0;
    }
  }
  if (a.advance) {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.embed + (size_t)tok * a.d);
    u32x4* dst = reinterpret_cast<u32x4*>(a.x);
    for (int c = tid; c < (a.d >> 3); c += SAMPLE_THREADS) dst[c] = src[c];
  }
}
#undef key
#undef q_

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
