// kernels_batch_decode.hip — one decode step for up to 64 independent sequences ("slots") that
// share ONE pass over the weights: the MCTS rollouts/sec kernel set (independent
// rollouts of one GPU are batched so W is read once for b sequences; bytes/step = W + sum_b K*t_b).
//
// k_gemv_b: Y[slot][n] = W[n][:] . X[slot][:] as a skinny GEMM on v_mfma_f32_16x16x32_bf16 with the
// slots as the MFMA N dimension (NT = 1, 2 or 4 tiles of 16 columns).  The weights are read from a FRAGMENT-MAJOR ("tiled") copy built
// once by k_retile: tile (n/16, k/32) is 1 KiB stored in MFMA A-operand lane order (lane = (k%32/8)*16
// + n%16, 8 bf16 per lane), so one wave-load is 1 KiB CONTIGUOUS and lands directly in the operand
// registers (row-major weights would make every wave-load touch 16 rows x 64 B: measured 3.5 TB/s).
// The B fragments are the slots' input vectors, kept by their producers in B-operand fragment order (xtile_off,
// common.h: one 1 KiB contiguous load per tile, L2 resident).  A block owns 16*T weight
// rows; its 8 waves split K (each wave streams a contiguous K slice of those rows, 4 k-steps in
// flight) and reduce their 16x16 partials through LDS.  Inactive slots are computed and discarded
// (columns are independent), so the captured graph is identical for every active set.
// Epilogues as in kernels_decode.hip (same HF rounding points), applied per active slot.
#include "kernels.h"
#include "mx_quant.h"
#include <stdlib.h>

#define GB_WAVES 8
#define GB_THREADS (GB_WAVES * 64)
#define GB_KSTEP 32
#define GB_UNROLL 4

// rows of tile t of a block.  Paired epilogues own P = T / 2 pairs of 16-row tiles: tile p and its partner p + P
// (QKV: RoPE partners i, i + 64; SWIGLU: gate row i, up row ff + i).  More tiles per block = more weight rows per x
// fragment read from L2: at 64 slots a k-step needs 4 KiB of x per KiB of weights and row tile.
template <int EPI, int T>
__device__ __forceinline__ int gb_tile_row0(const GemvBArgs& a, int blk, int t) {
  constexpr int P = T >= 2 ? T / 2 : 1;
  if (EPI == EPI_QKV) {  // block = 16 * P dims (< 64) of one head block over [H q | KVH k | KVH v]; 64 / (16 P) blocks per head block
    constexpr int BPH = 4 / P;
    return (blk / BPH) * 128 + (blk % BPH) * 16 * P + (t % P) * 16 + (t / P) * 64;
  }
  if (EPI == EPI_SWIGLU) return blk * 16 * P + (t % P) * 16 + (t / P) * a.ff;
  return (blk * T + t) * 16;
}

// two e4m3 words (8 weights of one row) -> the bf16 A fragment of one k-step (exact)
__device__ __forceinline__ bf16x8_t f8x8_to_bf16x8(uint32_t w0, uint32_t w1) {
  u32x4 o;
  o[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, false));
  o[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, true));
  o[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, false));
  o[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, true));
  return __builtin_bit_cast(bf16x8_t, o);
}

// F8: the weights come from the fp8 pair-tiled copy (half the bytes); they are widened to bf16 in registers and
// fed to the same bf16 MFMA in the same k order, the per-row power-of-two scale multiplies the reduced fp32
// sum — bit-identical to the bf16 kernel on the de-quantised weights.
// NT = 16-slot column tiles (1: up to 16 slots, 2: up to 32, 4: up to 64): the A (weight) fragment of a k-step is reused by NT
// MFMAs.  The cross-wave reduction goes through LDS two column tiles at a time (NP), so NT = 4 needs no more LDS than NT = 2.
template <int EPI, int T, int MODE = 0, bool F8 = false, int WAVES = GB_WAVES, int NT = 1, int KS = 4>   // KS = k-steps per register stage
__global__ __launch_bounds__(WAVES * 64, (MODE & 8) ? 4 : 1) void k_gemv_b(GemvBArgs a) {
  constexpr int NP = NT > 2 ? 2 : NT;   // column tiles per reduction pass
  __shared__ float red[WAVES][T][NP][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // MODE & 128 (N = d roles at 64 slots): a block owns T row tiles and HALF of the slots (NT = 2 of the 4 column tiles); the two
  // blocks of a row-tile group are given ids b and b + 8 — the same XCD, dispatched together — so the second one finds the weights
  // in that XCD's L2.  x traffic halves (each block reads 32 slots' worth), the K split and its reduction order are unchanged.
  constexpr bool SPLIT = (MODE & 128) != 0;
  const int b0 = blockIdx.x;
  const int blk = SPLIT ? (((b0 >> 4) << 3) | (b0 & 7)) : b0;
  const int nt0 = SPLIT ? ((b0 >> 3) & 1) * NT : 0;
  const int K = a.K;
  // K slice of this wave, in k-steps of 32 (the last step may be partial: K % 8 == 0)
  const int nsteps = (K + GB_KSTEP - 1) / GB_KSTEP;
  const int per = (nsteps + WAVES - 1) / WAVES;
  const int s0 = min(nsteps, wave * per), s1 = min(nsteps, s0 + per);

  // MODE (timing experiments only): 1 = no x loads, 2 = no MFMA, 4 = no reduction / epilogue, 8 = at least 4 waves per SIMD (<= 128
  // VGPRs: two 8-wave blocks per CU), 16 = two register stages in flight, 32 = one x fragment per k-step feeds all NT column
  // tiles (x traffic / NT: what perfect reuse of x would buy; wrong results), 64 = non-temporal x loads
  const int koff = (lane >> 4) * 8;
  // this lane's 16 bytes of tile 0 of the block's t-th row tile.  bf16: 1 KiB tile = one k-step; fp8: 1 KiB
  // pair tile = two k-steps.  A load "unit" below is one such tile.
  const int units_per_row = F8 ? (nsteps + 1) >> 1 : nsteps;
  const unsigned char* wrow[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    int tn = gb_tile_row0<EPI, T>(a, blk, t) >> 4;
    const int tn_max = ((a.N + 15) >> 4) - 1;
    if (tn > tn_max) tn = tn_max;
    wrow[t] = (F8 ? a.W8 : reinterpret_cast<const unsigned char*>(a.W)) + ((size_t)tn * units_per_row * 64 + lane) * 16;
  }
  const bf16_t* xlane = a.X + lane * 8;   // B fragment of tile (nt, k-step): slot = nt*16 + (lane & 15), fragment-major X (common.h)

  f32x4 acc[T][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A stage = 4 k-steps of weights + x fragments in registers.
  // fp8: the same K slice [s0, s1) as the bf16 kernel; a pair tile that straddles a slice boundary is loaded by
  // both neighbours and the foreign k-step is masked (x fragment = 0), so the k order per accumulator is the same.
  constexpr int U = F8 ? KS / 2 : KS;   // tiles per stage
  constexpr int XN = KS;                // x fragments (k-steps) per stage
  const int u0 = F8 ? (s0 >> 1) : s0, u1 = F8 ? ((s1 + 1) >> 1) : s1;
  auto load = [&](u32x4 (&w)[T][U], u32x4 (&x)[XN][NT], int u) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const bool okp = u + i < u1;
      const int uu = okp ? u + i : u;
#pragma unroll
      for (int t = 0; t < T; ++t) w[t][i] = ld_nt(reinterpret_cast<const u32x4*>(wrow[t] + (size_t)uu * 1024));
#pragma unroll
      for (int h = 0; h < (F8 ? 2 : 1); ++h) {
        const int st = F8 ? 2 * (u + i) + h : u + i;
        const int k = st * GB_KSTEP + koff;
        const bool ok = okp && st >= s0 && st < s1 && k < K;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          u32x4 xv;
          if (MODE & 1) xv = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          else if ((MODE & 32) && nt > 0) xv = x[F8 ? 2 * i + h : i][0];
          else if (MODE & 64) xv = ld_nt(reinterpret_cast<const u32x4*>(xlane + ((size_t)(nt0 + nt) * nsteps + (ok ? st : 0)) * 512));
          else xv = *reinterpret_cast<const u32x4*>(xlane + ((size_t)(nt0 + nt) * nsteps + (ok ? st : 0)) * 512);
          if (!ok) xv = (u32x4){0u, 0u, 0u, 0u};
          x[F8 ? 2 * i + h : i][nt] = xv;
        }
      }
    }
  };
  auto mma = [&](const u32x4 (&w)[T][U], const u32x4 (&x)[XN][NT]) {
#pragma unroll
    for (int i = 0; i < U; ++i)
#pragma unroll
      for (int h = 0; h < (F8 ? 2 : 1); ++h)
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const bf16x8_t af = F8 ? f8x8_to_bf16x8(w[t][i][2 * h], w[t][i][2 * h + 1]) : __builtin_bit_cast(bf16x8_t, w[t][i]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (MODE & 2) {   // timing experiment: no MFMA, the operands are folded in with integer ops so their loads stay live
              const u32x4 wa = __builtin_bit_cast(u32x4, af), xa = x[F8 ? 2 * i + h : i][nt];
              acc[t][nt][0] = __uint_as_float(__float_as_uint(acc[t][nt][0]) ^ wa[0] ^ wa[1] ^ wa[2] ^ wa[3] ^ xa[0] ^ xa[1] ^ xa[2] ^ xa[3]);
            } else
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, x[F8 ? 2 * i + h : i][nt]), acc[t][nt], 0, 0, 0);
          }
        }
  };
  if (F8 || (MODE & 16)) {   // two stages in flight (measured: fp8 step 3.24 -> 2.99 ms at B=16)
    u32x4 wA[T][U], xA[XN][NT], wB[T][U], xB[XN][NT];
    if (u0 < u1) load(wA, xA, u0);
    for (int u = u0; u < u1; u += 2 * U) {
      const bool hb = u + U < u1;
      if (hb) load(wB, xB, u + U);
      mma(wA, xA);
      if (u + 2 * U < u1) load(wA, xA, u + 2 * U);
      if (hb) mma(wB, xB);
    }
  } else {    // bf16: one stage (the second stage costs 128 VGPRs and measured 2 % slower: 3.90 -> 3.99 ms)
    u32x4 wA[T][U], xA[XN][NT];
    for (int u = u0; u < u1; u += U) {
      load(wA, xA, u);
      mma(wA, xA);
    }
  }
  // cross-wave reduction + epilogue, NP column tiles per pass
  auto finish = [&](int q, int nt, int ti) {
    // thread -> (m, n) of a 16x16 tile: C/D layout col n = lane&15, row m = (lane>>4)*4 + reg
    const int l2 = ti >> 2, r2 = ti & 3;
    const int n = (nt0 + nt) * 16 + (l2 & 15);  // slot
    const int m = (l2 >> 4) * 4 + r2;   // row inside the tile
    float v[T];
  #pragma unroll
    for (int t = 0; t < T; ++t) {
      float sum = 0.f;
  #pragma unroll
      for (int w = 0; w < WAVES; ++w) sum += red[w][t][q][ti];
      if (F8) {
        int row = gb_tile_row0<EPI, T>(a, blk, t) + m;
        if (row >= a.N) row = a.N - 1;
        sum *= a.wscale[row];   // power of two: exact
      }
      v[t] = sum;
    }
    if (!a.bs->active[n]) return;
    if (EPI == EPI_RESID) {
  #pragma unroll
      for (int t = 0; t < T; ++t) {
        const int row = gb_tile_row0<EPI, T>(a, blk, t) + m;
        if (row < a.N) {
          bf16_t* y = a.Y + (size_t)n * a.ldy + row;
          *y = f2bf(bf2f(*y) + rbf(v[t]));
        }
      }
    } else if (EPI == EPI_LOGITS) {
  #pragma unroll
      for (int t = 0; t < T; ++t) {
        const int row = gb_tile_row0<EPI, T>(a, blk, t) + m;
        if (row < a.N) a.logits[(size_t)n * a.N + row] = rbf(v[t]);
      }
    } else if (EPI == EPI_STORE) {
  #pragma unroll
      for (int t = 0; t < T; ++t) {
        const int row = gb_tile_row0<EPI, T>(a, blk, t) + m;
        if (row < a.N) a.Y[(size_t)n * a.ldy + row] = f2bf(v[t]);
      }
    } else if (EPI == EPI_SWIGLU) {
      constexpr int P = T >= 2 ? T / 2 : 1;
  #pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = blk * 16 * P + p * 16 + m;
        if (i < a.ff) {
          const float gte = rbf(v[p]), up = rbf(v[p + P]);
          const float sl = rbf(gte / (1.f + expf(-gte)));
          a.Y[xtile_off(n, i, (a.ff + 31) >> 5)] = f2bf(sl * up);   // input of the down projection: fragment-major
        }
      }
    } else if (EPI == EPI_QKV) {
      constexpr int P = T >= 2 ? T / 2 : 1, BPH = 4 / P;
      const int hb = blk / BPH;
      const int sec = hb < a.H ? 0 : (hb < a.H + a.KVH ? 1 : 2);
      const int head = sec == 0 ? hb : (sec == 1 ? hb - a.H : hb - a.H - a.KVH);
      const int pos = a.st[n].pos;
      const size_t slot_kv = (size_t)n * a.kv_slot_stride;
  #pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = (blk % BPH) * 16 * P + p * 16 + m;
        const float x1 = rbf(v[p]), x2 = rbf(v[p + P]);
        if (sec == 2) {
          bf16_t* dst = a.vcache + slot_kv + ((size_t)head * a.T_max + pos) * 128;
          dst[i] = f2bf(x1);
          dst[i + 64] = f2bf(x2);
        } else {
          const float c = bf2f(a.rope_cos[(size_t)pos * 64 + i]);
          const float s = bf2f(a.rope_sin[(size_t)pos * 64 + i]);
          const float o1 = rbf(rbf(x1 * c) + rbf(-x2 * s));
          const float o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
          bf16_t* dst = (sec == 0) ? (a.q_out + (size_t)n * a.d + head * 128)
                                   : (a.kcache + slot_kv + ((size_t)head * a.T_max + pos) * 128);
          dst[i] = f2bf(o1);
          dst[i + 64] = f2bf(o2);
        }
      }
    }

  };
  if (MODE & 4) {   // timing experiment: no cross-wave reduction, no epilogue (one store keeps the accumulators live)
    float sacc = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sacc += acc[t][nt][0] + acc[t][nt][1] + acc[t][nt][2] + acc[t][nt][3];
    if (sacc == 1.2345f) a.Y[tid] = 0;
    return;
  }
#pragma unroll
  for (int p0 = 0; p0 < NT; p0 += NP) {
    if (p0) __syncthreads();   // the previous pass's reads of `red` are done
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][t][q][lane * 4 + r] = acc[t][p0 + q][r];
    __syncthreads();
    for (int it = tid; it < 256 * NP; it += WAVES * 64) finish(it >> 8, p0 + (it >> 8), it & 255);
  }
}


// waves per block of the N = d kernels (o_proj, down: only N/16 = 256 blocks, so the K split is what fills a CU)
static int resid_waves() {
  static int w = 0;
  if (!w) { const char* e = getenv("DTK_GB_RESID_WAVES"); w = (e && atoi(e) == 16) ? 16 : 8; }   // measured: 8 and 16 within 1-2 % (DTK_GB_RESID_WAVES=16 to try)
  return w;
}
// Row tiles per block.  mode 0 = the round-1 shapes (T = 2 for the paired / logits kernels, 1 for the N = d kernels);
// 1 = twice the tiles, 8 waves; 3 = twice the tiles, 4 waves; 4 / 5 = as 1 / 3 but the N = d kernels keep one tile;
// 6 = twice the tiles for qkv and lm_head only; 2 = auto = 6 at 64 slots, else 0.  Measured at 64 slots, ds-7b
// (profiles/r02_batch_wide_kernel_stats.csv vs r02_batch_tail_kernel_stats.csv): qkv 33.1 -> 27.9 us (192 blocks: one round, half
// the x fragments through L2), lm_head 67.9 -> 62.2, gate/up 48.5 -> 48.4 (344 blocks: 1.34 rounds on 256 CUs eat the gain),
// o_proj / down 24.1 -> 29.2 (128 blocks).  Results are bit-identical across modes (a row's k order depends on the wave split
// of K only).
static int g_resid_split = 1;   // 1 (default): N = d roles at 64 slots as 2 row tiles x 32 slots per block (dtk_set_option "resid_split"): o_proj + down 24.0 -> 22.5 us avg
void set_resid_split(int v) { g_resid_split = v; }
static int g_gb_wide = -1;
void set_gemv_b_wide(int v) { g_gb_wide = v; }
static int gb_wide(int nt, bool f8) {
  if (g_gb_wide < 0) { const char* e = getenv("DTK_GB_WIDE"); g_gb_wide = e ? atoi(e) : 2; }
  int m = g_gb_wide;
  if (m == 2) m = nt >= 4 ? 6 : 0;
  if (f8 && nt >= 4) m = 0;                   // fp8 at 64 slots: four tiles do not fit the register file (spills)
  return m;
}
template <bool F8, int NT>
static void launch_gemv_b_impl(int epi, const GemvBArgs& a, hipStream_t s) {
  const int mode = gb_wide(NT, F8);
  const bool wide_qkv_logits = mode == 1 || mode == 3 || mode == 4 || mode == 5 || mode == 6;
  const bool wide = epi == EPI_SWIGLU ? (wide_qkv_logits && mode != 6) : wide_qkv_logits, w4 = mode == 3 || mode == 5;
  const bool wide_resid = mode == 1 || mode == 3;
  if (epi == EPI_QKV) {
    const int hb = a.H + 2 * a.KVH;
    if (wide && w4) hipLaunchKernelGGL((k_gemv_b<EPI_QKV, 4, 0, F8, 4, NT>), dim3(hb * 2), dim3(256), 0, s, a);
    else if (wide) hipLaunchKernelGGL((k_gemv_b<EPI_QKV, 4, 0, F8, GB_WAVES, NT>), dim3(hb * 2), dim3(GB_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_gemv_b<EPI_QKV, 2, 0, F8, GB_WAVES, NT>), dim3(hb * 4), dim3(GB_THREADS), 0, s, a);
  } else if (epi == EPI_SWIGLU) {
    if (wide && w4) hipLaunchKernelGGL((k_gemv_b<EPI_SWIGLU, 4, 0, F8, 4, NT>), dim3((a.ff + 31) / 32), dim3(256), 0, s, a);
    else if (wide) hipLaunchKernelGGL((k_gemv_b<EPI_SWIGLU, 4, 0, F8, GB_WAVES, NT>), dim3((a.ff + 31) / 32), dim3(GB_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_gemv_b<EPI_SWIGLU, 2, 0, F8, GB_WAVES, NT>), dim3((a.ff + 15) / 16), dim3(GB_THREADS), 0, s, a);
  } else if (epi == EPI_RESID) {
    // N = d: only N/16 = 256 one-tile blocks.  Measured in round 1: 8 k-steps per stage (DTK_GB_RESID_KS=8) or 16 waves
    // (DTK_GB_RESID_WAVES=16) change nothing — what bounds them is the 2 KiB (32 slots) / 4 KiB (64) of x fragments read
    // from L2 per KiB of weights, which two tiles per block halve (at the price of 128 blocks)
    static int ks = 0;
    if (!ks) { const char* e = getenv("DTK_GB_RESID_KS"); ks = (e && atoi(e) == 8) ? 8 : 4; }
    if (NT == 4 && g_resid_split && ((a.N + 31) / 32) % 8 == 0)       // (fp8 too: two column tiles per block fit the register file)
      hipLaunchKernelGGL((k_gemv_b<EPI_RESID, 2, 128, F8, GB_WAVES, 2>), dim3(2 * ((a.N + 31) / 32)), dim3(GB_THREADS), 0, s, a);
    else if (wide_resid) hipLaunchKernelGGL((k_gemv_b<EPI_RESID, 2, 0, F8, GB_WAVES, NT>), dim3((a.N + 31) / 32), dim3(GB_THREADS), 0, s, a);
    else if (resid_waves() == 16 && NT < 4 && !F8) hipLaunchKernelGGL((k_gemv_b<EPI_RESID, 1, 0, F8, 16, NT>), dim3((a.N + 15) / 16), dim3(1024), 0, s, a);
    else if (ks == 8) hipLaunchKernelGGL((k_gemv_b<EPI_RESID, 1, 0, F8, GB_WAVES, NT, 8>), dim3((a.N + 15) / 16), dim3(GB_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_gemv_b<EPI_RESID, 1, 0, F8, GB_WAVES, NT>), dim3((a.N + 15) / 16), dim3(GB_THREADS), 0, s, a);
  } else if (epi == EPI_LOGITS) {
    if (wide && w4) hipLaunchKernelGGL((k_gemv_b<EPI_LOGITS, 4, 0, F8, 4, NT>), dim3((a.N + 63) / 64), dim3(256), 0, s, a);
    else if (wide) hipLaunchKernelGGL((k_gemv_b<EPI_LOGITS, 4, 0, F8, GB_WAVES, NT>), dim3((a.N + 63) / 64), dim3(GB_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_gemv_b<EPI_LOGITS, 2, 0, F8, GB_WAVES, NT>), dim3((a.N + 31) / 32), dim3(GB_THREADS), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_gemv_b<EPI_STORE, 1, 0, F8, GB_WAVES, NT>), dim3((a.N + 15) / 16), dim3(GB_THREADS), 0, s, a);
  }
}
static int g_gemv_bx = -1;
void set_gemv_bx(int v) { g_gemv_bx = v; }
void launch_gemv_b(int epi, const GemvBArgs& a, hipStream_t s) {
  if (launch_gemv_bus(epi, a, s)) return;                   // 64 slots, qkv / gate-up: a block per CU, its 8 waves = the 8 K slices, every operand straight into registers (kernels_batch_ks.hip)
  if (launch_gemv_bc(epi, a, s)) return;                    // 64 slots, rows >> d roles: a compute wave per column tile (x from L2 into registers, weights through an LDS ring)
  if (launch_gemv_bl(epi, a, s)) return;                    // 64 slots, rows >> d roles: both operands through LDS rings filled by a loader wave
  if (g_gemv_bx < 0) { const char* e = getenv("DTK_GEMV_BX"); g_gemv_bx = e ? atoi(e) : 1; }
  if (launch_gemv_bx(epi, g_gemv_bx, a, s)) return;         // 64 slots, rows >> d roles: x once per CU (kernels_batch_gemm.hip); false: not covered
  if (a.nt >= 3) { if (a.W8) launch_gemv_b_impl<true, 4>(epi, a, s); else launch_gemv_b_impl<false, 4>(epi, a, s); }
  else if (a.nt == 2) { if (a.W8) launch_gemv_b_impl<true, 2>(epi, a, s); else launch_gemv_b_impl<false, 2>(epi, a, s); }
  else { if (a.W8) launch_gemv_b_impl<true, 1>(epi, a, s); else launch_gemv_b_impl<false, 1>(epi, a, s); }
}

// RMSNorm of the active slots' vectors: one block per slot (HF LlamaRMSNorm rounding); output fragment-major.
__global__ __launch_bounds__(256) void k_rmsnorm_b(const bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y,
                                                   int ldy, int D, float eps, const BatchState* bs, uint8_t* Y8, uint8_t* YS) {
  const int slot = blockIdx.x;
  if (!bs->active[slot]) return;
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D8 = D >> 3;
  const u32x4* x4 = reinterpret_cast<const u32x4*>(X + (size_t)slot * ldx);
  float ss = 0.f;
  for (int c = tid; c < D8; c += 256) {
    const u32x4 v = x4[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) {     // explicit fma: k_resid_norm_b (kernels_batch_gemm.hip) folds the squares in this exact order
      const float lo = pk_lo(v[e]), hi = pk_hi(v[e]);
      ss = __builtin_fmaf(lo, lo, ss);
      ss = __builtin_fmaf(hi, hi, ss);
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
  const u32x4* w4 = reinterpret_cast<const u32x4*>(w);
  const int nsteps = (D + 31) >> 5;   // Y is fragment-major (the next kernel's B operand)
  for (int c = tid; c < D8; c += 256) {
    const u32x4 v = x4[c], g = w4[c];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2(pk_lo(g[e]) * rbf(pk_lo(v[e]) * inv), pk_hi(g[e]) * rbf(pk_hi(v[e]) * inv));
    if (Y8) mx32_store8(Y8, YS, slot, c * 8, o);   // fp8 matrix-core step: the same bf16 values as MXFP8 (D % 32 == 0: whole groups per iteration)
    else *reinterpret_cast<u32x4*>(Y + xtile_off(slot, c * 8, nsteps)) = o;
  }
}
void launch_rmsnorm_b(const bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int ldy, int D, float eps,
                      const BatchState* bs, int nslots, hipStream_t s, uint8_t* Y8, uint8_t* YS) {
  hipLaunchKernelGGL(k_rmsnorm_b, dim3(nslots), dim3(256), 0, s, X, ldx, w, Y, ldy, D, eps, bs, Y8, YS);
}

// ------------------------------------------------------------------------------------------
// Shared prefixes on the matrix cores.  The rollouts of one image hold bit-identical copies of the image prefix (dtk_kv_fork):
// scoring it per slot on the VALU re-reads the same 243 rows once per slot from L2 (64 slots x 32 heads blocks, 255 MB of L2 -> CU
// traffic per layer for 4 MB of keys and values) and spends the step's largest kernel on a [slots x 128] . [128 x 243] GEMM done one
// dot product at a time.  k_attn_prefix_g does it per GROUP (PfxGroup: <= 16 slots that read keys [0, len) from one source slot) as a
// 16-query flash attention on v_mfma_f32_16x16x32_bf16: S^T = K Q^T and O^T = V^T P^T, computed transposed so that a lane's MFMA
// column is its own query and the S^T accumulator IS the P^T operand (the scheme of k_attention_mfma, kernels_batched.hip), fp32
// online softmax, P as a bf16 hi + lo pair (the fp32 probabilities of the VALU path to ~16 mantissa bits).
// One WAVE per block, grid (H, DTK_PFX_GROUPS, key splits): 32 x 4 x 4 = 512 live blocks for 64 forks of one image, 32 x 8 x 4 for
// BASELINE config 5's 8 images x 8 trees (round 2's k_attn_prefix_b had 64-128 blocks of four waves behind __syncthreads: no gain).
// A block's K fragments come straight from global memory in the A-operand order (16 keys x 8 dims per lane load, no LDS); only V
// goes through (wave-private) LDS, for the transposition.  It leaves every member's UN-normalised state (m, l, o[128]) per key
// split; k_attn_tail_b starts from it and continues over the slot's private keys.
#define PFX_HDP 136   // LDS row stride of a 128-dim V row (elements): 68 dwords, conflict-free 16-byte fragment reads
__global__ __launch_bounds__(64) void k_attn_prefix_g(AttnDecBArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * PFX_HDP];
  const BatchState* bs = a.bs;
  const int h = blockIdx.x, z = blockIdx.z, NPS = gridDim.z;
  const int lane = threadIdx.x, lq = lane & 15, g = lane >> 4;
  // The grid has DTK_PFX_GRID = 16 group rows; a step may hold up to DTK_PFX_GROUPS = 64 groups (64 slots that share nothing): block row
  // y takes groups y, y + 16, ... — one trip in every measured configuration (config 5: 8 groups), and no slot's prefix ever leaves the
  // matrix cores because of how many OTHER prefixes the step holds (a 64-row grid cost 1.5 % of the step in idle blocks).
  for (int gi = blockIdx.y; gi < DTK_PFX_GROUPS; gi += DTK_PFX_GRID) {
  // the whole group record is requested at once, BEFORE n_groups is known (the table is plain memory whatever it holds): one memory
  // round trip, then the q / K / V loads — the block is a chain of dependent loads, and each link costs 1-2 us under a full chip
  const PfxGroup* grp = bs->groups + gi;
  const int ngroups = bs->n_groups, Tk = grp->len, nmem = grp->n, src = grp->src;
  const int slot = grp->slot[lq];                       // this lane's query (the host fills the columns beyond the group with member 0; they are dropped)
  if (gi >= ngroups) return;
  const bool real = lq < nmem;
  const int kvh = h / a.G;
  const size_t src_off = (size_t)src * a.kv_slot_stride + (size_t)kvh * a.T_max * 128;
  const bf16_t* Kp = a.kcache + src_off;
  const bf16_t* Vp = a.vcache + src_off;
  const int tiles = (Tk + 63) >> 6, tps = (tiles + NPS - 1) / NPS;
  const int j_begin = z * tps * 64, j_end = min(Tk, (z + 1) * tps * 64);

  bf16x8_t qf[4];
  {
    const bf16_t* qp = a.q + (size_t)slot * a.d + h * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(qp + (ks * 4 + g) * 8));
  }
  float m = -1e30f, l = 0.f;
  f32x4 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int j0 = j_begin; j0 < j_end; j0 += 64) {
    // everything the tile needs is requested before anything is used: 16 K fragments + 16 V row pieces of 16 bytes per lane
    u32x4 kf[4][4], rv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = min(j0 + t * 16 + lq, Tk - 1);        // A operand: row = key t * 16 + lq, k = dims (ks * 4 + g) * 8 .. + 8
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[t][ks] = *reinterpret_cast<const u32x4*>(Kp + (size_t)j * 128 + (ks * 4 + g) * 8);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = lane + i * 64, r = idx >> 4, c = idx & 15;      // 64 rows x 16 pieces
      const int j = min(j0 + r, Tk - 1);
      rv[i] = *reinterpret_cast<const u32x4*>(Vp + (size_t)j * 128 + c * 8);
    }
    if (j0 != j_begin) __syncthreads();                   // (one wave: the previous tile's V reads are done before Vs is overwritten)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = lane + i * 64, r = idx >> 4, c = idx & 15;
      *reinterpret_cast<u32x4*>(&Vs[r * PFX_HDP + c * 8]) = rv[i];
    }
    f32x4 sc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[t][ks]), qf[ks], sc[t], 0, 0, 0);
    }
    float tmax = -1e30f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + t * 16 + g * 4 + r;
        const float v = key < j_end ? sc[t][r] * a.scale : -1e30f;
        sc[t][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m, tmax);
    const float corr = __expf(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + t * 16 + g * 4 + r;
        const float p = key < j_end ? __expf(sc[t][r] - mn) : 0.f;
        sc[t][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l = l * corr + psum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] *= corr;
    __syncthreads();                                      // V tile visible to every lane of the wave
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u32x4 pw, pl;   // p = hi + lo (two bf16)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const float p0 = sc[2 * s2 + half][2 * pr], p1 = sc[2 * s2 + half][2 * pr + 1];
          const uint32_t hi = pack2(p0, p1);
          pw[half * 2 + pr] = hi;
          pl[half * 2 + pr] = pack2(p0 - pk_lo(hi), p1 - pk_hi(hi));
        }
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw), pfl = __builtin_bit_cast(bf16x8_t, pl);
      const bf16_t* v0 = Vs + ((2 * s2) * 16 + g * 4) * PFX_HDP + lq;
      const bf16_t* v1 = v0 + 16 * PFX_HDP;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        u32x4 vw;
        vw[0] = (uint32_t)v0[dt * 16] | ((uint32_t)v0[dt * 16 + PFX_HDP] << 16);
        vw[1] = (uint32_t)v0[dt * 16 + 2 * PFX_HDP] | ((uint32_t)v0[dt * 16 + 3 * PFX_HDP] << 16);
        vw[2] = (uint32_t)v1[dt * 16] | ((uint32_t)v1[dt * 16 + PFX_HDP] << 16);
        vw[3] = (uint32_t)v1[dt * 16 + 2 * PFX_HDP] | ((uint32_t)v1[dt * 16 + 3 * PFX_HDP] << 16);
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vw);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfl, o[dt], 0, 0, 0);
      }
    }
  }
  if (real) {   // un-normalised state of (slot, head, key split z): lane holds dims dt * 16 + g * 4 + 0..3 of its query
    const size_t rec = ((size_t)slot * a.H + h) * NPS + z;
    float* po = a.pfx_o + rec * 128;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16 + g * 4) = o[dt];
    if (g == 0) { a.pfx_m[rec] = m; a.pfx_l[rec] = l; }
  }
  __syncthreads();          // (the next group of this block row reuses Vs)
  }
}

// Per-slot remainder: block (h, slot) walks the slot's keys that are not covered by the shared prefix — [pfx_len, n) for a
// member, [0, n) otherwise — in tiles of 16 * WAVES rows with the next tile in flight, starting from the prefix state, and
// writes the normalised bf16 head output in o_proj's fragment-major order: no split partials, no combine kernel (64 slots x
// 32 heads are 2048 blocks already).  Rows below a fork's share_len still come from its source slot (one copy in L2).
// GQ = query heads per block: 1, or the GQA group (H / KVH = 4 for the v2 models): the heads of a group attend over the SAME
// K / V rows, so the block loads a row once and scores it against GQ queries.  Per head the arithmetic — which lane owns which
// keys, the order of the online-softmax updates, the cross-lane and cross-wave merges — is exactly that of GQ = 1 (the compiler
// contracts the multiply-adds differently in the two instantiations: equal to fp32 rounding, tested, not bit for bit); the key /
// value traffic through L2 drops by GQ (ds-7b has no groups: GQ = 1).  A slot's result still does not depend on the other slots.
template <int THREADS, int GQ = 1>
__global__ __launch_bounds__(THREADS) void k_attn_tail_b(AttnDecBArgs a) {
  constexpr int WAVES = THREADS / 64, ROWS = WAVES * 16;
  const int h0 = blockIdx.x * GQ, slot = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, grp = lane >> 4;
  // Every scalar the block needs and its q rows are requested TOGETHER, before the first of them is looked at: with a few private keys
  // per slot the block is a chain of dependent memory round trips (1-2 us each under 2048 resident blocks), and round 4's order —
  // active, then share_src / share_len / pos, then the group's length, then K / V, then the prefix state split by split — was eight
  // of them: 19.7 us of k_attn_tail_b with 4 keys to score (profiles/r05b_batch64_fp8_prefix_kernel_stats.csv).
  const BatchState* bs = a.bs;
  const int is_active = bs->active[slot];
  const int ssrc = bs->share_src[slot], slen_raw = bs->share_len[slot];
  const int start = a.use_prefix ? bs->pfx_len_of[slot] : 0;       // keys [0, start) were scored by k_attn_prefix_g (0: none)
  const int n = a.st[slot].pos + 1;
  u32x4 qv[GQ];
#pragma unroll
  for (int g = 0; g < GQ; ++g) qv[g] = reinterpret_cast<const u32x4*>(a.q + (size_t)slot * a.d + (h0 + g) * 128)[sub];
  if (!is_active) return;
  const int kvh = h0 / a.G;
  const bf16_t* kbase = a.kcache + (size_t)slot * a.kv_slot_stride + (size_t)kvh * a.T_max * 128;
  const bf16_t* vbase = a.vcache + (size_t)slot * a.kv_slot_stride + (size_t)kvh * a.T_max * 128;
  const int slen = ssrc >= 0 ? slen_raw : 0;
  const size_t sdelta = ((size_t)(ssrc >= 0 ? ssrc : slot) - (size_t)slot) * a.kv_slot_stride;
  const bool member = start > 0;

  // Two register sets of K / V rows, ping-pong: while tile j is scored from one set, tile j + 1 is already in the other and tile
  // j + 2 is requested as soon as its set is free — one to two tiles (8-16 KiB per wave) in flight instead of one.  (Round 2 copied
  // the landed tile into a second set and then requested the next one: a block walked its context at one memory round trip per
  // 64 keys, 16 round trips per block slot and layer.)  The keys are scored in the same order, so the result is bit-identical.
  u32x4 kA[4], vA[4], kB[4], vB[4];
  // a tile that lies wholly beyond the shared prefix holds PRIVATE rows: read once per step by this block alone, so they are loaded
  // non-temporally (option "attn_nt", default on) and do not push the shared prefix rows and the x fragments out of the XCD's L2
  const bool nt_ok = a.nt_private != 0;
  // Rows at or beyond the context are NOT loaded (score_tile never looks at them): a private tile's rows lie in this slot's cache
  // alone, so what round 4 fetched there by clamping (up to 63 rows x 512 B per block: with 2048 blocks a tile that holds 4 real
  // keys cost 63 MB of HBM reads per layer, 13 us of a 19.5 us launch) was pure waste.
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_tile = [&](u32x4 (&kk)[4], u32x4 (&vv)[4], int j0) {
    if (nt_ok && j0 >= slen) {      // block-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = j0 + i * (ROWS / 4) + wave * 4 + grp;
        if (j < n) {
          kk[i] = ld_nt(reinterpret_cast<const u32x4*>(kbase + (size_t)j * 128) + sub);
          vv[i] = ld_nt(reinterpret_cast<const u32x4*>(vbase + (size_t)j * 128) + sub);
        } else { kk[i] = zero4; vv[i] = zero4; }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + i * (ROWS / 4) + wave * 4 + grp;
      if (j < n) {
        const size_t off = (size_t)j * 128 + (j < slen ? sdelta : (size_t)0);
        kk[i] = reinterpret_cast<const u32x4*>(kbase + off)[sub];
        vv[i] = reinterpret_cast<const u32x4*>(vbase + off)[sub];
      } else { kk[i] = zero4; vv[i] = zero4; }
    }
  };
  if (start < n) load_tile(kA, vA, start);
  if (start + ROWS < n) load_tile(kB, vB, start + ROWS);
  float m[GQ], l[GQ], o[GQ][8];
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    m[g] = -1e30f; l[g] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
  }
  if (member && wave == 0 && grp == 0) {   // the prefix state (its key splits merged in order) seeds one of the block's streams
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      const size_t rec0 = ((size_t)slot * a.H + h0 + g) * a.pfx_splits;
      float sm2[4], sl2[4];
      f32x4 sp0[4], sp1[4];
#pragma unroll
      for (int zsp = 0; zsp < 4; ++zsp) {    // all splits' records in flight at once (pfx_splits <= 4), merged in order below
        const size_t rec = rec0 + (zsp < a.pfx_splits ? zsp : 0);
        sm2[zsp] = a.pfx_m[rec]; sl2[zsp] = a.pfx_l[rec];
        const f32x4* po4 = reinterpret_cast<const f32x4*>(a.pfx_o + rec * 128 + sub * 8);
        sp0[zsp] = po4[0]; sp1[zsp] = po4[1];
      }
#pragma unroll
      for (int zsp = 0; zsp < 4; ++zsp) {
        if (zsp >= a.pfx_splits) break;
        const float m2 = sm2[zsp], l2 = sl2[zsp];
        const f32x4 p0 = sp0[zsp], p1 = sp1[zsp];
        const float mn = fmaxf(m[g], m2);
        const float c1 = __expf(m[g] - mn), c2 = __expf(m2 - mn);
        l[g] = l[g] * c1 + l2 * c2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[g][e] = o[g][e] * c1 + p0[e] * c2;
          o[g][4 + e] = o[g][4 + e] * c1 + p1[e] * c2;
        }
        m[g] = mn;
      }
    }
  }
  auto score_tile = [&](const u32x4 (&kc)[4], const u32x4 (&vc)[4], int j0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + i * (ROWS / 4) + wave * 4 + grp;
#pragma unroll
      for (int g = 0; g < GQ; ++g) {
        float s = dot8(qv[g], kc[i], 0.f);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        s *= a.scale;
        if (j < n) {
          const float mn = fmaxf(m[g], s);
          const float corr = __expf(m[g] - mn);
          const float p = __expf(s - mn);
          l[g] = l[g] * corr + p;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[g][2 * e] = o[g][2 * e] * corr + p * pk_lo(vc[i][e]);
            o[g][2 * e + 1] = o[g][2 * e + 1] * corr + p * pk_hi(vc[i][e]);
          }
          m[g] = mn;
        }
      }
    }
  };
  for (int j0 = start; j0 < n; j0 += 2 * ROWS) {
    score_tile(kA, vA, j0);
    if (j0 + 2 * ROWS < n) load_tile(kA, vA, j0 + 2 * ROWS);
    if (j0 + ROWS < n) {
      score_tile(kB, vB, j0 + ROWS);
      if (j0 + 3 * ROWS < n) load_tile(kB, vB, j0 + 3 * ROWS);
    }
  }
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float m2 = __shfl_xor(m[g], off, 64);
      const float l2 = __shfl_xor(l[g], off, 64);
      const float mn = fmaxf(m[g], m2);
      const float c1 = __expf(m[g] - mn), c2 = __expf(m2 - mn);
      l[g] = l[g] * c1 + l2 * c2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float o2 = __shfl_xor(o[g][e], off, 64);
        o[g][e] = o[g][e] * c1 + o2 * c2;
      }
      m[g] = mn;
    }
  }
  __shared__ float sm_m[GQ][WAVES][16], sm_l[GQ][WAVES][16], sm_o[GQ][WAVES][16][8];
  if (grp == 0) {
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
      sm_m[g][wave][sub] = m[g];
      sm_l[g][wave][sub] = l[g];
#pragma unroll
      for (int e = 0; e < 8; ++e) sm_o[g][wave][sub][e] = o[g][e];
    }
  }
  __syncthreads();
  if (tid < 16 * GQ) {
    const int g = tid >> 4, t16 = tid & 15;
    float M = sm_m[g][0][t16];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) M = fmaxf(M, sm_m[g][w][t16]);
    float L = 0.f;
    float oo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) oo[e] = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float c = __expf(sm_m[g][w][t16] - M);
      L += c * sm_l[g][w][t16];
#pragma unroll
      for (int e = 0; e < 8; ++e) oo[e] += c * sm_o[g][w][t16][e];
    }
    const float invL = 1.f / L;
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack2(oo[2 * e] * invL, oo[2 * e + 1] * invL);
    if (a.out8) mx32_store8(a.out8, a.outs, slot, (h0 + g) * 128 + t16 * 8, ov);   // fp8 matrix-core step: o_proj's input as MXFP8 (lanes t16 ^ 1, t16 ^ 2 = the rest of the group)
    else *reinterpret_cast<u32x4*>(a.out + xtile_off(slot, (h0 + g) * 128 + t16 * 8, (a.d + 31) >> 5)) = ov;   // 8 consecutive k of one fragment lane
  }
}

void launch_attn_decode_b(const AttnDecBArgs& a, hipStream_t s) {
  {   // shared prefixes once per group on the matrix cores (optional) + one block per (head, slot) for the private keys
    if (a.use_prefix) hipLaunchKernelGGL(k_attn_prefix_g, dim3(a.H, DTK_PFX_GRID, a.pfx_splits), dim3(64), 0, s, a);
    // one block shape for every slot count: a slot's result must not depend on how many column tiles the step has
    if (a.G == 4 && a.gqa_fused == 2) {   // pairs of query heads: half the sharing, twice the blocks
      hipLaunchKernelGGL((k_attn_tail_b<256, 2>), dim3(a.H / 2, a.nslots), dim3(256), 0, s, a);
      return;
    }
    if (a.G == 4 && a.gqa_fused) {   // GQA: the four query heads of a K / V head in one block (same arithmetic per head, a quarter of the K / V reads)
      if (a.tail_threads == 64) hipLaunchKernelGGL((k_attn_tail_b<64, 4>), dim3(a.H / 4, a.nslots), dim3(64), 0, s, a);
      else if (a.tail_threads == 128) hipLaunchKernelGGL((k_attn_tail_b<128, 4>), dim3(a.H / 4, a.nslots), dim3(128), 0, s, a);
      else if (a.tail_threads == 512) hipLaunchKernelGGL((k_attn_tail_b<512, 4>), dim3(a.H / 4, a.nslots), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((k_attn_tail_b<256, 4>), dim3(a.H / 4, a.nslots), dim3(256), 0, s, a);
      return;
    }
    if (a.G == 2 && a.gqa_fused) {
      if (a.tail_threads == 64) hipLaunchKernelGGL((k_attn_tail_b<64, 2>), dim3(a.H / 2, a.nslots), dim3(64), 0, s, a);
      else if (a.tail_threads == 128) hipLaunchKernelGGL((k_attn_tail_b<128, 2>), dim3(a.H / 2, a.nslots), dim3(128), 0, s, a);
      else hipLaunchKernelGGL((k_attn_tail_b<256, 2>), dim3(a.H / 2, a.nslots), dim3(256), 0, s, a);
      return;
    }
    if (a.tail_threads == 64) hipLaunchKernelGGL(k_attn_tail_b<64>, dim3(a.H, a.nslots), dim3(64), 0, s, a);
    else if (a.tail_threads == 128) hipLaunchKernelGGL(k_attn_tail_b<128>, dim3(a.H, a.nslots), dim3(128), 0, s, a);
    else if (a.tail_threads == 256) hipLaunchKernelGGL(k_attn_tail_b<256>, dim3(a.H, a.nslots), dim3(256), 0, s, a);
    else if (a.tail_threads == 1024) hipLaunchKernelGGL(k_attn_tail_b<1024>, dim3(a.H, a.nslots), dim3(1024), 0, s, a);   // multi-vector step: 256 keys per tile
    else hipLaunchKernelGGL(k_attn_tail_b<512>, dim3(a.H, a.nslots), dim3(512), 0, s, a);
    return;
  }
}

// Row-major [N][K] bf16 -> fragment-major tiles (see the header): storage = ceil(N/16)*ceil(K/32) tiles
// of 512 elements, zero padded.  One thread per 16-byte lane slot.
__global__ void k_retile(const bf16_t* src, bf16_t* dst, int N, int K) {
  const int K32 = (K + 31) >> 5, N16 = (N + 15) >> 4;
  const long total = (long)N16 * K32 * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tile = i >> 6;
    const int tk = (int)(tile % K32), tn = (int)(tile / K32);
    const int n = tn * 16 + (lane & 15), k = tk * 32 + (lane >> 4) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N && k < K) v = *reinterpret_cast<const u32x4*>(src + (size_t)n * K + k);
    reinterpret_cast<u32x4*>(dst)[i] = v;
  }
}
// Row-major fp8 [N][K] -> pair tiles (kernels.h): one thread per 16-byte lane slot.
__global__ void k_retile_f8(const uint8_t* src, uint8_t* dst, int N, int K) {
  const int K64 = (K + 63) >> 6, N16 = (N + 15) >> 4;
  const long total = (long)N16 * K64 * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tile = i >> 6;
    const int tk = (int)(tile % K64), tn = (int)(tile / K64);
    const int n = tn * 16 + (lane & 15), k = tk * 64 + (lane >> 4) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N) {   // K % 8 == 0: an 8-byte group is fully in or fully out
      if (k < K) { const u32x2 lo = *reinterpret_cast<const u32x2*>(src + (size_t)n * K + k); v[0] = lo[0]; v[1] = lo[1]; }
      if (k + 32 < K) { const u32x2 hi = *reinterpret_cast<const u32x2*>(src + (size_t)n * K + k + 32); v[2] = hi[0]; v[3] = hi[1]; }
    }
    reinterpret_cast<u32x4*>(dst)[i] = v;
  }
}
void launch_retile_f8(const uint8_t* src, uint8_t* dst, int N, int K, hipStream_t s) {
  const long total = (long)((N + 15) >> 4) * ((K + 63) >> 6) * 64;
  long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_retile_f8, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, N, K);
}
void launch_retile(const bf16_t* src, bf16_t* dst, int N, int K, hipStream_t s) {
  const long total = (long)((N + 15) >> 4) * ((K + 31) >> 5) * 64;
  long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_retile, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, N, K);
}
