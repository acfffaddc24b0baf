// gemv_inl.h — the streaming core shared by the single-sequence GEMVs (kernels_decode.hip: k_gemv) and the multi-vector GEMVs of
// the <= 4-slot batched step (kernels_decode_mv.hip: k_gemv_mv): a wave owns NR weight rows for the full K, lanes stride K in
// 16-byte chunks (64 lanes x 16 B = 1 KiB per row per load instruction), non-temporal loads straight into VGPRs, two register
// stages, v_dot2c_f32_bf16 against x chunks held in LDS.  A lane folds chunk lane, lane + 64, lane + 128, ... of its row in that
// order whatever U is, so every kernel built on these helpers produces the same fp32 sum for the same (row, x).
#pragma once
#include "common.h"

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
