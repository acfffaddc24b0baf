// mx_quant.h — MXFP8 activations of the batched decode step (kernels_batch_mx.hip has the layouts and the why): the quantiser every
// producer kernel shares, and the addresses of a value / a scale in the fragment-ordered buffers.  oracle/llama.py::mx_fake_quant
// restates mx_exp + the e4m3 rounding bit for bit.
#pragma once
#include "common.h"

// bytes of the MX weight tiles of an [N][K] fp8 matrix: 2 KiB per 16 rows x 128 k (both group sizes)
static inline size_t mx_w_bytes(int N, int K, int G) { (void)G; return (size_t)((N + 15) >> 4) * ((K + 127) / 128) * 2048; }
// bytes of the MXFP8 vectors of 64 slots (always laid out for 4 slot tiles) and of their scales (1 KiB per 512 k at G = 32, per 256 k at G = 16)
static inline size_t mx_x_bytes(int K, int G) { (void)G; return (size_t)((K + 127) / 128) * 4 * 2048; }
static inline size_t mx_s_bytes(int K, int G) { return (size_t)((K + 16 * G - 1) / (16 * G)) * 1024; }

// E8M0 exponent of a group whose largest magnitude is amax: the smallest e with amax * 2^-e <= 448 (= 1.75 * 2^8, the largest e4m3
// value), from the bits of amax = m * 2^E: e = E - 8, one more if m > 1.75.  An all-zero group takes e = 0.
__device__ __forceinline__ int mx_exp(float amax) {
  const unsigned u = __float_as_uint(amax) & 0x7fffffffu;
  if (u == 0u) return 0;
  int e = (int)(u >> 23) - 127 - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
  return max(-127, min(126, e));
}
__device__ __forceinline__ float mx_inv(int e) { return __uint_as_float((unsigned)(127 - e) << 23); }      // 2^-e, e in -127 .. 126

// Where the instruction wants its operands (measured: tools/probe/mx_probe.hip, profiles/r04_mx_probe.txt — NOT 32 consecutive k per
// lane): lane l = (g = l >> 4, i = l & 15) holds, for row / column i, operand byte p (0..31) = k 64 * (p >> 4) + 16 * g + (p & 15) of the
// instruction's 128; the four block scales of row i are the E8M0 bytes (selected by op_sel) of lanes 16 * b + i, block b = k >> 5.  So a
// 32-group of consecutive k sits in the same 16-byte half of two neighbouring lane groups.
//   G = 32: k-step ks = k >> 7; piece (half h = (k >> 6) & 1), lane ((k >> 4) & 3) * 16 + slot % 16, byte k & 15.
//   G = 16: a PAIR step ps = k >> 7 covers 128 k as two instructions s = (k >> 6) & 1 that each see 64 real k: in instruction s only the
//           lane groups g with (g & 1) == s carry weights (the kernel zeroes the others' weight operand), so block q = (k >> 4) & 3 of
//           that instruction holds the 16 values of ONE group: half h = q >> 1, lane group g = 2 * (q & 1) + s.  Scale byte of
//           (ps, s): dword row ps >> 1, byte (ps & 1) * 2 + s = (k >> 6) & 3, lane q * 16 + slot % 16.
__host__ __device__ __forceinline__ size_t mx32_off(int slot, int k) {
  return ((size_t)((k >> 7) * 4 + (slot >> 4)) * 2 + ((k >> 6) & 1)) * 1024 + (size_t)((((k >> 4) & 3) * 16 + (slot & 15)) * 16 + (k & 15));
}
__host__ __device__ __forceinline__ size_t mx32_soff(int slot, int k) {
  return (((size_t)(k >> 9) * 4 + (slot >> 4)) * 64 + ((k >> 5) & 3) * 16 + (slot & 15)) * 4 + ((k >> 7) & 3);
}
__host__ __device__ __forceinline__ size_t mx16_off(int slot, int k) {
  const int q = (k >> 4) & 3, s = (k >> 6) & 1;
  return ((size_t)((k >> 7) * 4 + (slot >> 4)) * 2 + (q >> 1)) * 1024 + (size_t)(((2 * (q & 1) + s) * 16 + (slot & 15)) * 16 + (k & 15));
}
__host__ __device__ __forceinline__ size_t mx16_soff(int slot, int k) {
  return (((size_t)(k >> 8) * 4 + (slot >> 4)) * 64 + ((k >> 4) & 3) * 16 + (slot & 15)) * 4 + ((k >> 6) & 3);
}

__device__ __forceinline__ float mx_amax8(const u32x4& v) {
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(pk_lo(v[e])), fabsf(pk_hi(v[e]))));
  return amax;
}
__device__ __forceinline__ u32x2 mx_cvt8(const u32x4& v, float inv) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(pk_lo(v[0]) * inv, pk_hi(v[0]) * inv, 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(pk_lo(v[1]) * inv, pk_hi(v[1]) * inv, lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(pk_lo(v[2]) * inv, pk_hi(v[2]) * inv, 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(pk_lo(v[3]) * inv, pk_hi(v[3]) * inv, hi, true);
  return (u32x2){(unsigned)lo, (unsigned)hi};
}
// 8 consecutive k (k0 % 8 == 0) of `slot`, held as 4 packed bf16 pairs by this lane; lane l holds k0 = 8 * (something with the
// lane's low bits = l & 3), so the 4 lanes l ^ 1, l ^ 2 hold the rest of the 32-group.  Every lane of a group must call it.
__device__ __forceinline__ void mx32_store8(uint8_t* X8, uint8_t* XS, int slot, int k0, const u32x4& v) {
  float amax = mx_amax8(v);
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  const int e = mx_exp(amax);
  *reinterpret_cast<u32x2*>(X8 + mx32_off(slot, k0)) = mx_cvt8(v, mx_inv(e));
  if ((k0 & 31) == 0) XS[mx32_soff(slot, k0)] = (uint8_t)(e + 127);
}
// the same for groups of 16 (lanes l, l ^ 1)
__device__ __forceinline__ void mx16_store8(uint8_t* X8, uint8_t* XS, int slot, int k0, const u32x4& v) {
  float amax = mx_amax8(v);
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  const int e = mx_exp(amax);
  *reinterpret_cast<u32x2*>(X8 + mx16_off(slot, k0)) = mx_cvt8(v, mx_inv(e));
  if ((k0 & 15) == 0) XS[mx16_soff(slot, k0)] = (uint8_t)(e + 127);
}
