// kernels_batch_mx.hip — the batched decode step of an fp8-weight model ON THE fp8 MATRIX CORES (BASELINE config 5: "detikzify-cl-7b fp8
// weights (CDNA4 fp8 MFMA)").
//
// Round 4's counters (profiles/r04_batch64_fp8_pmc_sq.csv) showed the fp8 64-slot kernels bound by the instruction stream of their
// compute waves — fp8 -> bf16 conversion of every weight byte + one bf16 MFMA per 32 k — not by memory: half the bytes of the bf16
// model in the same microseconds.  Here the weights stay fp8 all the way into v_mfma_scale_f32_16x16x128_f8f6f4 (no conversion, a
// quarter of the MFMA issues) and the slots' input vectors are MXFP8: e4m3 values with one power-of-two scale (E8M0) per group of
// consecutive k of one slot, produced by the kernels that produce the vectors anyway:
//   * RMSNorm output (input of q/k/v, gate/up, lm_head) and attention output (input of o_proj): groups of 32 — the block-scale
//     granule of the instruction;
//   * SwiGLU output (input of down): groups of 16 = the rows of one MFMA row tile, which is what a wave of the gate/up kernel
//     owns; down issues two instructions per 128 k, each with half of its weight lanes zeroed, so that a scale block of the
//     instruction (32 operand bytes across two lane groups) carries one 16-group (64 real k per MFMA).
// The scale of a group is the smallest power of two that brings its largest magnitude to <= 448 (e4m3 max): no saturation.  The
// weights keep their per-row power-of-two scale (k_quant_fp8_rows), applied to the fp32 sums (exact).
//
// Numerics: activations carry 3 mantissa bits into the four Linear inputs of a layer (and lm_head) instead of 7 — a different,
// coarser model than the bf16-activation fp8 path (option "act_fp8" = 0 restores that one).  oracle/llama.py restates the
// quantiser bit for bit (LlamaOracle.act_quant); the parity tests compare against THAT oracle and report the distance to the
// bf16-activation oracle next to it (fp8 parity = bounded error).
//
// Layouts.  The instruction does NOT take 32 consecutive k per lane (what ck_tile's descriptor suggests, harmless without scales):
// measured with tools/probe/mx_probe.hip (profiles/r04_mx_probe.txt), lane l = (g = l >> 4, i = l & 15) holds for row / column i the
// operand bytes p = 0..31 = k 64 * (p >> 4) + 16 * g + (p & 15), and scale block b = k >> 5 is the E8M0 byte of lane 16 * b + i.  Both
// operands are kept in memory as 2 KiB per (16 rows or slots, 128 k): [half h][lane][16 B] — every 1 KiB piece is one
// `global_load_lds_dwordx4` / `global_load_dwordx4` / `ds_read_b128` of a wave, fully coalesced; slot tiles always laid out for 4:
//   G = 32 (k-step ks = k / 128):   piece h, lane (g, i) = k ks*128 + 64 h + 16 g + 0..15;  one instruction per k-step
//   G = 16 (pair step ps = k / 128): piece h, lane (g, i) = k ps*128 + 64 (g & 1) + 32 h + 16 (g >> 1) + 0..15; TWO instructions s = 0, 1
//           per pair step, the weight operand of instruction s zeroed in the lane groups with (g & 1) != s: each of its four scale
//           blocks then holds exactly one 16-group (64 real k per instruction), the x operand is read once for both
//   scales: 1 KiB rows of dwords [tile][lane]; G = 32: row ks >> 2, byte ks & 3, lane b * 16 + slot;  G = 16: row ps >> 1, byte
//           (ps & 1) * 2 + s, lane q * 16 + slot (q = 16-group within the instruction)
#include "batch_epi.h"
#include "mx_quant.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
template <bool B> struct mx_flag { static constexpr bool value = B; };

__device__ __forceinline__ i32x8 mx_op(const u32x4& lo, const u32x4& hi) {
  return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
// a (fp8 weights, scale 1) x b (MXFP8 slots' vectors, block scale in every byte of sb): the weight row scale is applied afterwards
__device__ __forceinline__ f32x4 mx_mfma(const i32x8& a, const i32x8& b, const f32x4& c, unsigned sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, (int)sb);
}

// ---------------------------------------------------------------------------------------------------------------- retile
// row-major fp8 [N][K] -> MX weight tiles; one thread per 16-byte lane slot
__global__ void k_retile_mx(const uint8_t* src, uint8_t* dst, int N, int K, int G) {
  const int nks = (K + 127) >> 7, N16 = (N + 15) >> 4;
  const long total = (long)N16 * nks * 2 * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), g = lane >> 4;
    long p = i >> 6;                                        // piece = (tile * nks + ks) * 2 + half
    const int half = (int)(p & 1); p >>= 1;
    const int ks = (int)(p % nks), tn = (int)(p / nks);
    const int n = tn * 16 + (lane & 15);
    const int k = ks * 128 + (G == 32 ? 64 * half + 16 * g : 64 * (g & 1) + 32 * half + 16 * (g >> 1));
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N && k < K) v = *reinterpret_cast<const u32x4*>(src + (size_t)n * K + k);     // K % 16 == 0
    reinterpret_cast<u32x4*>(dst)[i] = v;
  }
}
void launch_retile_mx(const uint8_t* src, uint8_t* dst, int N, int K, int G, hipStream_t s) {
  const long total = (long)(mx_w_bytes(N, K, G) / 16);
  long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_retile_mx, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, N, K, G);
}

// row-major bf16 X [slots][K] -> MXFP8 (op-level tests: the quantiser and the layout in isolation).  One block per slot, a thread per
// 8 consecutive k — the shape every producer of the step has.
__global__ __launch_bounds__(256) void k_quant_mx_rows(const bf16_t* X, int K, uint8_t* X8, uint8_t* XS, int G) {
  const int slot = blockIdx.x;
  for (int c = threadIdx.x; c < (K >> 3); c += 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(X + (size_t)slot * K + c * 8);
    if (G == 32) mx32_store8(X8, XS, slot, c * 8, v);
    else mx16_store8(X8, XS, slot, c * 8, v);
  }
}
void launch_quant_mx_rows(const bf16_t* X, int K, uint8_t* X8, uint8_t* XS, int G, int nslots, hipStream_t s) {
  hipLaunchKernelGGL(k_quant_mx_rows, dim3(nslots), dim3(256), 0, s, X, K, X8, XS, G);
}

// ---------------------------------------------------------------------------------------------------------------- unit kernel
// q/k/v (RoPE pairs), gate/up (SwiGLU pairs), lm_head: a compute wave owns one UNIT = two paired row tiles x NT slot tiles over the
// whole K (accumulators 2 x NT x 4 registers); a block is NC compute waves + one loader wave.
//   * x (G = 32) is read ONCE per block: the loader wave streams it by LDS-DMA in phases of 4 k-steps (512 k: NT x 8 KiB + 1 KiB of
//     scales) into a ring of 3, two phases ahead; one raw s_barrier per phase is the hand-off (the loader waits `vmcnt` down to the
//     pieces of the younger phase first; a compute wave drains its LDS reads).  A __syncthreads() would drain the weight loads too.
//   * the weights go global -> registers (non-temporal, 1 KiB per wave instruction), two phases (2 x 16 KiB per wave) in flight: the
//     registers of a k-step are refilled for phase p + 2 as soon as its MFMAs are issued.
// K order of an accumulator: k-steps in order — independent of NT, so a slot's result does not depend on how many tiles decode.
template <int NT>
__device__ __forceinline__ void mx_swiglu_finish(const GemvBArgs& a, int g, const f32x4 (&tot)[2][NT], int lane) {
  const int m0 = (lane >> 4) * 4;
  int act[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) act[nt] = a.bs->active[nt * 16 + (lane & 15)];
  const f32x4 sg = *reinterpret_cast<const f32x4*>(a.wscale + g * 16 + m0);
  const f32x4 su = *reinterpret_cast<const f32x4*>(a.wscale + a.ff + g * 16 + m0);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float v[4], amax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                  // the rounding points of gg_epilogue<EPI_SWIGLU>
      const float gte = rbf(tot[0][nt][r] * sg[r]), up = rbf(tot[1][nt][r] * su[r]);
      const float sl = rbf(gte / (1.f + expf(-gte)));
      v[r] = rbf(sl * up);
      amax = fmaxf(amax, fabsf(v[r]));
    }
    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));                  // the 16 rows of the unit for this slot: lanes l, l^16, l^32, l^48
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    const int e = mx_exp(amax);
    const float inv = mx_inv(e);
    int q = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    q = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, q, true);
    if (act[nt]) {
      const int slot = nt * 16 + (lane & 15), k = g * 16 + m0;
      *reinterpret_cast<int*>(a.Y8 + mx16_off(slot, k)) = q;
      if (m0 == 0) a.YS[mx16_soff(slot, k)] = (uint8_t)(e + 127);
    }
  }
}

// WD = weight phases in flight per wave: 2, or 1 where five waves share the register file (NC = 4 at 4 slot tiles: 2 would spill)
template <int EPI, int NC, int NT, int WD = (NC == 4 && NT == 4) ? 1 : 2>
__global__ __launch_bounds__((NC + 1) * 64) void k_gemv_mxu(GemvBArgs a) {
  constexpr int PH = 4, R = 3;
  constexpr unsigned XPH = PH * NT * 2048u, PHB = XPH + 1024u;     // ring slot: the phase's x pieces + its 1 KiB of scales
  constexpr int PIECES = PH * NT * 2 + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nks = a.K >> 7, nph = nks / PH;                          // launcher: K % 512 == 0

  if (wave == NC) {   // ---- loader wave
    const unsigned char* xl = a.X8 + lane * 16;
    const unsigned char* sl = a.XS + lane * 16;
    auto issue = [&](int p) {
      const unsigned slot = (unsigned)(p % R) * PHB;
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int i = 0; i < NT * 2; ++i)                            // (tile, half) of k-step p * PH + j: the first NT * 2 of its 8 pieces
          glds16(xl + ((size_t)(p * PH + j) * 8 + i) * 1024, slot + (unsigned)(j * NT * 2 + i) * 1024u);
      glds16(sl + (size_t)p * 1024, slot + XPH);
    };
    issue(0);
    if (nph > 1) issue(1);
    for (int p = 0; p < nph; ++p) {
      if (p + 1 < nph) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();   // phase p has landed (phase p + 1 may still be in flight)
      asm volatile("s_barrier" ::: "memory");
      if (p + 2 < nph) issue(p + 2);                                // its ring slot held phase p - 1: every compute wave is past it
    }
    return;
  }

  // ---- compute waves
  const int ngroups = gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH), ntiles = (a.N + 15) >> 4;
  const int g = blockIdx.x * NC + wave;
  const int gc = min(g, ngroups - 1);                               // a surplus wave walks a valid unit and drops it (the barriers must match)
  const unsigned char* wp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = min(gg_tile_row0<EPI, 2>(a, gc, t) >> 4, ntiles - 1);
    wp[t] = a.Wm + (size_t)tile * nks * 2048 + lane * 16;
  }
  u32x4 wa[2][PH][2], wb[WD == 2 ? 2 : 1][WD == 2 ? PH : 1][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < PH; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) wa[t][j][h] = ld_nt(reinterpret_cast<const u32x4*>(wp[t] + (size_t)(j * 2 + h) * 1024));
  if (WD == 2 && nph > 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < PH; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) wb[WD == 2 ? t : 0][WD == 2 ? j : 0][h] = ld_nt(reinterpret_cast<const u32x4*>(wp[t] + (size_t)((PH + j) * 2 + h) * 1024));
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto phase = [&](int p, auto& w, auto refill_tag) {
    constexpr bool REFILL = decltype(refill_tag)::value;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned char* xb = smem + (unsigned)(p % R) * PHB + lane * 16;
    unsigned sd[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) sd[nt] = *reinterpret_cast<const unsigned*>(smem + (unsigned)(p % R) * PHB + XPH + (unsigned)nt * 256u + lane * 4);
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      i32x8 bf[NT];
      unsigned sc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(xb + (size_t)((j * NT + nt) * 2) * 1024);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(xb + (size_t)((j * NT + nt) * 2 + 1) * 1024);
        bf[nt] = mx_op(lo, hi);
        sc[nt] = __builtin_amdgcn_perm(sd[nt], sd[nt], 0x01010101u * (unsigned)j);      // byte j (= ks & 3) in every byte
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const i32x8 af = mx_op(w[t][j][0], w[t][j][1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mx_mfma(af, bf[nt], acc[t][nt], sc[nt]);
        if (REFILL) {
#pragma unroll
          for (int h = 0; h < 2; ++h) w[t][j][h] = ld_nt(reinterpret_cast<const u32x4*>(wp[t] + (size_t)(((p + WD) * PH + j) * 2 + h) * 1024));
        }
      }
    }
  };
  constexpr mx_flag<true> yes{};
  constexpr mx_flag<false> no{};
  if constexpr (WD == 2) {
    int p = 0;
    for (; p + 3 < nph; p += 2) { phase(p, wa, yes); phase(p + 1, wb, yes); }
    const int rest = nph - p;                                         // 1, 2 or 3 phases left
    if (rest == 3) { phase(p, wa, yes); phase(p + 1, wb, no); phase(p + 2, wa, no); }
    else if (rest == 2) { phase(p, wa, no); phase(p + 1, wb, no); }
    else phase(p, wa, no);
  } else {
    for (int p = 0; p + 1 < nph; ++p) phase(p, wa, yes);
    phase(nph - 1, wa, no);
  }
  if (g >= ngroups) return;
  if (EPI == EPI_SWIGLU) mx_swiglu_finish<NT>(a, g, acc, lane);
  else gg_finish_unit<EPI, 2, true, NT>(a, g, acc, lane);
}

template <int EPI, int NC, int NT>
static void launch_mxu_one(const GemvBArgs& a, int units, hipStream_t s) {
  constexpr int lds = 3 * (4 * NT * 2048 + 1024);
  static unsigned long long attr = 0;
  if (dtk_lds_attr_todo(attr)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_mxu<EPI, NC, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  hipLaunchKernelGGL((k_gemv_mxu<EPI, NC, NT>), dim3((units + NC - 1) / NC), dim3((NC + 1) * 64), lds, s, a);
}
static int g_mx_nc[3] = {0, 0, 0};                                    // compute waves per block by role (qkv, gate/up, lm_head); 0 = from the CU count
void set_mx_nc(int role, int nc) { if (role >= 0 && role < 3) g_mx_nc[role] = nc < 0 ? 0 : (nc > 4 ? 4 : nc); }
static int mx_cu_count() {
  static int n = 0;
  if (!n) { int dev = 0; hipDeviceProp_t p; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
  return n;
}
template <int EPI, int NT>
static void launch_mxu_nt(const GemvBArgs& a, hipStream_t s) {
  const int units = gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH);
  int nc = g_mx_nc[EPI == EPI_QKV ? 0 : (EPI == EPI_SWIGLU ? 1 : 2)];
  if (!nc) { nc = (units + mx_cu_count() - 1) / mx_cu_count(); if (nc > 4) nc = 4; }   // one block per CU where the units allow: x is read once per CU
  if (nc <= 1) launch_mxu_one<EPI, 1, NT>(a, units, s);
  else if (nc == 2) launch_mxu_one<EPI, 2, NT>(a, units, s);
  else if (nc == 3 || (EPI == EPI_QKV && NT == 4)) launch_mxu_one<EPI, 3, NT>(a, units, s);   // (the RoPE epilogue of five waves x 4 tiles would spill: not built)
  else launch_mxu_one<EPI, (EPI == EPI_QKV && NT == 4) ? 3 : 4, NT>(a, units, s);
}
template <int EPI>
static void launch_mxu_epi(const GemvBArgs& a, hipStream_t s) {
  if (a.nt >= 3) launch_mxu_nt<EPI, 4>(a, s);
  else if (a.nt == 2) launch_mxu_nt<EPI, 2>(a, s);
  else launch_mxu_nt<EPI, 1>(a, s);
}
bool mx_unit_covers(int K) { return K > 0 && (K & 511) == 0; }
void launch_gemv_mxu(int epi, const GemvBArgs& a, hipStream_t s) {   // a.Wm, a.X8, a.XS (G = 32); SWIGLU: a.Y8 / a.YS (G = 16)
  if (epi == EPI_QKV) launch_mxu_epi<EPI_QKV>(a, s);
  else if (epi == EPI_SWIGLU) launch_mxu_epi<EPI_SWIGLU>(a, s);
  else launch_mxu_epi<EPI_LOGITS>(a, s);
}

// ---------------------------------------------------------------------------------------------------------------- N = d roles
// o_proj (G = 32) and down (G = 16) in k_gemv_bkp's form: block = (row group of TPG row tiles, one of 8 K slices), a wave per row
// tile; the 16 x (NT x 16) fp32 partial of the slice is stored and the RMSNorm kernel that follows the role anyway adds the 8
// partials + the residual (k_resid_norm_b, unchanged arithmetic).  A slice is small enough to be in flight ALL AT ONCE: the wave's
// weights (<= 28 KiB) go straight to registers, the slice's x (<= 112 KiB for 64 slots) + its scales by LDS-DMA into LDS, every
// wave issuing its share; then one drain + barrier and nothing but LDS reads and MFMAs.  No ring, no flags.
template <int TPG, int NT, int G, int MAXL>
__global__ __launch_bounds__(TPG * 64) void k_gemv_mxk(GemvBArgs a) {
  constexpr int SPQ = G == 32 ? 4 : 2;                               // (pair) steps per 1 KiB scale row
  constexpr int MAXQ = (MAXL + SPQ - 1) / SPQ + 1;
  constexpr unsigned XB = (unsigned)MAXL * NT * 2048u;               // scales sit behind the x pieces
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nks = a.K >> 7, per = (nks + 7) >> 3;
  const int b = blockIdx.x, idx = b >> 3;
  const int rgs_per_xcd = (int)(gridDim.x >> 6);                    // grid = 8 XCDs x rgs_per_xcd row groups x 8 slices
  const int rg = (b & 7) * rgs_per_xcd + (idx >> 3), ks = idx & 7;
  const int s0 = min(nks, ks * per), s1 = min(nks, s0 + per);
  const int Lc = s1 - s0;                                           // 1 .. MAXL (launcher)
  const int q0 = s0 / SPQ, nq = (s1 - 1) / SPQ - q0 + 1;            // scale rows the slice touches
  const int tn = rg * TPG + wave;
  const unsigned char* wrow = a.Wm + ((size_t)tn * nks + s0) * 2048 + lane * 16;
  u32x4 w[MAXL][2];
#pragma unroll
  for (int j = 0; j < MAXL; ++j)
    if (j < Lc) {
#pragma unroll
      for (int h = 0; h < 2; ++h) w[j][h] = ld_nt(reinterpret_cast<const u32x4*>(wrow + (size_t)(j * 2 + h) * 1024));
    }
  {
    const unsigned char* xl = a.X8 + lane * 16;
    const int per_step = NT * 2, npieces = Lc * per_step;           // (tile, half) pieces of a step are the first NT * 2 of its 8
    for (int i = wave; i < npieces; i += TPG) {
      const int j = i / per_step, r = i - j * per_step;
      glds16(xl + ((size_t)(s0 + j) * 8 + r) * 1024, (unsigned)i * 1024u);
    }
    const unsigned char* sl = a.XS + lane * 16;
    for (int q = wave; q < nq; q += TPG) glds16(sl + (size_t)(q0 + q) * 1024, XB + (unsigned)q * 1024u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 c[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) c[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const u32x4 zero = {0u, 0u, 0u, 0u};
  const bool odd = (lane >> 4) & 1;                                 // G = 16: this lane's weights belong to instruction s = 1
#pragma unroll
  for (int j = 0; j < MAXL; ++j) {
    if (j < Lc) {                                                   // block-uniform
      const int kk = s0 + j;
      const unsigned q = (unsigned)(kk / SPQ - q0);
      i32x8 xf[NT];
      unsigned sdw[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned char* xp = smem + (size_t)((j * NT + nt) * 2) * 1024 + lane * 16;
        xf[nt] = mx_op(*reinterpret_cast<const u32x4*>(xp), *reinterpret_cast<const u32x4*>(xp + 1024));
        sdw[nt] = *reinterpret_cast<const unsigned*>(smem + XB + q * 1024u + (unsigned)nt * 256u + lane * 4);
      }
      if (G == 32) {
        const unsigned sel = 0x01010101u * (unsigned)(kk & 3);
        const i32x8 af = mx_op(w[j][0], w[j][1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[nt] = mx_mfma(af, xf[nt], c[nt], __builtin_amdgcn_perm(sdw[nt], sdw[nt], sel));
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {                               // the 64 real k of this half of the pair step
          const unsigned sel = 0x01010101u * (unsigned)((kk & 1) * 2 + s);
          const bool live = odd == (s == 1);
          const i32x8 af = mx_op(live ? w[j][0] : zero, live ? w[j][1] : zero);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) c[nt] = mx_mfma(af, xf[nt], c[nt], __builtin_amdgcn_perm(sdw[nt], sdw[nt], sel));
        }
      }
    }
  }
  const f32x4 sc = *reinterpret_cast<const f32x4*>(a.wscale + tn * 16 + (lane >> 4) * 4);     // per-row power-of-two scale: exact
  float* out = a.kpart + ((size_t)ks * 64 + (lane & 15)) * a.N + tn * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(out + (size_t)nt * 16 * a.N) = c[nt] * sc;
}

template <int TPG, int NT, int G, int MAXL>
static void launch_mxk_one(const GemvBArgs& a, hipStream_t s) {
  constexpr int SPQ = G == 32 ? 4 : 2;
  constexpr int lds = (MAXL * NT * 2 + (MAXL + SPQ - 1) / SPQ + 1) * 1024;
  static unsigned long long attr = 0;
  if (dtk_lds_attr_todo(attr)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_mxk<TPG, NT, G, MAXL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  const int ntiles = a.N >> 4;
  hipLaunchKernelGGL((k_gemv_mxk<TPG, NT, G, MAXL>), dim3((ntiles / TPG) * 8), dim3(TPG * 64), lds, s, a);
}
template <int TPG, int NT>
static bool launch_mxk_g(const GemvBArgs& a, int G, int per, hipStream_t s) {
  if (G == 32) {
    if (per <= 4) launch_mxk_one<TPG, NT, 32, 4>(a, s); else if (per <= 8) launch_mxk_one<TPG, NT, 32, 8>(a, s); else return false;
  } else {
    if (per <= 6) launch_mxk_one<TPG, NT, 16, 6>(a, s); else if (per <= 11) launch_mxk_one<TPG, NT, 16, 11>(a, s);
    else if (per <= 14) launch_mxk_one<TPG, NT, 16, 14>(a, s); else return false;
  }
  return true;
}
static inline int mx_tpg(int ntiles) { return (ntiles % 8 == 0 && ntiles / 8 >= 32) ? 8 : 4; }   // row tiles (waves) per block: 256 blocks where the width allows
// N = d role with K-slice partials: N / 16 row tiles in 8 k row groups of 8 (or 4) tiles, 8 non-empty K slices of <= 8 (G = 32) /
// <= 14 (G = 16) steps of 128 k, a width k_resid_norm_b handles
bool mx_kparts_covers(int N, int K, int G) {
  if (N <= 0 || K <= 0 || (N & 127) || (K & 127)) return false;
  const int nks = K >> 7, per = (nks + 7) >> 3;
  if (7 * per >= nks || per > (G == 32 ? 8 : 14)) return false;
  const int ntiles = N >> 4, tpg = mx_tpg(ntiles);
  if ((ntiles % tpg) || (ntiles / tpg) % 8) return false;
  const int D8 = N >> 3;
  return D8 == 256 || D8 == 512 || D8 == 1024;
}
bool launch_gemv_mxk(const GemvBArgs& a, int G, hipStream_t s) {      // a.Wm, a.X8, a.XS, a.kpart, a.wscale; false = no kernel for this shape, NOTHING launched
  const int nks = a.K >> 7, per = (nks + 7) >> 3;
  const int ntiles = a.N >> 4;
  const bool t8 = mx_tpg(ntiles) == 8;
  bool ok;
  if (a.nt >= 3) ok = t8 ? launch_mxk_g<8, 4>(a, G, per, s) : launch_mxk_g<4, 4>(a, G, per, s);
  else if (a.nt == 2) ok = t8 ? launch_mxk_g<8, 2>(a, G, per, s) : launch_mxk_g<4, 2>(a, G, per, s);
  else ok = t8 ? launch_mxk_g<8, 1>(a, G, per, s) : launch_mxk_g<4, 1>(a, G, per, s);
  return ok;                                                         // (mx_kparts_covers is asked at dtk_create; a disagreement must fail the step, not skip a projection)
}
