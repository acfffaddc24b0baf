// kernels_batch_ks.hip — k_gemv_bus: the rows >> d roles (qkv, gate/up) of the 64-slot decode step with every wave FREE-RUNNING: no
// loader wave, no phase hand-offs, no flags, no LDS in the k loop (round 6).  Hand-issued loads with counted vmcnt (the compiler's waitcnt
// pass would wait for the whole ring).  The same file held k_gemv_bks, the same idea for the K-slice partial kernel of the N = d roles
// (x chunks of 16 k-steps staged in LDS by ALL waves, a per-wave weight ring): bit-identical and 15 -> 12.7 us per launch under rocprofv3,
// but NOT faster end to end once the contexts grow (200-step runs, profiles/r06n2_step_bench.txt: cl-7b fp8 + 0.5 %, ds-7b + 2.5 %;
// with 32-k-step chunks + 5 %) — removed again; DESIGN §8 has the numbers.
#include "kernels.h"
#include "batch_epi.h"

namespace {

template <int V> struct ks_ic { static constexpr int value = V; };
template <bool B> struct ks_flag { static constexpr bool value = B; };
template <int I, int N, class F>
__device__ __forceinline__ void ks_for(F&& f) {
  if constexpr (I < N) { f(ks_ic<I>{}); ks_for<I + 1, N>(f); }
}
__device__ __forceinline__ void ks_load_nt(u32x4& dst, const void* src) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(src)); }
__device__ __forceinline__ void ks_load(u32x4& dst, const void* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); }
template <int N>
__device__ __forceinline__ void ks_wait(u32x4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }   // ties the register to the wait
__device__ __forceinline__ void ks_tie(u32x4& x) { asm volatile("" : "+v"(x)); }                                 // ... and further registers the same wait covers
__device__ __forceinline__ void ks_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }  // LDS traffic only: the weight ring stays in flight
// two e4m3 words (8 weights of one row) -> the bf16 A fragment of one k-step (exact)
__device__ __forceinline__ bf16x8_t ks_f8x8_to_bf16x8(uint32_t w0, uint32_t w1) {
  u32x4 o;
  o[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, false));
  o[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, true));
  o[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, false));
  o[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, true));
  return __builtin_bit_cast(bf16x8_t, o);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// k_gemv_bus — the rows >> d roles (qkv, gate/up) at 49..64 slots: one block per CU, its 8 waves = k_gemv_b's 8 K slices, and
// EVERY load a wave issues is its own.
//
// What tools/probe/xbw_probe measured (profiles/r06k2_xbw_probe.txt): all 256 CUs at once pull the 64 slots' x (512 KiB, L2 hits)
// at 80-145 GB/s per CU when every wave loads its share — 3.2 us on top of a launch, not the 13-20 us the one loader wave of
// k_gemv_bx / bl / br / bc needs for it (25-40 GB/s: that wave, not the x bytes, was the bound of those kernels) — and a weight
// stream issued by 8 waves of a block per CU runs at 6.6 TB/s after 4.8 us of launch + ramp (17 / 50 / 90 / 180 MB: 7.3 / 12.2 /
// 17.9 / 32.3 us).  So:
//   * wave w owns K slice w (k_gemv_b's slices: nsteps / 8 k-steps) of ALL the block's row tiles — a RoPE pair unit + a V row tile
//     (qkv of an MHA model: 3 tiles per CU), up to three gate / up pair units — and walks its k-steps once: per k-step the four
//     1 KiB x fragments of the 64 slots (needed by this wave ONLY: no LDS, no barrier, nothing shared) and one weight tile per row
//     tile, all straight into registers, a ring of 2-4 k-steps deep, hand-issued with counted vmcnt (~96 KiB of weights in flight
//     per CU);
//   * the slice sums meet in LDS once, at the end: every wave parks its accumulators, the (unit, column tile) owners add the
//     eight partials in slice order from zero and run the epilogue of gg_pre_store with the operands it reads requested beforehand.
// Per accumulator the same MFMA chain over the same slice and the same order of slice sums as k_gemv_b / bx / bl / br / bc:
// BIT-IDENTICAL to them (tests/test_gpu_parity.py, the variants table).  x traffic: one pass per CU, as in those kernels.
template <int EPI, int TU, int VT, bool F8, int PER>      // TU pair units + VT single (V) row tiles per block; PER = k-steps per slice (16: K = 4096, 8: K = 2048)
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemv_bus(GemvBArgs a, int nblk_v0) {
  constexpr int NT = 4, NTILE = 2 * TU + VT;
  constexpr int KG = F8 ? 2 : 1;                                  // k-steps per load group (an fp8 weight tile holds two)
  constexpr int NG = PER / KG;                                    // groups per slice
  constexpr int LG = KG * NT + NTILE;                             // loads per group
  constexpr int D0 = F8 ? (NTILE > 4 ? 2 : 3) : (NTILE > 4 ? 2 : 4);     // (qkv, fp8: 2 / 3 / 4 groups measured 3.73 / 3.70 / 3.70-3.77 ms per step)
  constexpr int D = D0 < NG ? D0 : NG;                            // groups in flight
  static_assert((D - 1) * LG <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the reduction's staging: [tile][slice][column tile] x 1 KiB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = a.K >> 5;                                    // = 8 * PER (launcher)
  const int b = blockIdx.x, G = gridDim.x;
  const int units = EPI == EPI_QKV ? (a.H + a.KVH) * 4 : gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH);
  // ---- the block's row tiles
  int ug[TU > 0 ? TU : 1];                                        // its units (clamped: a surplus unit streams valid memory and stores nothing)
  bool uok[TU > 0 ? TU : 1];
  int tn[NTILE];
#pragma unroll
  for (int u = 0; u < TU; ++u) {
    const int g = b + u * G;
    uok[u] = g < units; ug[u] = uok[u] ? g : units - 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) tn[2 * u + t] = gg_tile_row0<EPI, 2>(a, ug[u], t) >> 4;
  }
  const int nv = a.KVH * 8;
  const int vraw = b - nblk_v0;                                   // the V row tile of this block (qkv): block nblk_v0 + v owns tile v
  const bool vok = VT && vraw >= 0 && vraw < nv;
  const int vt = vok ? vraw : 0;
  if (VT) tn[2 * TU] = (a.H + a.KVH) * 8 + vt;
  const unsigned char* wb[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
    wb[t] = F8 ? a.W8 + (((size_t)tn[t] * (nsteps >> 1) + (size_t)wave * (PER / 2)) * 64 + lane) * 16
               : reinterpret_cast<const unsigned char*>(a.W) + (((size_t)tn[t] * nsteps + (size_t)wave * PER) * 64 + lane) * 16;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.X) + ((size_t)wave * PER * 64 + lane) * 16;   // column tile nt at + nt * nsteps KiB

  // what this wave's epilogue will read — active flags, positions, RoPE entries (a load that depends on the position load), fp8 row
  // scales — requested NOW, ahead of the stream (k_gemv_bc's lesson: two or three dependent round trips behind the last MFMA were
  // 4.8 us per launch): wave w finishes column tile w & 3 of item w >> 2 of every round
  constexpr int ROUNDS = (NTILE + 3) / 4;
  gg_pre<EPI, 2, F8> pre[ROUNDS];
  int vact = 0, vpos = 0;
  f32x4 vsc = (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int tg = rd * 4 + (wave >> 2) * 2, col = wave & 3;
    if (tg < 2 * TU) gg_pre_load<EPI, 2, F8>(a, ug[tg >> 1], lane, col, pre[rd]);
    else if (VT && tg == 2 * TU) {
      const int n = col * 16 + (lane & 15);
      vact = a.bs->active[n]; vpos = a.st[n].pos;
      if (F8) vsc = *reinterpret_cast<const f32x4*>(a.wscale + (a.H + a.KVH) * 128 + vt * 16 + (lane >> 4) * 4);
    }
  }
  asm volatile("" ::: "memory");                                  // (the requests stay above the stream)
  u32x4 xr[D][KG * NT], wr[D][NTILE];
  auto issue = [&](auto Gi) {
    constexpr int g = decltype(Gi)::value, r = g % D;
#pragma unroll
    for (int j = 0; j < KG; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) ks_load(xr[r][j * NT + nt], xb + ((size_t)nt * nsteps + (size_t)(g * KG + j)) * 1024);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) ks_load_nt(wr[r][t], wb[t] + (size_t)g * 1024);
  };
  ks_for<0, D>(issue);
  f32x4 acc[NTILE][NT];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  ks_for<0, NG>([&](auto Gi) {
    constexpr int g = decltype(Gi)::value, r = g % D;
    constexpr int younger = (NG - 1 - g < D - 1 ? NG - 1 - g : D - 1) * LG;      // the groups issued after this one
    ks_wait<younger>(xr[r][0]);
#pragma unroll
    for (int i = 1; i < KG * NT; ++i) ks_tie(xr[r][i]);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) ks_tie(wr[r][t]);
#pragma unroll
    for (int j = 0; j < KG; ++j)
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const bf16x8_t af = F8 ? ks_f8x8_to_bf16x8(wr[r][t][2 * j], wr[r][t][2 * j + 1]) : __builtin_bit_cast(bf16x8_t, wr[r][t]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, xr[r][j * NT + nt]), acc[t][nt], 0, 0, 0);
      }
    if constexpr (g + D < NG) issue(ks_ic<g + D>{});
  });

  // ---- the eight slice sums of every accumulator meet in LDS; rounds of up to four row tiles (128 KiB).  Wave w owns column tile w & 3 of
  // item w >> 2 of a round (items: the pair units, then the V tile); what its epilogue READS was requested before the k loop (below).
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int t0 = rd * 4, t1 = (rd * 4 + 4 < NTILE) ? rd * 4 + 4 : NTILE;     // tiles of this round
    if (rd > 0) ks_barrier();                                     // the previous round's reads are done
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      if (t < t0 || t >= t1) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(smem + ((size_t)(((t - t0) * 8 + wave) * NT + nt)) * 1024 + lane * 16) = acc[t][nt];
    }
    const int tl = (wave >> 2) * 2, tg = t0 + tl, col = wave & 3;   // the item's first tile: of the round, of the block
    const bool unit_item = tg < 2 * TU, have = tg < t1;
    ks_barrier();
    if (!have) continue;
    const int ntl = unit_item ? 2 : 1;
    f32x4 sum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t < ntl) {
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)                            // slice order from zero: k_gemv_b's reduction
          sum[t] += *reinterpret_cast<const f32x4*>(smem + ((size_t)(((tl + t) * 8 + sl) * NT + col)) * 1024 + lane * 16);
      }
    }
    if (unit_item) {
      if (uok[tg >> 1]) gg_pre_store<EPI, 2, F8>(a, ug[tg >> 1], sum, lane, col, pre[rd]);
    } else if (VT && vok && vact) {                               // V row tile vt: dims (vt & 7) * 16 .. of V head vt >> 3
      const int n = col * 16 + (lane & 15);
      bf16_t* dst = a.vcache + (size_t)n * a.kv_slot_stride + ((size_t)(vt >> 3) * a.T_max + vpos) * 128 + (vt & 7) * 16 + (lane >> 4) * 4;
      const f32x4 v = sum[0] * vsc;
      *reinterpret_cast<u32x2*>(dst) = (u32x2){pack2(rbf(v[0]), rbf(v[1])), pack2(rbf(v[2]), rbf(v[3]))};
    }
  }
}

static int g_gemv_bus = -1;
void set_gemv_bus(int v) { g_gemv_bus = v; }
template <int EPI, int TU, int VT, bool F8, int PER>
static bool bus_usable(int lds) {
  static int usable[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  if (!usable[dev]) {
    const void* fn = reinterpret_cast<const void*>(&k_gemv_bus<EPI, TU, VT, F8, PER>);
    hipFuncAttributes fa;
    const bool ok = hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.localSizeBytes == 0
                    && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    usable[dev] = ok ? 1 : 2;
  }
  return usable[dev] == 1;
}
template <int EPI, int TU, int VT, int PER>
static bool launch_bus_one(const GemvBArgs& a, int grid, int nblk_v0, hipStream_t s) {
  constexpr int ntile = 2 * TU + VT;
  constexpr int lds = (ntile > 4 ? 4 : ntile) * 8 * 4 * 1024;
  if (a.W8) { if (!bus_usable<EPI, TU, VT, true, PER>(lds)) return false; hipLaunchKernelGGL((k_gemv_bus<EPI, TU, VT, true, PER>), dim3(grid), dim3(512), lds, s, a, nblk_v0); }
  else { if (!bus_usable<EPI, TU, VT, false, PER>(lds)) return false; hipLaunchKernelGGL((k_gemv_bus<EPI, TU, VT, false, PER>), dim3(grid), dim3(512), lds, s, a, nblk_v0); }
  return true;
}
static int bus_cu_count() {
  static int n = 0;
  if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
  return n;
}
// variant: 0 off; bit 0 qkv, bit 1 gate/up; 128 (default) = per role and weight format by measurement.  false = not covered (fewer than 49 slots,
// other roles, K other than 2048 / 4096, a ragged ff, a head layout the block map does not cover): the caller goes on to k_gemv_bc / ...
bool launch_gemv_bus(int epi, const GemvBArgs& a, hipStream_t s) {
  if (g_gemv_bus < 0) { const char* e = getenv("DTK_GEMV_BUS"); g_gemv_bus = e ? atoi(e) : 128; }
  if (g_gemv_bus <= 0 || a.nt < 3) return false;
  if (epi != EPI_QKV && epi != EPI_SWIGLU) return false;
  const int roles = (g_gemv_bus & 128) ? (a.W8 ? 3 : 1) : (g_gemv_bus & 3);
  if (!(roles & (epi == EPI_QKV ? 1 : 2))) return false;
  if (a.K != 4096 && a.K != 2048) return false;
  if ((g_gemv_bus & 128) && a.K != 4096) return false;      // d = 2048 (ds-1.3b): 1.5 row tiles per CU, the block map leaves half of every block idle — measured slower than k_gemv_bc
  const int cus = bus_cu_count();
  if (epi == EPI_QKV) {
    if (a.N != (a.H + 2 * a.KVH) * 128) return false;
    const int npairs = (a.H + a.KVH) * 4, nv = a.KVH * 8;
    int grid, v0;
    if (npairs + nv <= cus) { grid = npairs + nv; v0 = npairs; }        // a block per pair unit, then a block per V row tile (ds-1.3b: 128 + 128)
    else if (npairs <= cus) { grid = npairs; v0 = 0; }                  // block b: pair unit b and (b < nv) V row tile b (ds-7b: 256 x 3 row tiles)
    else return false;
    return a.K == 4096 ? launch_bus_one<EPI_QKV, 1, 1, 16>(a, grid, v0, s) : launch_bus_one<EPI_QKV, 1, 1, 8>(a, grid, v0, s);
  }
  if (a.ff & 15) return false;
  const int units = a.ff >> 4;
  const int tu = (units + cus - 1) / cus;
  if (tu > 3) return false;
#define BUS_GU(TU_) (a.K == 4096 ? launch_bus_one<EPI_SWIGLU, TU_, 0, 16>(a, (units + TU_ - 1) / TU_, 0, s) : launch_bus_one<EPI_SWIGLU, TU_, 0, 8>(a, (units + TU_ - 1) / TU_, 0, s))
  if (tu == 3) return BUS_GU(3);
  if (tu == 2) return BUS_GU(2);
  return BUS_GU(1);
#undef BUS_GU
}
