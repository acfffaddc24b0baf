// kernels_batched.hip — MFMA-bound rows of the hot path: ViT trunk, MAP pooling head,
// mm_projector and the LLaMA prefill (SURVEY §8 rows a·V, a·C, a·E, a·D-pre).
//
// gemm_mfma: C[M,N] = A[M,K] . W[N,K]^T, both operands K-contiguous ("B^T input"), bf16
// in / fp32 accumulate on v_mfma_f32_16x16x32_bf16.  64x64 block tile (the shapes here are
// M = 243..2048 / 729 rows, so small tiles are what fills 256 CUs), BK = 64, 4 waves each
// owning a 32x32 sub-tile (2x2 MFMA tiles), register-staged global->LDS with the next
// tile's loads issued before the current tile's MFMAs (T14 split), padded LDS rows.
// Epilogue fuses bias / GELU / residual exactly where timm / HF round to bf16.
#include "kernels.h"
#include <stdlib.h>
#include <string.h>


__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  return 0.5f * x * (1.f + tanhf(k0 * (x + k1 * x * x * x)));
}

__device__ __forceinline__ float gemm_epilogue(float acc, int m, int n, const GemmArgs& a) {
  float v = acc;
  if (a.flags & GEMM_BIAS) v += bf2f(a.bias[n]);
  v = rbf(v);  // the Linear's bf16 output tensor
  if (a.flags & GEMM_GELU_ERF) v = rbf(gelu_erf(v));
  else if (a.flags & GEMM_GELU_TANH) v = rbf(gelu_tanh(v));
  if (a.flags & GEMM_RESIDUAL) v = rbf(bf2f(a.residual[(size_t)m * a.ldr + n]) + v);
  return v;
}

// the same epilogue with its memory operands already in registers.  Round 4 found the GEMMs' epilogues — not their main loops —
// to be half of a ViT launch (k_gemm_g3 with neither fills nor MFMAs: 68.7 of 132 us, profiles/r04_vit8_g3_*_kernel_stats.csv):
// every output element loaded its bias / residual value right before use inside a branchy loop, i.e. 16-64 dependent memory
// round trips per lane.  The kernels now issue every epilogue load first (clamped addresses, no branch), then compute, then store.
__device__ __forceinline__ float gemm_epilogue_pre(float acc, float bias, float res, int flags) {
  float v = acc;
  if (flags & GEMM_BIAS) v += bias;
  v = rbf(v);  // the Linear's bf16 output tensor
  if (flags & GEMM_GELU_ERF) v = rbf(gelu_erf(v));
  else if (flags & GEMM_GELU_TANH) v = rbf(gelu_tanh(v));
  if (flags & GEMM_RESIDUAL) v = rbf(res + v);
  return v;
}
// epilogue of the kernels whose lanes hold the standard C/D layout of TM x TN 16 x 16 tiles (col = lane & 15, row = (lane >> 4) * 4 + reg)
template <int TM, int TN>
__device__ __forceinline__ void gemm_store_tiles(const GemmArgs& a, const f32x4 (&acc)[TM][TN], int mbase, int nbase, int lane) {
  const int flags = a.flags;
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(nbase + j * 16 + (lane & 15), a.N - 1);
    bv[j] = (flags & GEMM_BIAS) ? bf2f(a.bias[n]) : 0.f;
  }
  float rv[TM][TN][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rv[i][j][r] = 0.f;
        if (flags & GEMM_RESIDUAL) {
          const int m = min(mbase + i * 16 + (lane >> 4) * 4 + r, a.M - 1), n = min(nbase + j * 16 + (lane & 15), a.N - 1);
          rv[i][j][r] = bf2f(a.residual[(size_t)m * a.ldr + n]);
        }
      }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mbase + i * 16 + (lane >> 4) * 4 + r;
        const int n = nbase + j * 16 + (lane & 15);
        const bf16_t o = f2bf(gemm_epilogue_pre(acc[i][j][r], bv[j], rv[i][j][r], flags));
        if (m < a.M && n < a.N) a.C[(size_t)m * a.ldc + n] = o;
      }
}

// TBM x TBN block tile, 4 waves in a 2 x 2 grid, each wave (TBM/2) x (TBN/2) = TM x TN MFMA tiles: per 32-wide k-step a
// wave reads TM + TN fragments from LDS for TM*TN MFMAs.  64x64 (2+2 reads per 4 MFMAs) is LDS-read-bound; 128x64 and
// 128x128 (4+4 per 16) are not, but need M*N large enough to fill 256 CUs: launch_gemm_mfma picks per shape.  The k order
// per output element is the same for every tile shape, so all variants (and the naive twin) round identically.
template <int TBM, int TBN, int D = 1, int TBK = 64>   // D = k-tiles of global loads kept in flight in registers; TBK = k-tile
__global__ __launch_bounds__(256) void k_gemm_mfma(GemmArgs a) {
  constexpr int TM = TBM / 32, TN = TBN / 32;      // MFMA tiles per wave (2 x 2 wave grid; TBM, TBN >= 32)
  constexpr int CH = TBK / 8;                      // 16-byte chunks per tile row
  constexpr int RPP = 256 / CH;                    // tile rows staged per pass of the 256 threads
  constexpr int AI = TBM / RPP, WI = TBN / RPP;    // 16-byte staging chunks per thread per operand
  constexpr int LDS_STRIDE = TBK + 8;              // bf16 elements per LDS row (+8 pad: 16-byte aligned, conflict-free fragment reads)
  __shared__ __attribute__((aligned(16))) bf16_t As[TBM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[TBN * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware block -> tile map: consecutive block ids go round-robin over the 8 XCDs (each with its own L2), so
  // the MB row-blocks that share one W tile are given ids with the same id % 8, adjacent in that XCD's dispatch order:
  // the W tile is fetched from HBM once and re-used from that XCD's L2 (prefill: M = 243 -> 4 row-blocks; with the
  // plain (n, m) grid every row-block re-read all weights: 4 x 13 GB per prefill).
  const int MB = (a.M + TBM - 1) / TBM, NB = (a.N + TBN - 1) / TBN;
  const int b = blockIdx.x;
  const int nt = (b & 7) + 8 * ((b >> 3) / MB), mb = (b >> 3) % MB;
  if (nt >= NB) return;
  const int m0 = mb * TBM, n0 = nt * TBN;

  // staging map: thread -> (row, 16-byte chunk); rows srow + RPP*i
  const int srow = tid / CH;
  const int schk = tid % CH;  // 8 bf16 each
  const int K = a.K;

  const bf16_t* Ag[AI];
  const bf16_t* Wg[WI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int am = m0 + srow + RPP * i; if (am >= a.M) am = a.M - 1;
    Ag[i] = a.A + (size_t)am * a.lda;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int wn = n0 + srow + RPP * i; if (wn >= a.N) wn = a.N - 1;
    Wg[i] = a.W + (size_t)wn * a.ldw;
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // D register stages: the loads of k-tile t + D are issued as soon as stage t % D has been written to LDS, so D tiles
  // of global-load latency overlap with the MFMAs (shapes with few blocks per CU — N = d GEMMs of the prefill, the
  // N = 1152 GEMMs of the ViT — have nothing else to hide that latency behind).
  u32x4 ra[D][AI], rw[D][WI];
  auto stage_load = [&](u32x4 (&pa)[AI], u32x4 (&pw)[WI], int k0) {
    const int k = k0 + schk * 8;
    const bool ok = k < K;  // K % 8 == 0: a chunk is fully in or fully out
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < AI; ++i) pa[i] = ok ? *reinterpret_cast<const u32x4*>(Ag[i] + k) : z;
#pragma unroll
    for (int i = 0; i < WI; ++i) pw[i] = ok ? *reinterpret_cast<const u32x4*>(Wg[i] + k) : z;
  };
  auto stage_write = [&](const u32x4 (&pa)[AI], const u32x4 (&pw)[WI]) {
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<u32x4*>(&As[(srow + RPP * i) * LDS_STRIDE + schk * 8]) = pa[i];
#pragma unroll
    for (int i = 0; i < WI; ++i) *reinterpret_cast<u32x4*>(&Bs[(srow + RPP * i) * LDS_STRIDE + schk * 8]) = pw[i];
  };

  const int nk = (K + TBK - 1) / TBK;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) stage_load(ra[d], rw[d], d * TBK);
  for (int t0 = 0; t0 < nk; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      if (t >= nk) break;
      __syncthreads();  // previous tile's fragment reads are done
      stage_write(ra[d], rw[d]);
      __syncthreads();
      if (t + D < nk) stage_load(ra[d], rw[d], (t + D) * TBK);  // in flight under the next D tiles of MFMAs
#pragma unroll
      for (int ks = 0; ks < TBK / 32; ++ks) {
        bf16x8_t af[TM], bfr[TN];
        const int kk = ks * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const bf16x8_t*>(&As[(wr * (TBM / 2) + i * 16 + (lane & 15)) * LDS_STRIDE + kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8_t*>(&Bs[(wc * (TBN / 2) + j * 16 + (lane & 15)) * LDS_STRIDE + kk]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  gemm_store_tiles<TM, TN>(a, acc, m0 + wr * (TBM / 2), n0 + wc * (TBN / 2), lane);
}

// ------------------------------------------------------------------------------------------
// k_gemm_dma: the same GEMM with its operand tiles written into LDS by the LDS-DMA path (`global_load_lds_dwordx4`: 1 KiB per
// wave-instruction, no VGPR, no ds_write) in MFMA FRAGMENT order: a 1 KiB LDS block is one (16-row tile, 32-wide k-step) with
// lane l = (row l & 15, k-chunk l >> 4) — exactly what the lane feeds to v_mfma_f32_16x16x32_bf16, so an operand read is one
// conflict-free ds_read_b128 of contiguous memory.  A ring of RING stages (one stage = one 64-wide k-tile of both operands), the
// fills RING - 1 stages ahead, ONE barrier per k-tile.  k_gemm_mfma pays, per k-tile and wave, 4 global loads into VGPRs, 4
// ds_write_b128 (13 cycles each) and two barriers for its 8 MFMAs (128 cycles): 13.8 % MFMA-busy (profiles/r01_pmc_mfma.csv).
// The k order per output element is unchanged (k-steps of 32 in order), so the results equal k_gemm_mfma's bit for bit.
__device__ __forceinline__ void gemm_glds16(const void* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


// ------------------------------------------------------------------------------------------
// k_gemm_glds: the GEMM for shapes with enough tiles to fill the chip at 128 x 128 (batched ViT M = images x 729, long prompts, the
// prefill's N >> d GEMMs).  Round 2 showed that k_gemm_mfma saturates at ~246 TFLOP/s whatever its tile shape (4 waves, one LDS
// k-tile in use, two barriers per k-tile, operands staged through registers in 16 rows x 64 B pieces) and that k_gemm_dma
// (fragment-shaped LDS-DMA fills of the same structure) is no better.  This is the structure guides/cdna_hip_programming.md §5
// measures at 874-912 TFLOP/s on 4096^3 ("step 3" + the glds table's first row):
//   * 128 x 128 block tile, BK = 64, 4 waves in a 2 x 2 grid, each wave 64 x 64 = 4 x 4 MFMA tiles: per 32-wide k-step 8 fragment
//     reads feed 16 MFMAs (k_gemm_mfma's 64 x 64 tile: 4 reads per 4 MFMAs);
//   * BOTH operands go global -> LDS by `global_load_lds_dwordx4` in FULL 128-byte lines: one instruction = 8 tile rows x 128 B
//     (lane l: row l >> 3, 16-byte chunk l & 7), no VGPR, no ds_write;
//   * the LDS image is the plain [row][8 chunks] tile; bank conflicts of the fragment reads (rows 128 B apart: 4-way) are removed
//     by XOR-swizzling the chunk index with (row >> 1) & 7 — applied on the SOURCE address of the fill (LDS-DMA writes are
//     lane-linear) and on the read address (conflict-free for all four ds_read_b128 lane groups, brute-forced);
//   * two LDS stages (2 x 32 KiB, 2 blocks per CU): the fill of k-tile t + 1 is in flight under the MFMAs of k-tile t, ONE barrier
//     per k-tile (vmcnt(0) for the wave's own 8 fills, barrier, issue the next fill, compute).
// A K that is not a multiple of 64 ends with one register-staged, zero-filled tile.  The k order per output element is
// k_gemm_mfma's (32-wide k-steps in order, MFMA accumulation from zero): results are bit-identical to every other GEMM variant.
__device__ __forceinline__ void glds16_row(const void* gsrc, unsigned lds_byte) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__global__ __launch_bounds__(256, 2) void k_gemm_glds(GemmArgs a) {
  constexpr int BM = 128, BN = 128, BK = 64;
  constexpr unsigned OPB = BM * BK * 2;            // bytes of one operand tile (16 KiB)
  constexpr unsigned STB = 2 * OPB;                // one stage: A tile, then W tile
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];   // the kernel's only LDS object: LDS address 0
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int MB = (a.M + BM - 1) / BM, NB = (a.N + BN - 1) / BN;
  const int b = blockIdx.x;
  const int nt = (b & 7) + 8 * ((b >> 3) / MB), mb = (b >> 3) % MB;   // XCD-aware map (see k_gemm_mfma)
  if (nt >= NB) return;
  const int m0 = mb * BM, n0 = nt * BN;
  const int K = a.K;

  // fill map: a stage is 32 pieces of 1 KiB (16 of A, 16 of W); wave w issues pieces 8w .. 8w+7.  Piece p covers tile rows
  // 8 (p & 15) .. +7 of operand p >> 4; lane l fills LDS chunk (row l >> 3, position l & 7) FROM source chunk (l & 7) ^ swz(row)
  const bf16_t* src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = wave * 8 + i, op = p >> 4, row = (p & 15) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (op == 0) { int am = m0 + row; if (am >= a.M) am = a.M - 1; src[i] = a.A + (size_t)am * a.lda + chunk * 8; }
    else { int wn = n0 + row; if (wn >= a.N) wn = a.N - 1; src[i] = a.W + (size_t)wn * a.ldw + chunk * 8; }
  }
  auto fill = [&](int t) {
    const unsigned base = (unsigned)(t & 1) * STB + (unsigned)wave * 8u * 1024u;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16_row(src[i] + (size_t)t * BK, base + (unsigned)i * 1024u);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment read addresses inside a stage: row r of the operand tile, k-chunk q = 4 ks + (lane >> 4)
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ra = (unsigned)(wr * 64 + i * 16 + (lane & 15)), rb = (unsigned)(wc * 64 + i * 16 + (lane & 15));
    aoff[i] = ra * 128u; boff[i] = OPB + rb * 128u;
  }
  const unsigned swz = (unsigned)(((lane & 15) >> 1) & 7);          // (row >> 1) & 7: the tile rows of a fragment start at a multiple of 16
  const unsigned q0 = (unsigned)(lane >> 4);
  auto compute = [&](int stage) {
    const unsigned char* st = gsm + (unsigned)stage * STB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned coff = (((unsigned)ks * 4u + q0) ^ swz) * 16u;
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(st + aoff[i] + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + boff[j] + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = K / BK;                           // full k-tiles (LDS-DMA); K % 64 (a multiple of 8) goes through registers below
  if (nk > 0) fill(0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of k-tile t have landed
    __syncthreads();                                        // ... and everybody's; all waves are done reading stage (t + 1) & 1
    if (t + 1 < nk) fill(t + 1);
    compute(t & 1);
  }
  if (K % BK) {                                    // ragged tail: register-staged, zero-filled, same image (swizzle included)
    const int stage = nk & 1, k0 = nk * BK;
    __syncthreads();                               // stage `stage` was last read by compute(nk - 2): everybody is past it
    unsigned char* st = gsm + (unsigned)stage * STB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                  // 2 x 1024 chunks over 256 threads
      const int c = tid + 256 * i, op = c >> 10, row = (c >> 3) & 127, pos = c & 7;
      const int chunk = pos ^ ((row >> 1) & 7);
      const int k = k0 + chunk * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k < K) {
        if (op == 0) { int am = m0 + row; if (am >= a.M) am = a.M - 1; v = *reinterpret_cast<const u32x4*>(a.A + (size_t)am * a.lda + k); }
        else { int wn = n0 + row; if (wn >= a.N) wn = a.N - 1; v = *reinterpret_cast<const u32x4*>(a.W + (size_t)wn * a.ldw + k); }
      }
      *reinterpret_cast<u32x4*>(st + (unsigned)op * OPB + (unsigned)row * 128u + (unsigned)pos * 16u) = v;
    }
    __syncthreads();
    compute(stage);
  }
  gemm_store_tiles<4, 4>(a, acc, m0 + wr * 64, n0 + wc * 64, lane);
}
static int g_glds_min_tiles = 160;   // 128 x 128 tiles a shape must have for k_gemm_glds (gemm_impl 2 / 3); below that the small-tile kernel fills the chip better
void set_gemm_glds_min_tiles(int v) { g_glds_min_tiles = v; }
static bool launch_gemm_glds(const GemmArgs& a, hipStream_t s) {
  if ((a.lda % 8) || (a.ldw % 8) || a.K < 64 || (a.K % 8)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.W)) & 15) return false;
  const long mbs = (a.M + 127) / 128, nbs = (a.N + 127) / 128;
  if (mbs * nbs < g_glds_min_tiles) return false;
  constexpr int lds = 2 * 2 * 128 * 64 * 2;
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_glds), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  hipLaunchKernelGGL(k_gemm_glds, dim3((unsigned)(8 * ((nbs + 7) / 8) * mbs)), dim3(256), lds, s, a);
  return true;
}

// ------------------------------------------------------------------------------------------
// k_gemm_g3 — 8 waves, 256 x 128 (or 128 x 256) block tile, THREE LDS stages filled by LDS-DMA two k-tiles ahead.
//
// Why k_gemm_glds sits at ~300 TFLOP/s on the batched ViT (M = 5832, K = 1152): a 128 x 128 x 64 k-tile is 32 KiB of operands
// for 2.1 MFLOP (65 FLOP per byte into the CU), two blocks per CU keep ONE fill each in flight = 64 KiB per CU, and an L2 ->
// LDS round trip under load is 2-3 us: 64 KiB / 2.5 us = 26 GB/s per CU = 1.7 TFLOP/s per CU = 430 TFLOP/s before epilogues and
// tile quantisation (measured: 21 GB/s per CU, 341 TFLOP/s on the qkv GEMM).  The operand stream is latency-bound by its
// in-flight depth, not the matrix cores (MFMA-busy 20 %, profiles/r03_pmc_mfma.csv).  This kernel changes both terms:
//   * 256 x 128 tile, BK = 64: 48 KiB per k-tile for 4.2 MFLOP = 87 FLOP per byte (1.33 x);
//   * 3 stages x 48 KiB = 144 KiB of LDS, one block per CU, TWO fills (96 KiB) in flight (1.5 x): counted `s_waitcnt vmcnt(6)`
//     (a wave's six 1 KiB pieces of the NEXT tile may stay outstanding) + a raw `s_barrier` — `__syncthreads()` would drain the
//     fills (guides/cdna_hip_programming.md §5, glds table: "3 LDS buffers, counted vmcnt(N), raw s_barrier");
//   * 8 waves = 2 per SIMD, each a 64 x 64 sub-tile (16 fragment reads per 32 MFMAs and k-tile): one wave's ds_reads and
//     waits hide under the other's MFMAs;
//   * the MFMA runs TRANSPOSED (A operand = W rows, B operand = activation rows), so a lane ends up with 4 consecutive output
//     columns of one row: the epilogue stores 8 bytes per lane instead of four scattered 2-byte values.
// LDS image, swizzle and fill map are k_gemm_glds's ([row][8 x 16 B], chunk ^ ((row >> 1) & 7), applied on the SOURCE address).
// Same k order per output element as every other GEMM here (32-wide k-steps in order from zero).
// SK (sliced-K roles, launch_gemm_sk): a block is (tile, K slice ks) — grid = tiles x kslices, ks = blockIdx % kslices, so that under the
// round-robin dispatch over the 8 XCDs an XCD's L2 holds ONE slice of the activations (kslices = 8; two XCDs per slice at 4) — runs the
// k-tiles of its slice through the same pipeline from zero and stores its fp32 accumulators to part[ks]; k_sk_reduce adds the slices in
// order and applies the epilogue.  The ragged K tail belongs to the last slice.
// WT: the W stage is filled from the fragment-major copy a.Wt (GemmArgs) and read back as whole 1 KiB operands.
// SL (a sliced-K role at an M that fills the chip by its tiles alone, launch_gemm_g3_sliced): ONE block per tile walks all k-tiles and keeps a
// second accumulator set — at every slice boundary tot = tot + acc (the first: tot = acc), acc = 0, an empty slice adds its zeros — i.e. the
// sums k_sk_reduce would have formed from the SK blocks' partials, in the same order: bit-identical to that pair, no partials in memory.
template <int BM, int BN, bool SK = false, bool WT = false, bool SL = false>
__global__ __launch_bounds__(512, 1) void k_gemm_g3(GemmArgs a) {
  constexpr int BK = 64, NST = 3, WAVES = 8;
  constexpr unsigned OPA = BM * BK * 2, OPW = BN * BK * 2, STB = OPA + OPW;   // 48 KiB per stage
  constexpr int PIECES = (int)(STB / 1024), PPW = PIECES / WAVES;              // 48 pieces of 1 KiB, 6 per wave
  constexpr int WN = BN / 64;                                                   // waves along n (the grid of 64 x 64 wave tiles is (BM/64) x (BN/64) = 8)
  static_assert((BM / 64) * (BN / 64) == WAVES && PIECES % WAVES == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];          // the kernel's only LDS object: LDS address 0
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int MB = (a.M + BM - 1) / BM, NB = (a.N + BN - 1) / BN;
  const int b = blockIdx.x;
  int nt, mb, ks = 0, kt0 = 0;
  if (SK) { ks = b % a.kslices; const int tile = b / a.kslices; nt = tile / MB; mb = tile % MB; }
  else { nt = (b & 7) + 8 * ((b >> 3) / MB); mb = (b >> 3) % MB; }             // XCD-aware map (see k_gemm_mfma)
  if (nt >= NB) return;
  const int m0 = mb * BM, n0 = nt * BN;
  const int K = a.K;
  int nk = K / BK;
  if (SK) { const int q = sk_tiles_per_slice(K, a.kslices); kt0 = min(ks * q, nk); nk = min(kt0 + q, nk) - kt0; }
  const bool tail = (K % BK) && (!SK || ks == a.kslices - 1);

  // fill map: piece p covers 8 tile rows of one operand (A: pieces 0 .. BM/8-1, then W); lane l fills LDS chunk (row l >> 3,
  // position l & 7) from source chunk (l & 7) ^ swz(row)
  // WT: W piece pw = p - BM/8 is fragment (row tile pw >> 1, k-step pw & 1) of the k-tile — 1 KiB contiguous in a.Wt, lane-linear in LDS
  const bf16_t* src[PPW];
  const int K32 = (K + 31) >> 5, N16 = (a.N + 15) >> 4;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave * PPW + i;
    const bool isA = p < BM / 8;
    const int row = (isA ? p : p - BM / 8) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (isA) { int am = m0 + row; if (am >= a.M) am = a.M - 1; src[i] = a.A + (size_t)am * a.lda + chunk * 8; }
    else if (WT) { const int pw = p - BM / 8; const int tn = min(n0 / 16 + (pw >> 1), N16 - 1); src[i] = a.Wt + ((size_t)tn * K32 + (pw & 1)) * 512 + lane * 8; }
    else { int wn = n0 + row; if (wn >= a.N) wn = a.N - 1; src[i] = a.W + (size_t)wn * a.ldw + chunk * 8; }
  }
  auto fill = [&](int t) {
    const unsigned base = (unsigned)(t % NST) * STB + (unsigned)wave * (unsigned)PPW * 1024u;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const bool isW = WT && wave * PPW + i >= BM / 8;                         // a fragment-major k-tile is two 1 KiB tiles further on
      glds16_row(src[i] + (size_t)(kt0 + t) * (isW ? 1024 : BK), base + (unsigned)i * 1024u);
    }
  };

  f32x4 acc[4][4];      // acc[i][j]: m tile i, n tile j; register r = column n + r of row m (transposed MFMA)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 tot[4][4];
  int cur = 0;                                     // SL: the slice the accumulators belong to
  const int slq = SL ? sk_tiles_per_slice(K, a.kslices) : 0;
  auto fold = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cur == 0) tot[i][j] = acc[i][j];
        else { tot[i][j][0] += acc[i][j][0]; tot[i][j][1] += acc[i][j][1]; tot[i][j][2] += acc[i][j][2]; tot[i][j][3] += acc[i][j][3]; }
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    ++cur;
  };
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ra = (unsigned)(wr * 64 + i * 16 + (lane & 15)), rb = (unsigned)(wc * 64 + i * 16 + (lane & 15));
    aoff[i] = ra * 128u; boff[i] = WT ? OPA + (unsigned)((wc * 4 + i) * 2) * 1024u + (unsigned)lane * 16u : OPA + rb * 128u;
  }
  const unsigned swz = (unsigned)(((lane & 15) >> 1) & 7);
  const unsigned q0 = (unsigned)(lane >> 4);
  auto compute = [&](int stage) {
    const unsigned char* st = gsm + (unsigned)stage * STB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned coff = (((unsigned)ks * 4u + q0) ^ swz) * 16u;
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(st + aoff[i] + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + boff[j] + (WT ? (unsigned)ks * 1024u : coff));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D[n][m]: rows = W rows, columns = activation rows
    }
  };

  if (nk > 0) fill(0);
  if (nk > 1) fill(1);
  for (int t = 0; t < nk; ++t) {
    // this wave's pieces of k-tile t have landed when only the next tile's PPW pieces may still be outstanding; then everybody's
    // have, and everybody is past compute(t - 1), whose stage the fill below overwrites.  One asm: no LDS access crosses it.
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 2 < nk && !(a.flags & GEMM_PROBE_NOFILL)) fill(t + 2);
    if (SL) { while (cur < a.kslices - 1 && t >= (cur + 1) * slq) fold(); }         // k-tile t opens a later slice
    if (!(a.flags & GEMM_PROBE_NOMFMA)) compute(t % NST);
  }
  if (SL) { while (cur < a.kslices - 1) fold(); }     // the ragged tail (and nothing else, if its tiles ran out) belongs to the last slice
  if (tail) {                                      // ragged tail: register-staged, zero-filled, same image (swizzle included)
    const int stage = nk % NST, k0 = (kt0 + nk) * BK;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // nothing in flight; stage last read by compute(nk - 3)
    unsigned char* st = gsm + (unsigned)stage * STB;
    for (int c = tid; c < (BM + BN) * 8; c += 512) {
      const bool isA = c < BM * 8;
      const int cc = isA ? c : c - BM * 8, row = cc >> 3, pos = cc & 7;
      const int chunk = pos ^ ((row >> 1) & 7);
      const int k = k0 + chunk * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k < K) {
        if (isA) { int am = m0 + row; if (am >= a.M) am = a.M - 1; v = *reinterpret_cast<const u32x4*>(a.A + (size_t)am * a.lda + k); }
        else { int wn = n0 + row; if (wn >= a.N) wn = a.N - 1; v = *reinterpret_cast<const u32x4*>(a.W + (size_t)wn * a.ldw + k); }
      }
      if (WT && !isA) *reinterpret_cast<u32x4*>(st + OPA + (unsigned)((row >> 4) * 2 + (chunk >> 2)) * 1024u + (unsigned)((chunk & 3) * 16 + (row & 15)) * 16u) = v;
      else *reinterpret_cast<u32x4*>(st + (isA ? 0u : OPA) + (unsigned)row * 128u + (unsigned)pos * 16u) = v;
    }
    __syncthreads();
    compute(stage);
  }
  if (SL) {
    fold();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = tot[i][j];
  }
  if (!SK && WT && (a.flags & GEMM_SWIGLU)) {
    // W row tiles arrive as (gate tile q, up tile q) pairs: this wave's tiles j = 0, 2 are gate tiles, j = 1, 3 the up tiles of the same 16
    // channels; a lane holds 4 consecutive channels of output row m.  k_silu_mul's rounding points: both Linear outputs to bf16, SiLU to bf16,
    // the product to bf16.
    const int ff = a.N >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + (lane & 15);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int ch = ((n0 >> 4) + wc * 4 + 2 * jp) / 2 * 16 + (lane >> 4) * 4;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = rbf(acc[i][2 * jp][r]), u = rbf(acc[i][2 * jp + 1][r]);
          const float sl = rbf(g / (1.f + expf(-g)));
          o[r] = sl * u;
        }
        if (m < a.M && ch < ff) { const u32x2 pk = {pack2(o[0], o[1]), pack2(o[2], o[3])}; *reinterpret_cast<u32x2*>(a.C + (size_t)m * a.ldc + ch) = pk; }
      }
    }
    return;
  }
  // transposed C/D layout: column (lane & 15) = row m of the output, rows (lane >> 4) * 4 + r = 4 consecutive output columns n.
  // Epilogue through LDS (default): a lane's direct stores are 8 bytes (16 as a K slice's fp32) at 16 different rows per instruction — 32-byte
  // pieces, 15 of the 77 us of the prefill's gate/up launch.  The stages are free now: every wave parks its 64 x 64 fp32 tile in its own
  // 17 KiB of LDS (row stride 272 B: the 16-byte slots of a store instruction spread evenly over the banks), then walks it row by row — 8
  // lanes per row, 8 consecutive columns each: bias / residual arrive as 16-byte loads, the output leaves as 16-byte stores, 8 rows x 128 B
  // per instruction (a K slice: 32-byte stores, 256 B per row).  Same arithmetic per element.
  if (!(a.flags & GEMM_EPI_DIRECT) && (a.N & 7) == 0 && (SK || ((a.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0))) {
    constexpr unsigned ROWB = 272u, WREG = 64u * ROWB;
    unsigned char* reg = gsm + (unsigned)wave * WREG;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                  // every wave is done with the last stage
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(reg + (unsigned)(i * 16 + (lane & 15)) * ROWB + (unsigned)(j * 16 + (lane >> 4) * 4) * 4u) = acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   // the wave's own tile: no barrier
    const int flags = a.flags;
    const int nl = (lane & 7) * 8, n = n0 + wc * 64 + nl;
    const bool nin = n < a.N;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!SK && (flags & GEMM_BIAS)) {
      const int nb = min(n, a.N - 8);
      if ((reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) {
        const u32x4 b4 = *reinterpret_cast<const u32x4*>(a.bias + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[2 * e] = pk_lo(b4[e]); bv[2 * e + 1] = pk_hi(b4[e]); }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bf2f(a.bias[nb + e]);
      }
    }
    const bool rvec = !SK && (flags & GEMM_RESIDUAL) && (a.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(a.residual) & 15) == 0;
    float* pbase = SK ? a.part + (size_t)ks * (size_t)a.part_stride : nullptr;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ml = it * 8 + (lane >> 3), m = m0 + wr * 64 + ml;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(reg + (unsigned)ml * ROWB + (unsigned)nl * 4u);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(reg + (unsigned)ml * ROWB + (unsigned)nl * 4u + 16u);
      if (SK) {
        if (m < a.M && nin) { float* d = pbase + (size_t)m * a.N + n; *reinterpret_cast<f32x4*>(d) = v0; *reinterpret_cast<f32x4*>(d + 4) = v1; }
        continue;
      }
      float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (flags & GEMM_RESIDUAL) {
        const bf16_t* rp = a.residual + (size_t)min(m, a.M - 1) * a.ldr + min(n, a.N - 8);
        if (rvec) {
          const u32x4 r4 = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
          for (int e = 0; e < 4; ++e) { rv[2 * e] = pk_lo(r4[e]); rv[2 * e + 1] = pk_hi(r4[e]); }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = bf2f(rp[e]);
        }
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = e < 2 ? v0[2 * e] : v1[2 * e - 4], x1 = e < 2 ? v0[2 * e + 1] : v1[2 * e - 3];
        o[e] = (uint32_t)f2bf(gemm_epilogue_pre(x0, bv[2 * e], rv[2 * e], flags)) | ((uint32_t)f2bf(gemm_epilogue_pre(x1, bv[2 * e + 1], rv[2 * e + 1], flags)) << 16);
      }
      if (m < a.M && nin) *reinterpret_cast<u32x4*>(a.C + (size_t)m * a.ldc + n) = o;
    }
    return;
  }
  if (SK) {                                        // the slice's fp32 sums, 16 bytes per lane (N % 4 == 0: launch_gemm_sk)
    float* pbase = a.part + (size_t)ks * (size_t)a.part_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
        if (m < a.M && n < a.N) *reinterpret_cast<f32x4*>(pbase + (size_t)m * a.N + n) = acc[i][j];
      }
    }
    return;
  }
  // All bias / residual loads first (8 bytes each, clamped), then the arithmetic, then 8-byte stores.
  const int flags = a.flags;
  const bool vec = (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 7) == 0;
  const bool nvec = (a.N & 3) == 0;      // n is a multiple of 4: a lane's 4 columns are all inside or all outside the matrix
  const bool rvec = nvec && (flags & GEMM_RESIDUAL) && (a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.residual) & 7) == 0;
  const bool bvec = nvec && (flags & GEMM_BIAS) && (reinterpret_cast<uintptr_t>(a.bias) & 7) == 0;
  float bv[4][4], rv[4][4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
    if (bvec) {
      const u32x2 p2 = *reinterpret_cast<const u32x2*>(a.bias + min(n, a.N - 4));
      bv[j][0] = pk_lo(p2[0]); bv[j][1] = pk_hi(p2[0]); bv[j][2] = pk_lo(p2[1]); bv[j][3] = pk_hi(p2[1]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (flags & GEMM_BIAS) ? bf2f(a.bias[min(n + r, a.N - 1)]) : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + wr * 64 + i * 16 + (lane & 15), a.M - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      if (rvec) {
        const u32x2 p2 = *reinterpret_cast<const u32x2*>(a.residual + (size_t)m * a.ldr + min(n, a.N - 4));
        rv[i][j][0] = pk_lo(p2[0]); rv[i][j][1] = pk_hi(p2[0]); rv[i][j][2] = pk_lo(p2[1]); rv[i][j][3] = pk_hi(p2[1]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[i][j][r] = (flags & GEMM_RESIDUAL) ? bf2f(a.residual[(size_t)m * a.ldr + min(n + r, a.N - 1)]) : 0.f;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      const uint32_t o0 = f2bf(gemm_epilogue_pre(acc[i][j][0], bv[j][0], rv[i][j][0], flags));
      const uint32_t o1 = f2bf(gemm_epilogue_pre(acc[i][j][1], bv[j][1], rv[i][j][1], flags));
      const uint32_t o2 = f2bf(gemm_epilogue_pre(acc[i][j][2], bv[j][2], rv[i][j][2], flags));
      const uint32_t o3 = f2bf(gemm_epilogue_pre(acc[i][j][3], bv[j][3], rv[i][j][3], flags));
      bf16_t* dst = a.C + (size_t)m * a.ldc + n;
      if (m < a.M && n < a.N) {
        if (vec && n + 3 < a.N) {
          const u32x2 pk = {o0 | (o1 << 16), o2 | (o3 << 16)};
          *reinterpret_cast<u32x2*>(dst) = pk;
        } else {
          dst[0] = (bf16_t)o0;
          if (n + 1 < a.N) dst[1] = (bf16_t)o1;
          if (n + 2 < a.N) dst[2] = (bf16_t)o2;
          if (n + 3 < a.N) dst[3] = (bf16_t)o3;
        }
      }
    }
  }
}
static int g_g3_epi_direct = 0;       // 1: k_gemm_g3's lanes store their accumulators directly (dtk_set_option "gemm_epi_direct"; bit-identical)
void set_gemm_epi_direct(int v) { g_g3_epi_direct = v; }
static int g_g3_min_blocks = 128;     // blocks a shape must give k_gemm_g3 (one per CU): below that the smaller tiles fill the chip better
void set_gemm_g3_min_blocks(int v) { g_g3_min_blocks = v; }
static bool use_wt(const GemmArgs& a);
template <int BM, int BN, bool WT = false>
static void launch_gemm_g3_t(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * 64 * 2;
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_g3<BM, BN, false, WT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  const long mbs = (a.M + BM - 1) / BM, nbs = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_gemm_g3<BM, BN, false, WT>), dim3((unsigned)(8 * ((nbs + 7) / 8) * mbs)), dim3(512), lds, s, a);
}
static bool launch_gemm_g3(const GemmArgs& a, hipStream_t s) {
  if ((a.lda % 8) || (a.ldw % 8) || a.K < 128 || (a.K % 8)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.W)) & 15) return false;
  int cus = 256;
  { static int n = 0; if (!n) { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount; if (n <= 0) n = 256; } cus = n; }
  // orientation: the one that needs fewer rounds of one block per CU; ties go to the tall tile (activations are the larger operand)
  auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  const long tall = blocks(256, 128), wide = blocks(128, 256);
  const long rt = (tall + cus - 1) / cus, rw = (wide + cus - 1) / cus;
  const bool use_wide = rw < rt || (rw == rt && wide < tall && a.M < 256);
  if ((use_wide ? wide : tall) < g_g3_min_blocks) return false;
  static int probe = -1;          // DTK_G3_PROBE: 1 = no fills after the first two k-tiles, 2 = no MFMAs (timing experiments: wrong results)
  if (probe < 0) { const char* e = getenv("DTK_G3_PROBE"); probe = e ? atoi(e) : 0; }
  GemmArgs b = a;
  if (g_g3_epi_direct) b.flags |= GEMM_EPI_DIRECT;
  if (probe & 1) b.flags |= GEMM_PROBE_NOFILL;
  if (probe & 2) b.flags |= GEMM_PROBE_NOMFMA;
  if (use_wt(b)) { if (use_wide) launch_gemm_g3_t<128, 256, true>(b, s); else launch_gemm_g3_t<256, 128, true>(b, s); }
  else if (use_wide) launch_gemm_g3_t<128, 256>(b, s); else launch_gemm_g3_t<256, 128>(b, s);
  return true;
}

// ------------------------------------------------------------------------------------------
// Sliced-K GEMM: the prefill's N = d roles (o_proj, down) give the 256 x 128 tile 32 blocks for 256 CUs and q/k/v 96; the small tiles that
// filled the chip instead ran them at 230-390 TFLOP/s (down_proj: 90 MB of weights in 96 us).  Here every CU gets a block of the big tile:
// (tile, K slice), fp32 partials, one reduction pass that is also the role's epilogue and the RMSNorm behind it.
static int g_sk_force_wide = -1;     // DTK_SK_TILE: 0 = 256 x 128, 1 = 128 x 256, unset = by M (<= 128 rows: the wide tile wastes fewer activation fills)
void set_gemm_sk_tile(int v) { g_sk_force_wide = v; }
bool gemm_sk_supported(const GemmArgs& a) {
  if ((a.lda % 8) || (a.ldw % 8) || a.K < 128 || (a.K % 8) || (a.N % 4) || a.kslices < 1 || a.kslices > 8) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.W)) & 15) return false;
  return true;
}
template <int BM, int BN, bool WT>
static void launch_gemm_sk_t(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * 64 * 2;
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_g3<BM, BN, true, WT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  const long mbs = (a.M + BM - 1) / BM, nbs = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_gemm_g3<BM, BN, true, WT>), dim3((unsigned)(mbs * nbs * a.kslices)), dim3(512), lds, s, a);
}
static int g_gemm_wt = 1;      // 0: ignore GemmArgs::Wt (dtk_set_option "gemm_wt")
void set_gemm_wt(int v) { g_gemm_wt = v; }
static bool use_wt(const GemmArgs& a) { return g_gemm_wt && a.Wt && !(reinterpret_cast<uintptr_t>(a.Wt) & 15); }

// Partials added in slice order (fp32), the GEMM epilogue, the row stored as bf16; then — norm_w — the HF RMSNorm of the stored row.  1024
// threads = 4096 columns of one row per pass, every slice's 16 bytes requested before the first add (S x 16 KiB in flight per block): without
// a norm the grid is (row, 4096-column chunk); with one a block owns its whole row (fp32 sum of squares: per thread over its columns in
// order, wave_sum, the 16 waves' sums in wave order) and keeps the row in registers (N <= 4096) or reads its own stores back.
template <int S>
__device__ __forceinline__ f32x4 sk_sum(const float* p, size_t stride) {
  f32x4 u[S];
#pragma unroll
  for (int k = 0; k < S; ++k) u[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)k * stride));
  f32x4 v = u[0];
#pragma unroll
  for (int k = 1; k < S; ++k) { v[0] += u[k][0]; v[1] += u[k][1]; v[2] += u[k][2]; v[3] += u[k][3]; }
  return v;
}
template <int S>
__global__ __launch_bounds__(1024) void k_sk_reduce(GemmArgs a, const bf16_t* norm_w, bf16_t* Y, int ldy, float eps) {
  __shared__ float red[16];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int flags = a.flags;
  const float* p0 = a.part + (size_t)m * a.N;
  bf16_t* crow = a.C + (size_t)m * a.ldc;
  const bool vec = (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 7) == 0;
  const bool rvec = (flags & GEMM_RESIDUAL) && (a.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a.residual) & 7) == 0;
  const bool bvec = (flags & GEMM_BIAS) && (reinterpret_cast<uintptr_t>(a.bias) & 7) == 0;
  const int nbeg = norm_w ? 0 : blockIdx.y * 4096, nend = norm_w ? a.N : min(a.N, nbeg + 4096);
  float ss = 0.f, keep[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n = nbeg + tid * 4; n < nend; n += 4096) {
    const f32x4 v = sk_sum<S>(p0 + n, (size_t)a.part_stride);
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bvec) { const u32x2 p2 = *reinterpret_cast<const u32x2*>(a.bias + n); bv[0] = pk_lo(p2[0]); bv[1] = pk_hi(p2[0]); bv[2] = pk_lo(p2[1]); bv[3] = pk_hi(p2[1]); }
    else if (flags & GEMM_BIAS) { for (int r = 0; r < 4; ++r) bv[r] = bf2f(a.bias[n + r]); }
    if (rvec) { const u32x2 p2 = *reinterpret_cast<const u32x2*>(a.residual + (size_t)m * a.ldr + n); rv[0] = pk_lo(p2[0]); rv[1] = pk_hi(p2[0]); rv[2] = pk_lo(p2[1]); rv[3] = pk_hi(p2[1]); }
    else if (flags & GEMM_RESIDUAL) { for (int r = 0; r < 4; ++r) rv[r] = bf2f(a.residual[(size_t)m * a.ldr + n + r]); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { keep[r] = bf2f(f2bf(gemm_epilogue_pre(v[r], bv[r], rv[r], flags))); ss += keep[r] * keep[r]; }
    if (vec) { const u32x2 pk = {pack2(keep[0], keep[1]), pack2(keep[2], keep[3])}; *reinterpret_cast<u32x2*>(crow + n) = pk; }
    else { for (int r = 0; r < 4; ++r) crow[n + r] = f2bf(keep[r]); }
  }
  if (!norm_w) return;
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float tot = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) tot += red[w];
  const float inv = rsqrtf(tot / (float)a.N + eps);
  bf16_t* yrow = Y + (size_t)m * ldy;
  const bool yvec = (ldy & 3) == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(norm_w)) & 7) == 0;
  for (int n = tid * 4; n < a.N; n += 4096) {
    float x[4];
    if (a.N <= 4096) { for (int r = 0; r < 4; ++r) x[r] = keep[r]; }
    else { for (int r = 0; r < 4; ++r) x[r] = bf2f(crow[n + r]); }         // the thread's own stores above
    if (yvec) {
      const u32x2 g = *reinterpret_cast<const u32x2*>(norm_w + n);
      const u32x2 pk = {pack2(pk_lo(g[0]) * rbf(x[0] * inv), pk_hi(g[0]) * rbf(x[1] * inv)), pack2(pk_lo(g[1]) * rbf(x[2] * inv), pk_hi(g[1]) * rbf(x[3] * inv))};
      *reinterpret_cast<u32x2*>(yrow + n) = pk;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) yrow[n + r] = f2bf(bf2f(norm_w[n + r]) * rbf(x[r] * inv));
    }
  }
}
// k_sk_reduce's RMSNorm on rows that are already stored (behind launch_gemm_g3_sliced): the same thread -> column map and the same order of
// the sum of squares, so Y equals what the fused reduction writes, bit for bit
__global__ __launch_bounds__(1024) void k_rmsnorm_rows_sk(const bf16_t* X, int ldx, const bf16_t* norm_w, bf16_t* Y, int ldy, int N, float eps) {
  __shared__ float red[16];
  const int m = blockIdx.x, tid = threadIdx.x;
  const bf16_t* xrow = X + (size_t)m * ldx;
  float ss = 0.f;
  for (int n = tid * 4; n < N; n += 4096) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float x = bf2f(xrow[n + r]); ss += x * x; }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float tot = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) tot += red[w];
  const float inv = rsqrtf(tot / (float)N + eps);
  bf16_t* yrow = Y + (size_t)m * ldy;
  for (int n = tid * 4; n < N; n += 4096) {
#pragma unroll
    for (int r = 0; r < 4; ++r) yrow[n + r] = f2bf(bf2f(norm_w[n + r]) * rbf(bf2f(xrow[n + r]) * inv));
  }
}
void launch_rmsnorm_rows_sk(const bf16_t* X, int ldx, const bf16_t* norm_w, bf16_t* Y, int ldy, int M, int N, float eps, hipStream_t s) {
  hipLaunchKernelGGL(k_rmsnorm_rows_sk, dim3((unsigned)M), dim3(1024), 0, s, X, ldx, norm_w, Y, ldy, N, eps);
}
// A sliced gate/up role (d = 2048 models): the reduction is SiLU*mul — per channel the slices' sums of the gate column and of the up column
// (ff further on) added in slice order, both rounded to bf16 (the Linear's outputs), then k_silu_mul's arithmetic; 1024 threads x 4 channels
template <int S>
__global__ __launch_bounds__(1024) void k_sk_reduce_swiglu(const float* part, long part_stride, int M, int ff, bf16_t* ACT, int ldact) {
  const int m = blockIdx.x, ch = blockIdx.y * 4096 + threadIdx.x * 4;
  if (ch >= ff) return;
  const float* row = part + (size_t)m * 2 * ff;
  const f32x4 g4 = sk_sum<S>(row + ch, (size_t)part_stride), u4 = sk_sum<S>(row + ff + ch, (size_t)part_stride);
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float g = rbf(g4[r]), u = rbf(u4[r]);
    const float sl = rbf(g / (1.f + expf(-g)));
    o[r] = sl * u;
  }
  const u32x2 pk = {pack2(o[0], o[1]), pack2(o[2], o[3])};
  *reinterpret_cast<u32x2*>(ACT + (size_t)m * ldact + ch) = pk;
}
void launch_sk_reduce_swiglu(const GemmArgs& a, bf16_t* ACT, int ldact, hipStream_t s) {
  const int ff = a.N >> 1;
  const dim3 grid((unsigned)a.M, (unsigned)((ff + 4095) / 4096));
#define SK_SWI(S_) case S_: hipLaunchKernelGGL(k_sk_reduce_swiglu<S_>, grid, dim3(1024), 0, s, a.part, a.part_stride, a.M, ff, ACT, ldact); break;
  switch (a.kslices) { SK_SWI(1) SK_SWI(2) SK_SWI(3) SK_SWI(4) SK_SWI(5) SK_SWI(6) SK_SWI(7) SK_SWI(8) default: break; }
#undef SK_SWI
}
void launch_sk_reduce(const GemmArgs& a, const bf16_t* norm_w, bf16_t* Y, int ldy, float eps, hipStream_t s) {
  const dim3 grid((unsigned)a.M, norm_w ? 1u : (unsigned)((a.N + 4095) / 4096));
#define SK_RED(S_) case S_: hipLaunchKernelGGL(k_sk_reduce<S_>, grid, dim3(1024), 0, s, a, norm_w, Y, ldy, eps); break;
  switch (a.kslices) { SK_RED(1) SK_RED(2) SK_RED(3) SK_RED(4) SK_RED(5) SK_RED(6) SK_RED(7) SK_RED(8) default: break; }
#undef SK_RED
}
bool launch_gemm_sk(const GemmArgs& a, const bf16_t* norm_w, bf16_t* Y, int ldy, float eps, hipStream_t s) {
  if (!launch_gemm_sk_partials(a, s)) return false;
  launch_sk_reduce(a, norm_w, Y, ldy, eps, s);
  return true;
}
bool launch_gemm_sk_partials(const GemmArgs& a, hipStream_t s) {
  if (!gemm_sk_supported(a) || !a.part || a.part_stride < (long)a.M * a.N) return false;
  if (g_sk_force_wide < 0) { const char* e = getenv("DTK_SK_TILE"); g_sk_force_wide = e ? (atoi(e) ? 1 : 0) : 2; }
  const bool wide = g_sk_force_wide == 2 ? a.M <= 128 : g_sk_force_wide == 1;
  static int probe = -1;          // DTK_G3_PROBE (timing experiments: wrong results), as in launch_gemm_g3
  if (probe < 0) { const char* e = getenv("DTK_G3_PROBE"); probe = e ? atoi(e) : 0; }
  if (g_g3_epi_direct && !probe) { GemmArgs b = a; b.flags |= GEMM_EPI_DIRECT;
    if (use_wt(b)) { if (wide) launch_gemm_sk_t<128, 256, true>(b, s); else launch_gemm_sk_t<256, 128, true>(b, s); }
    else if (wide) launch_gemm_sk_t<128, 256, false>(b, s); else launch_gemm_sk_t<256, 128, false>(b, s);
    return true; }
  if (probe) { GemmArgs b = a; if (probe & 1) b.flags |= GEMM_PROBE_NOFILL; if (probe & 2) b.flags |= GEMM_PROBE_NOMFMA;
               if (use_wt(b)) { if (wide) launch_gemm_sk_t<128, 256, true>(b, s); else launch_gemm_sk_t<256, 128, true>(b, s); }
               else if (wide) launch_gemm_sk_t<128, 256, false>(b, s); else launch_gemm_sk_t<256, 128, false>(b, s);
               return true; }
  if (use_wt(a)) { if (wide) launch_gemm_sk_t<128, 256, true>(a, s); else launch_gemm_sk_t<256, 128, true>(a, s); }
  else if (wide) launch_gemm_sk_t<128, 256, false>(a, s); else launch_gemm_sk_t<256, 128, false>(a, s);
  return true;
}

// gate/up + SiLU*mul: k_gemm_g3 with its W stage from the pair-interleaved copy
bool launch_gemm_g3_swiglu(const GemmArgs& a, hipStream_t s) {
  if (!use_wt(a) || (a.N & 31) || (a.ldc & 3) || (reinterpret_cast<uintptr_t>(a.C) & 7)) return false;
  GemmArgs b = a;
  b.flags |= GEMM_SWIGLU;
  return launch_gemm_g3(b, s);
}
// Row-major [2 ff][K] (gate rows, then up rows) -> fragment-major tiles (k_retile's image) in pair order: tile 2 q = gate rows 16 q .. 16 q + 15,
// tile 2 q + 1 = up rows ff + 16 q ..; one thread per 16-byte lane slot
__global__ void k_retile_pairs(const bf16_t* src, bf16_t* dst, int ff, int K) {
  const int K32 = (K + 31) >> 5, N16 = (2 * ff + 15) >> 4;
  const long total = (long)N16 * K32 * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long tile = i >> 6;
    const int tk = (int)(tile % K32), tn = (int)(tile / K32);
    const int n = ((tn & 1) ? ff : 0) + (tn >> 1) * 16 + (lane & 15), k = tk * 32 + (lane >> 4) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((tn >> 1) * 16 + (lane & 15) < ff && k < K) v = *reinterpret_cast<const u32x4*>(src + (size_t)n * K + k);
    reinterpret_cast<u32x4*>(dst)[i] = v;
  }
}
void launch_retile_pairs(const bf16_t* src, bf16_t* dst, int ff, int K, hipStream_t s) {
  const long total = (long)((2 * ff + 15) >> 4) * ((K + 31) >> 5) * 64;
  long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_retile_pairs, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, ff, K);
}

// a sliced-K role in ONE launch (k_gemm_g3<.., SL>): for M large enough that tiles alone fill the chip; bit-identical to launch_gemm_sk
template <int BM, int BN, bool WT>
static void launch_gemm_g3_sl_t(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * 64 * 2;
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_g3<BM, BN, false, WT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  const long mbs = (a.M + BM - 1) / BM, nbs = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_gemm_g3<BM, BN, false, WT, true>), dim3((unsigned)(8 * ((nbs + 7) / 8) * mbs)), dim3(512), lds, s, a);
}
bool launch_gemm_g3_sliced(const GemmArgs& a, hipStream_t s) {
  if (!gemm_sk_supported(a) || a.kslices < 2) return false;
  auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  const bool wide = blocks(128, 256) < blocks(256, 128);
  GemmArgs b = a;
  if (g_g3_epi_direct) b.flags |= GEMM_EPI_DIRECT;
  if (use_wt(b)) { if (wide) launch_gemm_g3_sl_t<128, 256, true>(b, s); else launch_gemm_g3_sl_t<256, 128, true>(b, s); }
  else if (wide) launch_gemm_g3_sl_t<128, 256, false>(b, s); else launch_gemm_g3_sl_t<256, 128, false>(b, s);
  return true;
}

static int g_gemm_impl = -1;    // 0 = k_gemm_mfma (register-staged), 1 = k_gemm_dma (LDS-DMA ring), 2 = k_gemm_glds for shapes with >= g_glds_min_tiles 128 x 128 tiles (else k_gemm_mfma), 3 = auto (default: 2 for M >= 1024, else 0); dtk_set_option "gemm_impl" / DTK_GEMM_IMPL
void set_gemm_impl(int v) { g_gemm_impl = v; }
static int g_gemm_ring = 3;
void set_gemm_ring(int v) { g_gemm_ring = v; }

static int g_gemm_stages = -1;  // register stages of the 64x64 kernel: 1..4 (dtk_set_option "gemm_stages" / DTK_GEMM_STAGES), default 3
void set_gemm_stages(int v) { g_gemm_stages = v; }
static int g_gemm_bk = 64;      // k-tile of the 64x64 kernel: 64 | 128 (dtk_set_option "gemm_bk")
void set_gemm_bk(int v) { g_gemm_bk = v; }
static int g_gemm_tile = -1;   // 0 auto, 1 = 64x64, 2 = 128x64, 3 = 128x128, 4 = 64x32, 5 = 32x32 (dtk_set_option "gemm_tile" / DTK_GEMM_TILE)
void set_gemm_tile(int v) { g_gemm_tile = v; }
static int gemm_tile_override() {
  if (g_gemm_tile < 0) {
    const char* e = getenv("DTK_GEMM_TILE");
    g_gemm_tile = !e ? 0 : (!strcmp(e, "64x64") ? 1 : (!strcmp(e, "128x64") ? 2 : (!strcmp(e, "128x128") ? 3 : (!strcmp(e, "64x32") ? 4 : (!strcmp(e, "32x32") ? 5 : 0)))));
  }
  return g_gemm_tile;
}
void launch_gemm_mfma(const GemmArgs& a, hipStream_t s) {
  auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  int tile = gemm_tile_override();
  if (!tile) {
    // measured (ds-7b, M = 243 / 729): 64x64 17.1 / 4.6 ms, 128x64 18.7 / 5.5, 128x128 22.6 / 7.1 — these GEMMs are latency- and
    // tile-count-bound, not LDS-read-bound.  Shapes that give fewer than two 64x64 blocks per CU (N = d in the prefill,
    // N = 1152 in the ViT) take the smaller tiles: more blocks, more waves in flight
    tile = 1;
    if (blocks(64, 64) < 512) tile = blocks(64, 32) >= 512 ? 4 : 5;
  }
  auto grid = [&](int bm, int bn) { return dim3((unsigned)(8 * ((((a.N + bn - 1) / bn) + 7) / 8) * ((a.M + bm - 1) / bm))); };
  if (g_gemm_impl < 0) { const char* e = getenv("DTK_GEMM_IMPL"); g_gemm_impl = e ? atoi(e) : 3; }
  // 3 = auto (default): k_gemm_glds for M >= 1024 (the batched ViT: same wall time as k_gemm_mfma, 319 instead of 711 MB fetched from
  // the memory side per launch — profiles/r03_pmc_mfma.csv — which is what a reward pass that runs BESIDE the HBM-bound decode
  // steps should cost them), k_gemm_mfma below (prefill M = 243, one image M = 729: measured 14.4 vs 14.8 ms and 4.0 vs 4.6 ms)
  // 3 = auto, 4 (kept as a name for the same choice): k_gemm_g3 wherever the shape gives it a block per two CUs — the batched ViT
  // (8 images: 2.25 -> 1.91 ms per image, profiles/r04_bench_vit_epilogue.txt) and the prefill's gate/up GEMM (14.4 -> 13.6 ms for
  // ViT + projector + prefill) — then k_gemm_glds for M >= 1024, then k_gemm_mfma; all bit-identical
  if ((g_gemm_impl == 3 || g_gemm_impl == 4) && !gemm_tile_override() && launch_gemm_g3(a, s)) return;
  if ((g_gemm_impl == 2 || ((g_gemm_impl == 3 || g_gemm_impl == 4) && a.M >= 1024)) && !gemm_tile_override() && launch_gemm_glds(a, s)) return;
  if (g_gemm_stages < 0) { const char* e = getenv("DTK_GEMM_STAGES"); g_gemm_stages = e ? atoi(e) : 3; }
  const int D = g_gemm_stages;
#define GEMM_LAUNCH(BM_, BN_)                                                                                         \
  switch (D) {                                                                                                        \
    case 1: hipLaunchKernelGGL((k_gemm_mfma<BM_, BN_, 1>), grid(BM_, BN_), dim3(256), 0, s, a); break;                \
    case 2: hipLaunchKernelGGL((k_gemm_mfma<BM_, BN_, 2>), grid(BM_, BN_), dim3(256), 0, s, a); break;                \
    case 4: hipLaunchKernelGGL((k_gemm_mfma<BM_, BN_, 4>), grid(BM_, BN_), dim3(256), 0, s, a); break;                \
    default: hipLaunchKernelGGL((k_gemm_mfma<BM_, BN_, 3>), grid(BM_, BN_), dim3(256), 0, s, a); break;               \
  }
  if (tile == 3) { GEMM_LAUNCH(128, 128) }
  else if (tile == 2) { GEMM_LAUNCH(128, 64) }
  else if (tile == 4) { GEMM_LAUNCH(64, 32) }
  else if (tile == 5) { GEMM_LAUNCH(32, 32) }
  else if (g_gemm_bk == 128) {
    switch (D) {
      case 1: hipLaunchKernelGGL((k_gemm_mfma<64, 64, 1, 128>), grid(64, 64), dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL((k_gemm_mfma<64, 64, 2, 128>), grid(64, 64), dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL((k_gemm_mfma<64, 64, 3, 128>), grid(64, 64), dim3(256), 0, s, a); break;
    }
  }
  else { GEMM_LAUNCH(64, 64) }
#undef GEMM_LAUNCH
}

// Plain one-thread-per-output GEMM: the obviously-correct twin of k_gemm_mfma (selected
// by DTK_GEMM=naive and by the op-level parity test) — same epilogue, same rounding.
__global__ void k_gemm_naive(GemmArgs a) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= a.M || n >= a.N) return;
  const bf16_t* ar = a.A + (size_t)m * a.lda;
  const bf16_t* wr = a.W + (size_t)n * a.ldw;
  float acc = 0.f;
  if (a.kslices > 1) {                  // a sliced-K role: a chain per slice of k-tiles, the slices' sums added in order
    const int q = sk_tiles_per_slice(a.K, a.kslices) * 64;
    for (int ks = 0; ks < a.kslices; ++ks) {
      const int k0 = min(ks * q, (a.K / 64) * 64), k1 = ks == a.kslices - 1 ? a.K : min(k0 + q, (a.K / 64) * 64);
      float sl = 0.f;
      for (int k = k0; k < k1; ++k) sl = fmaf(bf2f(ar[k]), bf2f(wr[k]), sl);
      acc = ks ? acc + sl : sl;
    }
  } else
  for (int k = 0; k < a.K; ++k) acc = fmaf(bf2f(ar[k]), bf2f(wr[k]), acc);
  a.C[(size_t)m * a.ldc + n] = f2bf(gemm_epilogue(acc, m, n, a));
}

void launch_gemm_naive(const GemmArgs& a, hipStream_t s) {
  dim3 grid((a.N + 63) / 64, (a.M + 3) / 4);
  hipLaunchKernelGGL(k_gemm_naive, grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// Row kernels: one wave per row, 16-byte loads, fp32 statistics, one bf16 rounding of the
// result (torch layer_norm) / two (HF LlamaRMSNorm: normalise -> bf16 -> * weight -> bf16).
__global__ __launch_bounds__(256) void k_layernorm_rows(const bf16_t* X, int ldx, const bf16_t* w,
                                                        const bf16_t* b, bf16_t* Y, int ldy,
                                                        int M, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int D8 = D >> 3;
  const u32x4* x4 = reinterpret_cast<const u32x4*>(X + (size_t)row * ldx);
  float s = 0.f;
  for (int c = lane; c < D8; c += 64) {
    const u32x4 v = x4[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) s += pk_lo(v[e]) + pk_hi(v[e]);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane; c < D8; c += 64) {
    const u32x4 v = x4[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = pk_lo(v[e]) - mean, d1 = pk_hi(v[e]) - mean;
      q += d0 * d0 + d1 * d1;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  const u32x4* w4 = reinterpret_cast<const u32x4*>(w);
  const u32x4* b4 = reinterpret_cast<const u32x4*>(b);
  u32x4* y4 = reinterpret_cast<u32x4*>(Y + (size_t)row * ldy);
  for (int c = lane; c < D8; c += 64) {
    const u32x4 v = x4[c], g = w4[c], be = b4[c];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2((pk_lo(v[e]) - mean) * rstd * pk_lo(g[e]) + pk_lo(be[e]),
                   (pk_hi(v[e]) - mean) * rstd * pk_hi(g[e]) + pk_hi(be[e]));
    y4[c] = o;
  }
}
void launch_layernorm_rows(const bf16_t* X, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* Y,
                           int ldy, int M, int D, float eps, hipStream_t s) {
  hipLaunchKernelGGL(k_layernorm_rows, dim3((M + 3) / 4), dim3(256), 0, s, X, ldx, w, b, Y, ldy,
                     M, D, eps);
}

__global__ __launch_bounds__(256) void k_rmsnorm_rows(const bf16_t* X, int ldx, const bf16_t* w,
                                                      bf16_t* Y, int ldy, int M, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int D8 = D >> 3;
  const u32x4* x4 = reinterpret_cast<const u32x4*>(X + (size_t)row * ldx);
  float ss = 0.f;
  for (int c = lane; c < D8; c += 64) {
    const u32x4 v = x4[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = pk_lo(v[e]), hi = pk_hi(v[e]);
      ss += lo * lo;
      ss += hi * hi;
    }
  }
  const float inv = rsqrtf(wave_sum(ss) / (float)D + eps);
  const u32x4* w4 = reinterpret_cast<const u32x4*>(w);
  u32x4* y4 = reinterpret_cast<u32x4*>(Y + (size_t)row * ldy);
  for (int c = lane; c < D8; c += 64) {
    const u32x4 v = x4[c], g = w4[c];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2(pk_lo(g[e]) * rbf(pk_lo(v[e]) * inv), pk_hi(g[e]) * rbf(pk_hi(v[e]) * inv));
    y4[c] = o;
  }
}
void launch_rmsnorm_rows(const bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int ldy, int M,
                         int D, float eps, hipStream_t s) {
  hipLaunchKernelGGL(k_rmsnorm_rows, dim3((M + 3) / 4), dim3(256), 0, s, X, ldx, w, Y, ldy, M, D,
                     eps);
}

// act[m][i] = bf16(bf16(silu(gate)) * up), GU row = [gate(ff) | up(ff)]
__global__ void k_silu_mul(const bf16_t* GU, int ff, bf16_t* ACT, int M) {
  const int F8 = ff >> 3;
  const long total = (long)M * F8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / F8), c = (int)(idx - (long)m * F8);
    const u32x4 g = reinterpret_cast<const u32x4*>(GU + (size_t)m * 2 * ff)[c];
    const u32x4 u = reinterpret_cast<const u32x4*>(GU + (size_t)m * 2 * ff + ff)[c];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = pk_lo(g[e]), g1 = pk_hi(g[e]);
      const float s0 = rbf(g0 / (1.f + expf(-g0))), s1 = rbf(g1 / (1.f + expf(-g1)));
      o[e] = pack2(s0 * pk_lo(u[e]), s1 * pk_hi(u[e]));
    }
    reinterpret_cast<u32x4*>(ACT + (size_t)m * ff)[c] = o;
  }
}
void launch_silu_mul(const bf16_t* GU, int ff, bf16_t* ACT, int M, hipStream_t s) {
  long total = (long)M * (ff >> 3);
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_silu_mul, dim3(grid), dim3(256), 0, s, GU, ff, ACT, M);
}

__global__ void k_embed_gather(const int32_t* ids, const bf16_t* embed, bf16_t* X, int T, int d) {
  const int t = blockIdx.x;
  const u32x4* src = reinterpret_cast<const u32x4*>(embed + (size_t)ids[t] * d);
  u32x4* dst = reinterpret_cast<u32x4*>(X + (size_t)t * d);
  for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) dst[c] = src[c];
}
void launch_embed_gather(const int32_t* ids, const bf16_t* embed, bf16_t* X, int T, int d,
                         hipStream_t s) {
  hipLaunchKernelGGL(k_embed_gather, dim3(T), dim3(256), 0, s, ids, embed, X, T, d);
}

__global__ void k_copy_rows(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int M, int D) {
  const int m = blockIdx.x;
  const u32x4* s4 = reinterpret_cast<const u32x4*>(src + (size_t)m * lds_);
  u32x4* d4 = reinterpret_cast<u32x4*>(dst + (size_t)m * ldd);
  for (int c = threadIdx.x; c < (D >> 3); c += blockDim.x) d4[c] = s4[c];
}
void launch_copy_rows(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int M, int D,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_copy_rows, dim3(M), dim3(256), 0, s, src, lds_, dst, ldd, M, D);
}

// pixels fp32 [3][S][S] -> patches bf16 [(S/p)^2][ldp], column = c*p*p + kh*p + kw (the
// flattening of the conv weight [D][3][p][p]); columns >= 3*p*p are zero padding.
__global__ void k_im2col(const float* pixels, bf16_t* patches, int image, int patch, int ldp) {
  const int np = image / patch;
  const int p = blockIdx.x;  // patch index, row-major (py, px)
  const int py = p / np, px = p - py * np;
  const int kk = 3 * patch * patch;
  for (int col = threadIdx.x; col < ldp; col += blockDim.x) {
    float v = 0.f;
    if (col < kk) {
      const int c = col / (patch * patch);
      const int r = col - c * patch * patch;
      const int kh = r / patch, kw = r - kh * patch;
      v = pixels[((size_t)c * image + (py * patch + kh)) * image + (px * patch + kw)];
    }
    patches[(size_t)p * ldp + col] = f2bf(v);
  }
}
void launch_im2col(const float* pixels, bf16_t* patches, int image, int patch, int ldp,
                   hipStream_t s) {
  const int np = image / patch;
  hipLaunchKernelGGL(k_im2col, dim3(np * np), dim3(256), 0, s, pixels, patches, image, patch, ldp);
}

// grid (T, H), 64 threads: thread i owns the RoPE pair (i, i+64) of q head h; blocks h < KVH also rotate k head h
// and copy v head h.  QKV row layout: [H q heads | KVH k heads | KVH v heads] x 128.
__global__ void k_rope_scatter(const bf16_t* QKV, bf16_t* Qh, bf16_t* kcache, bf16_t* vcache,
                               const bf16_t* cos_t, const bf16_t* sin_t, int T, int start_pos,
                               int H, int KVH, int T_max) {
  const int t = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
  const int qd = H * 128, kvd = KVH * 128;
  const int pos = start_pos + t;
  const bf16_t* row = QKV + (size_t)t * (qd + 2 * kvd);
  const float c = bf2f(cos_t[(size_t)pos * 64 + i]);
  const float s = bf2f(sin_t[(size_t)pos * 64 + i]);
  {
    const float x1 = bf2f(row[h * 128 + i]), x2 = bf2f(row[h * 128 + i + 64]);
    bf16_t* dst = Qh + ((size_t)h * T + t) * 128;
    dst[i] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
    dst[i + 64] = f2bf(rbf(x2 * c) + rbf(x1 * s));
  }
  if (h >= KVH) return;
  {
    const float x1 = bf2f(row[qd + h * 128 + i]), x2 = bf2f(row[qd + h * 128 + i + 64]);
    bf16_t* dst = kcache + ((size_t)h * T_max + pos) * 128;
    dst[i] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
    dst[i + 64] = f2bf(rbf(x2 * c) + rbf(x1 * s));
  }
  {
    bf16_t* dst = vcache + ((size_t)h * T_max + pos) * 128;
    dst[i] = row[qd + kvd + h * 128 + i];
    dst[i + 64] = row[qd + kvd + h * 128 + i + 64];
  }
}
void launch_rope_scatter(const bf16_t* QKV, bf16_t* Qh, bf16_t* kcache, bf16_t* vcache,
                         const bf16_t* cos_t, const bf16_t* sin_t, int T, int start_pos, int H, int KVH,
                         int T_max, hipStream_t s) {
  hipLaunchKernelGGL(k_rope_scatter, dim3(T, H), dim3(64), 0, s, QKV, Qh, kcache, vcache, cos_t,
                     sin_t, T, start_pos, H, KVH, T_max);
}

// The q/k/v role as a sliced-K GEMM: its reduction IS k_rope_scatter's input — the slices' fp32 sums added in order, rounded to bf16 (the
// Linear's output: what k_sk_reduce would have stored), then k_rope_scatter's arithmetic; no [T][qkvn] buffer in between.
template <int S>
__global__ void k_sk_rope_scatter(const float* part, long part_stride, bf16_t* Qh, bf16_t* kcache, bf16_t* vcache,
                                  const bf16_t* cos_t, const bf16_t* sin_t, int T, int start_pos, int H, int KVH, int T_max) {
  const int t = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
  const int qd = H * 128, kvd = KVH * 128, N = qd + 2 * kvd;
  const int pos = start_pos + t;
  const float* row = part + (size_t)t * N;
  auto val = [&](int n) {
    float v = row[n];
#pragma unroll
    for (int k = 1; k < S; ++k) v += row[(size_t)k * (size_t)part_stride + n];
    return rbf(v);
  };
  const float c = bf2f(cos_t[(size_t)pos * 64 + i]);
  const float s = bf2f(sin_t[(size_t)pos * 64 + i]);
  {
    const float x1 = val(h * 128 + i), x2 = val(h * 128 + i + 64);
    bf16_t* dst = Qh + ((size_t)h * T + t) * 128;
    dst[i] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
    dst[i + 64] = f2bf(rbf(x2 * c) + rbf(x1 * s));
  }
  if (h >= KVH) return;
  {
    const float x1 = val(qd + h * 128 + i), x2 = val(qd + h * 128 + i + 64);
    bf16_t* dst = kcache + ((size_t)h * T_max + pos) * 128;
    dst[i] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
    dst[i + 64] = f2bf(rbf(x2 * c) + rbf(x1 * s));
  }
  {
    bf16_t* dst = vcache + ((size_t)h * T_max + pos) * 128;
    dst[i] = f2bf(val(qd + kvd + h * 128 + i));
    dst[i + 64] = f2bf(val(qd + kvd + h * 128 + i + 64));
  }
}
void launch_sk_rope_scatter(const float* part, long part_stride, int kslices, bf16_t* Qh, bf16_t* kcache, bf16_t* vcache,
                            const bf16_t* cos_t, const bf16_t* sin_t, int T, int start_pos, int H, int KVH, int T_max, hipStream_t s) {
#define SK_ROPE(S_) case S_: hipLaunchKernelGGL(k_sk_rope_scatter<S_>, dim3(T, H), dim3(64), 0, s, part, part_stride, Qh, kcache, vcache, cos_t, sin_t, T, start_pos, H, KVH, T_max); break;
  switch (kslices) { SK_ROPE(1) SK_ROPE(2) SK_ROPE(3) SK_ROPE(4) SK_ROPE(5) SK_ROPE(6) SK_ROPE(7) SK_ROPE(8) default: break; }
#undef SK_ROPE
}

// ------------------------------------------------------------------------------------------
// Attention for the batched paths (ViT: N = 729, hd = 72, full; prefill: hd = 128, causal).
// One wave per query row, 4 queries per block sharing 64-key K/V tiles staged in LDS.
// Scores: lanes over keys (lane l owns key l of the tile, dot2 over the packed q kept in
// VGPRs); P.V: lanes over dims (p_l broadcast with v_readlane).  fp32 online softmax,
// one bf16 rounding of the output (fused-attention semantics, as timm's / HF's SDPA).
template <int HD>
__global__ __launch_bounds__(256) void k_attention(AttnArgs a) {
  constexpr int HD2 = HD / 2;    // dwords per row
  constexpr int RS = HD2 + 1;    // padded row stride in dwords (odd -> conflict-free columns)
  __shared__ uint32_t Ks[64 * RS];
  __shared__ uint32_t Vs[64 * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y;
  { const long bz = blockIdx.z; a.Q += bz * a.q_sb; a.K += bz * a.k_sb; a.V += bz * a.v_sb; a.O += bz * a.o_sb; }   // batched problems
  const int hk = a.kv_group > 1 ? h / a.kv_group : h;   // GQA
  const int qi = blockIdx.x * 4 + wave;
  const bool qok = qi < a.Tq;
  const int qrow = qok ? qi : a.Tq - 1;
  const int qpos = a.q_offset + qrow;

  // q packed bf16 pairs, replicated in every lane
  uint32_t q2[HD2];
  {
    const uint32_t* qp = reinterpret_cast<const uint32_t*>(a.Q + (size_t)h * a.q_sh + (size_t)qrow * a.q_st);
#pragma unroll
    for (int e = 0; e < HD2; ++e) q2[e] = qp[e];
  }
  float m = -1e30f, l = 0.f;
  float o0 = 0.f, o1 = 0.f;  // dims 2*lane, 2*lane+1 (lanes < HD2)

  // keys needed by this block: causal -> up to the last query's position
  int kmax = a.Tk;
  if (a.causal) {
    const int last_q = min(a.Tq - 1, blockIdx.x * 4 + 3);
    kmax = min(a.Tk, a.q_offset + last_q + 1);
  }
  const int klim = a.causal ? min(a.Tk, qpos + 1) : a.Tk;  // keys visible to this query

  for (int j0 = 0; j0 < kmax; j0 += 64) {
    __syncthreads();
    // cooperative tile load: 64 rows x HD2 dwords, coalesced along the row
    for (int idx = tid; idx < 64 * HD2; idx += 256) {
      const int r = idx / HD2, c = idx - r * HD2;
      int j = j0 + r; if (j >= a.Tk) j = a.Tk - 1;
      Ks[r * RS + c] = reinterpret_cast<const uint32_t*>(a.K + (size_t)hk * a.k_sh + (size_t)j * a.k_st)[c];
      Vs[r * RS + c] = reinterpret_cast<const uint32_t*>(a.V + (size_t)hk * a.v_sh + (size_t)j * a.v_st)[c];
    }
    __syncthreads();
    // scores: lane = key
    float sc = 0.f;
#pragma unroll
    for (int e = 0; e < HD2; ++e) sc = dot2(q2[e], Ks[lane * RS + e], sc);
    sc *= a.scale;
    const bool vis = (j0 + lane) < klim;
    if (!vis) sc = -1e30f;
    const float tmax = wave_max(sc);
    const float mn = fmaxf(m, tmax);
    const float corr = __expf(m - mn);
    const float p = vis ? __expf(sc - mn) : 0.f;
    l = l * corr + wave_sum(p);
    o0 *= corr; o1 *= corr;
    m = mn;
    // P.V: lane = dim pair
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) {
      const float pk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), kk));
      const uint32_t vv = (lane < HD2) ? Vs[kk * RS + lane] : 0u;
      o0 = fmaf(pk, pk_lo(vv), o0);
      o1 = fmaf(pk, pk_hi(vv), o1);
    }
  }
  if (qok && lane < HD2) {
    const float inv = 1.f / l;
    uint32_t* op = reinterpret_cast<uint32_t*>(a.O + (size_t)h * a.o_sh + (size_t)qi * a.o_st);
    op[lane] = pack2(o0 * inv, o1 * inv);
  }
}

// MFMA flash attention for the ViT (hd 72, 729 x 729, bidirectional) and the decoder prefill (hd 128, causal).
// One block = 4 waves = 64 queries, a wave owns 16 queries; key tiles of 64 staged through LDS (register
// prefetch of the next tile under the MFMAs).  Everything is computed TRANSPOSED so that a lane's MFMA
// column is always "its" query (lane & 15):
//   S^T[key][q] = K . Q^T   A = K rows from LDS (16-byte reads), B = Q rows (registers, loaded once)
//   O^T[d][q]   = V^T . P^T A = V^T gathered from the row-major LDS tile (8 ds_read_u16 per fragment),
//                           B = P^T = the lane's own probabilities: the k index of the MFMA is a free
//                           permutation, so k = (lane group g, element e) is mapped to key g*4+e of key
//                           sub-tile 2s (e < 4) / 2s+1 (e >= 4) — exactly the keys whose scores the lane holds
//                           in its S^T accumulators.  P never goes through LDS, V is never transposed.
// Row max / sum: 16 in-lane values + two xor-shuffles (lane groups 16/32 apart).  fp32 scores, fp32 running
// max/sum, probabilities fed to the P.V MFMA as a bf16 hi + lo pair (~fp32-probability semantics, the same
// as the VALU kernel and the oracle), fp32 O, one bf16 rounding of the output.
template <int HD>
__global__ __launch_bounds__(256) void k_attention_mfma(AttnArgs a) {
  constexpr int CH = HD / 8;                           // 16-byte chunks per row
  constexpr int KS = (HD + 31) / 32;                   // k-steps over d for S^T
  constexpr int DT = (HD + 15) / 16;                   // 16-row tiles of O^T
  constexpr int HDP = (HD % 64 == 0) ? HD + 8 : HD;    // row stride (elements): 72 -> 36 dwords, 128 -> 68 dwords
  constexpr int NLD = (64 * CH + 255) / 256;           // 16-byte chunks a thread stages per operand
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * HDP + 32];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * HDP + 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int h = blockIdx.y;
  { const long bz = blockIdx.z; a.Q += bz * a.q_sb; a.K += bz * a.k_sb; a.V += bz * a.v_sb; a.O += bz * a.o_sb; }   // batched problems
  const int hk = a.kv_group > 1 ? h / a.kv_group : h;   // GQA
  const int myq = blockIdx.x * 64 + wave * 16 + lq;
  const int qrow = myq < a.Tq ? myq : a.Tq - 1;

  bf16x8_t qf[KS];
  {
    const bf16_t* qp = a.Q + (size_t)h * a.q_sh + (size_t)qrow * a.q_st;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 4 + g;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < CH) v = *reinterpret_cast<const u32x4*>(qp + c * 8);
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  const int klim = a.causal ? min(a.Tk, a.q_offset + qrow + 1) : a.Tk;   // keys visible to this lane's query
  int kmax = a.Tk;                                                        // keys this block needs
  if (a.causal) kmax = min(a.Tk, a.q_offset + min(a.Tq - 1, (int)blockIdx.x * 64 + 63) + 1);

  float m = -1e30f, l = 0.f;
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 rk[NLD], rv[NLD];
  auto tile_load = [&](int j0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 256;
      if (idx < 64 * CH) {
        const int r = idx / CH, c = idx - r * CH;
        int j = j0 + r; if (j >= a.Tk) j = a.Tk - 1;
        rk[i] = *reinterpret_cast<const u32x4*>(a.K + (size_t)hk * a.k_sh + (size_t)j * a.k_st + c * 8);
        rv[i] = *reinterpret_cast<const u32x4*>(a.V + (size_t)hk * a.v_sh + (size_t)j * a.v_st + c * 8);
      }
    }
  };
  auto tile_write = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 256;
      if (idx < 64 * CH) {
        const int r = idx / CH, c = idx - r * CH;
        *reinterpret_cast<u32x4*>(&Ks[r * HDP + c * 8]) = rk[i];
        *reinterpret_cast<u32x4*>(&Vs[r * HDP + c * 8]) = rv[i];
      }
    }
  };

  tile_load(0);
  for (int j0 = 0; j0 < kmax; j0 += 64) {
    __syncthreads();   // the previous tile's fragment reads are done
    tile_write();
    __syncthreads();
    if (j0 + 64 < kmax) tile_load(j0 + 64);

    // ---- S^T = K . Q^T : s[t][r] = score(key j0 + t*16 + g*4 + r, query lq)
    f32x4 sc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 4 + g;
        u32x4 kv = *reinterpret_cast<const u32x4*>(&Ks[(t * 16 + lq) * HDP + c * 8]);
        if ((HD % 32) != 0 && ks == KS - 1 && c >= CH) kv = (u32x4){0u, 0u, 0u, 0u};
        sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kv), qf[ks], sc[t], 0, 0, 0);
      }
    }
    float tmax = -1e30f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + t * 16 + g * 4 + r;
        const float v = key < klim ? sc[t][r] * a.scale : -1e30f;
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
        const float p = key < klim ? __expf(sc[t][r] - mn) : 0.f;
        sc[t][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l = l * corr + psum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] *= corr;

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // p = hi + lo with hi = bf16(p), lo = bf16(p - hi): two MFMAs on the same V^T fragment keep ~16
      // mantissa bits of the fp32 probabilities (a single bf16 P costs ~2e-3 relative L2 on the output)
      u32x4 pw, pl;
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const float p0 = sc[2 * s2 + half][2 * pr], p1 = sc[2 * s2 + half][2 * pr + 1];
          const uint32_t hi = pack2(p0, p1);
          pw[half * 2 + pr] = hi;
          pl[half * 2 + pr] = pack2(p0 - pk_lo(hi), p1 - pk_hi(hi));
        }
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
      const bf16x8_t pfl = __builtin_bit_cast(bf16x8_t, pl);
      const bf16_t* v0 = Vs + ((2 * s2) * 16 + g * 4) * HDP + lq;       // keys of sub-tile 2*s2
      const bf16_t* v1 = v0 + 16 * HDP;                                  // keys of sub-tile 2*s2 + 1
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        u32x4 vw;
        vw[0] = (uint32_t)v0[dt * 16] | ((uint32_t)v0[dt * 16 + HDP] << 16);
        vw[1] = (uint32_t)v0[dt * 16 + 2 * HDP] | ((uint32_t)v0[dt * 16 + 3 * HDP] << 16);
        vw[2] = (uint32_t)v1[dt * 16] | ((uint32_t)v1[dt * 16 + HDP] << 16);
        vw[3] = (uint32_t)v1[dt * 16 + 2 * HDP] | ((uint32_t)v1[dt * 16 + 3 * HDP] << 16);
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vw);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfl, o[dt], 0, 0, 0);
      }
    }
  }
  if (myq < a.Tq) {
    const float inv = 1.f / l;
    bf16_t* op = a.O + (size_t)h * a.o_sh + (size_t)myq * a.o_st;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d < HD) {   // HD % 4 == 0: a group of 4 is fully in or fully out
        u32x2 w;
        w[0] = pack2(o[dt][0] * inv, o[dt][1] * inv);
        w[1] = pack2(o[dt][2] * inv, o[dt][3] * inv);
        *reinterpret_cast<u32x2*>(op + d) = w;
      }
    }
  }
}

void launch_attention(const AttnArgs& a, hipStream_t s) {
  const bool can_mfma = (a.hd == 72 || a.hd == 128) && (a.q_st % 8 == 0) && (a.k_st % 8 == 0) && (a.v_st % 8 == 0) &&
                        (a.o_st % 4 == 0) && (a.q_sh % 8 == 0) && (a.k_sh % 8 == 0) && (a.v_sh % 8 == 0) && (a.o_sh % 4 == 0);
  // the choice must not depend on Tq: a tail prefill after a KV-prefix reuse (1 query) and the full prefill (all queries) have
  // to round identically, and they do when the same kernel scans the same 64-key tiles (a query's result does not depend on
  // which block / lane holds it)
  const bool mfma = can_mfma && a.impl != 1;
  if (mfma) {
    dim3 grid((a.Tq + 63) / 64, a.H, a.nbatch > 0 ? a.nbatch : 1);
    if (a.hd == 72) hipLaunchKernelGGL((k_attention_mfma<72>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_attention_mfma<128>), grid, dim3(256), 0, s, a);
    return;
  }
  dim3 grid((a.Tq + 3) / 4, a.H, a.nbatch > 0 ? a.nbatch : 1);
  if (a.hd == 72) hipLaunchKernelGGL((k_attention<72>), grid, dim3(256), 0, s, a);
  else if (a.hd == 128) hipLaunchKernelGGL((k_attention<128>), grid, dim3(256), 0, s, a);
  else if (a.hd == 64) hipLaunchKernelGGL((k_attention<64>), grid, dim3(256), 0, s, a);
  else if (a.hd == 32) hipLaunchKernelGGL((k_attention<32>), grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// Deterministic synthetic weights (no checkpoints exist offline): value(i) = offset +
// scale * u(i), u in [-sqrt(3), sqrt(3)) from a 32-bit integer hash of (seed, tag, i),
// rounded to bf16.  Bit-identical to oracle/synth.py (integer hash, one fp32 multiply-add).
__host__ __device__ __forceinline__ uint32_t synth_hash(uint32_t seed_lo, uint32_t seed_hi,
                                                        uint32_t tag, uint32_t i) {
  uint32_t x = i * 0x9E3779B1u + tag * 0x85EBCA77u + seed_lo;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  x += seed_hi * 0xC2B2AE3Du;
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  return x;
}
__global__ void k_fill_synth(bf16_t* dst, int64_t n, uint32_t seed_lo, uint32_t seed_hi,
                             uint32_t tag, float scale, float offset) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t hsh = synth_hash(seed_lo, seed_hi, tag, (uint32_t)i);
    // top 24 bits -> [-1, 1) exactly representable in fp32
    const float u = (float)((int32_t)(hsh >> 8) - 8388608) * (1.0f / 8388608.0f);
    dst[i] = f2bf(__fmaf_rn(u, scale * 1.7320508f, offset));
  }
}
void launch_fill_synth(bf16_t* dst, int64_t n, uint64_t seed, uint32_t tag, float scale,
                       float offset, hipStream_t s) {
  int64_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_fill_synth, dim3((unsigned)blocks), dim3(256), 0, s, dst, n,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), tag, scale, offset);
}

__global__ void k_f32_to_bf16(const float* src, bf16_t* dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f2bf(src[i]);
}
void launch_f32_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
}

// ------------------------------------------------------------------------------------------
// fp8 (OCP e4m3fn) weight quantisation, one block per row: scale = 2^ceil(log2(max|w| / 448)) (a power
// of two, so q * scale is exactly representable in bf16), q = rne_e4m3(w / scale).  The bf16 master row is
// overwritten with the de-quantised values: prefill GEMMs, the fragment-major copy, dtk_read_tensor and the
// CPU oracle all see the same "effective" weights the fp8 decode kernels compute with.
__global__ __launch_bounds__(256) void k_quant_fp8_rows(bf16_t* W, uint8_t* W8, float* scale, int N, int K) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int row = blockIdx.x;
  bf16_t* w = W + (size_t)row * K;
  uint8_t* q = W8 + (size_t)row * K;
  __shared__ float red[4];
  float amax = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(bf2f(w[k])));
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc = 1.f;
  if (amax > 0.f) sc = exp2f(ceilf(log2f(amax / 448.f)));
  if (amax > 448.f * sc) sc *= 2.f;   // guard the log2/ceil rounding
  if (threadIdx.x == 0) scale[row] = sc;
  const float inv = 1.f / sc;
  for (int k = threadIdx.x * 2; k < K; k += 512) {   // K is even (K % 16 == 0)
    const float a0 = bf2f(w[k]) * inv, a1 = bf2f(w[k + 1]) * inv;
    const int p = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
    q[k] = (uint8_t)(p & 0xff);
    q[k + 1] = (uint8_t)((p >> 8) & 0xff);
    const f32x2 d = __builtin_amdgcn_cvt_pk_f32_fp8(p, false);
    w[k] = f2bf(d[0] * sc);
    w[k + 1] = f2bf(d[1] * sc);
  }
}
void launch_quant_fp8_rows(bf16_t* W, uint8_t* W8, float* scale, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(k_quant_fp8_rows, dim3(N), dim3(256), 0, s, W, W8, scale, N, K);
}
