// kernels_batch_gemm.hip — the batched decode GEMVs with the slots' input vectors streamed through LDS by the LDS-DMA path.
//
// Measured (tools/probe_batch.py, profiles/r02_probe_batch.txt; ds-7b gate/up, dtk_bench_gemv role 5): k_gemv_b streams its
// weights at 6.2-6.4 TB/s when its x-fragment loads are removed — at 16, 32 AND 64 slots — and loses 2 / 6.5 / 19.6 us to them
// (30.5 / 35.0 / 48.8 us for the full kernel); neither the MFMAs nor the reduction + epilogue cost anything measurable.  A k-step
// needs 1 KiB of x per 16 slots and row tile, read from L2 through the CU's vector-memory path by every wave: at 64 slots that is
// 352 MB per launch next to 180 MB of weights.  Here
//   * a block is KQ x RP waves: RP row groups walk the SAME k-steps, so a 1 KiB x fragment is fetched ONCE per block and stage
//     (L2 -> CU traffic for x / RP) while KQ splits of K keep enough waves on the weight stream;
//   * x never touches a VGPR: each wave issues `global_load_lds_dwordx4` (1 KiB per instruction, straight into a ring of XA + 1
//     LDS slots) XA stages ahead; the MFMA B operand is a ds_read_b128 of the landed fragment;
//   * the weights (fragment-major tiles, non-temporal loads into the A operand registers) run XA stages ahead as well, in a
//     register ring of XA + 1 stages.
// K order per accumulator: k-steps in order inside each of the KQ splits, the splits added in order through LDS — fixed per
// (role, shape), independent of the number of column tiles: a slot's result does not depend on how many slots are active.
// Epilogues and rounding points as in kernels_batch_decode.hip / kernels_decode.hip.
#include "kernels.h"

#include "batch_epi.h"
#include "mx_quant.h"



// ------------------------------------------------------------------------------------------------------------------------
// k_gemv_bx — the rows >> d roles (qkv, gate/up, lm_head) at 49..64 slots with the x operand read ONCE PER CU.
//
// What the probes say (tools/probe_batch.py, profiles/r02_probe_batch_x_traffic.txt): k_gemv_b's weights stream at 6.4 TB/s, its
// MFMAs and epilogue are free, neither occupancy nor deeper register staging changes anything — and feeding all four column
// tiles from ONE x fragment (x traffic / 4, wrong results) takes it from 48.8 to 32.6 us.  The kernel is bound by the bytes of x
// it pulls through the CU's vector-memory path: 688 blocks x 8 waves each read their K slice of all 64 slots = 352 MB per
// launch.  k_gemm_b shared x between waves through LDS with a barrier every 2-4 k-steps and lost the weight stream to the
// barriers (and to the compiler's vmcnt(0) in front of every stage: its waitcnt pass gives up on branchy loops).  Here:
//   * ONE block per CU: UNITS compute waves + one loader wave.  A compute wave owns one unit (= the T = 2 paired row tiles
//     k_gemv_b gives a block) over the FULL K, so there is no cross-wave reduction and the epilogue runs from the accumulator
//     registers;
//   * K is walked in phases of 8 k-steps; the 32 KiB of x fragments of a phase sit in LDS, double buffered, put there by the
//     LOADER wave (phase q + 2 in flight in its registers while it parks phase q + 1): one barrier per phase.  x traffic = one
//     pass over x per CU = 118 MB for gate/up (230 CUs), a third of k_gemv_b's;
//   * the compute waves' vector-memory queue holds nothing but weights — a register ring two phases (16 k-steps, 32 KiB per
//     wave) deep, non-temporal 1 KiB loads, refilled as it is consumed across phases and barriers.  vmcnt completes in order:
//     with x fetched by the same wave (the first version of this kernel: 35.2 us) every wait for an x fragment also waited
//     for the younger half of the weight ring; role-specialised waves keep the two streams' waits apart.  The phase bodies
//     are branch-free (the last two phases are peeled): the compiler's waitcnt pass gives up (vmcnt(0)) on branchy loops;
//   * grid = ceil(units / UNITS) <= CU count: every block is resident at once, a CU streams at most UNITS x K x 64 B of weights
//     (768 KiB for gate/up, 9 % above the fair share) — the launch is HBM-bound as a whole, not per CU.
// K order per accumulator: the k-steps of one k_gemv_b wave slice (CHP phases = nsteps / 8 k-steps) are summed by MFMA
// accumulation from zero in k order, the slice sums are added in slice order in fp32 — the same chains and the same order as
// k_gemv_b's 8 wave slices and their reduction, so the results are BIT-IDENTICAL to k_gemv_b (tested): a slot's tokens do not
// depend on which of the two kernels served the step.
template <bool B> struct bx_flag { static constexpr bool value = B; };
// two e4m3 words (8 weights of one row) -> the bf16 A fragment of one k-step (exact; as in kernels_batch_decode.hip)
__device__ __forceinline__ bf16x8_t gg_f8x8_to_bf16x8(uint32_t w0, uint32_t w1) {
  u32x4 o;
  o[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, false));
  o[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, true));
  o[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, false));
  o[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, true));
  return __builtin_bit_cast(bf16x8_t, o);
}
// F8: the weights come from the fp8 pair-tiled copy (1 KiB = 16 rows x 64 k = two k-steps), are widened to bf16 in registers and
// fed to the same MFMAs in the same k order; the per-row power-of-two scale multiplies the finished sum — bit-identical to the
// fp8 k_gemv_b kernels (whose four-tile variants spill at 64 slots: here a compute wave has the whole register file)
template <int EPI, int UNITS, int CHP, bool F8 = false>
__global__ __launch_bounds__((UNITS + 1) * 64) void k_gemv_bx(GemvBArgs a) {
  constexpr int T = 2, NT = 4, PH = 8, RING = 2 * PH;           // k-steps per phase; weight ring = two phases
  constexpr int FR = PH * NT;                                   // 1 KiB x fragments of one phase
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x FR KiB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = a.K >> 5;                                  // the launcher guarantees K = 32 * 8 * PH * CHP
  const int NPH = nsteps / PH;                                  // 8 * CHP phases (even); a chain (one k_gemv_b wave slice) = CHP phases

  if (wave == UNITS) {
    // ---- the loader wave: x fragments only.  Phase q + 2 is in flight (registers) while phase q + 1 is parked in the buffer
    // nobody reads; its vector-memory queue holds nothing but L2 hits, the compute waves' queues nothing but weights, so
    // neither stream's in-order waits ever touch the other
    const bf16_t* xlane = a.X + lane * 8;
    u32x4 xr[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) xr[f] = *reinterpret_cast<const u32x4*>(xlane + ((size_t)(f / PH) * nsteps + (f % PH)) * 512);
#pragma unroll
    for (int f = 0; f < FR; ++f) *reinterpret_cast<u32x4*>(smem + (size_t)f * 1024 + lane * 16) = xr[f];
#pragma unroll
    for (int f = 0; f < FR; ++f) xr[f] = *reinterpret_cast<const u32x4*>(xlane + ((size_t)(f / PH) * nsteps + PH + (f % PH)) * 512);
    __syncthreads();
    for (int q = 0; q < NPH; ++q) {
      if (q + 1 < NPH) {
        unsigned char* xn = smem + (size_t)((q + 1) & 1) * FR * 1024 + lane * 16;
#pragma unroll
        for (int f = 0; f < FR; ++f) *reinterpret_cast<u32x4*>(xn + (size_t)f * 1024) = xr[f];
      }
      if (q + 2 < NPH) {
#pragma unroll
        for (int f = 0; f < FR; ++f) xr[f] = *reinterpret_cast<const u32x4*>(xlane + ((size_t)(f / PH) * nsteps + (size_t)(q + 2) * PH + (f % PH)) * 512);
      }
      __syncthreads();
    }
    return;
  }

  // ---- compute waves: one unit each over the full K
  const int groups = gg_groups<EPI, T>(a.N, a.ff, a.H, a.KVH);
  const int g = blockIdx.x * UNITS + wave;
  const int gc = g < groups ? g : groups - 1;                   // a surplus wave streams valid memory and stores nothing
  const unsigned char* wrow[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    int tn = gg_tile_row0<EPI, T>(a, gc, t) >> 4;
    const int tn_max = ((a.N + 15) >> 4) - 1;
    if (tn > tn_max) tn = tn_max;
    wrow[t] = F8 ? a.W8 + ((size_t)tn * (nsteps >> 1) * 64 + lane) * 16
                 : reinterpret_cast<const unsigned char*>(a.W) + ((size_t)tn * nsteps * 64 + lane) * 16;
  }
  constexpr int WS = F8 ? 2 : 1;                                 // k-steps per 1 KiB weight tile
  u32x4 wr[RING / WS][T];
#pragma unroll
  for (int i = 0; i < RING / WS; ++i)
#pragma unroll
    for (int t = 0; t < T; ++t) wr[i][t] = ld_nt(reinterpret_cast<const u32x4*>(wrow[t] + (size_t)i * 1024));
  f32x4 tot[T][NT], c[T][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { tot[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __syncthreads();

  // phase q (q & 1 == H): x from buffer H, weights from ring slots H * PH + j; REFILL loads the weights of phase q + 2 into them
  auto phase = [&](int q, auto half_tag, auto refill_tag) {
    constexpr int H = decltype(half_tag)::value ? 1 : 0;
    constexpr bool REFILL = decltype(refill_tag)::value;
    const unsigned char* xb = smem + (size_t)H * FR * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      bf16x8_t xf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xf[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(xb + (size_t)(nt * PH + j) * 1024));
      constexpr int slot_base = H * PH / WS;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const u32x4 wv = wr[slot_base + j / WS][t];
        const bf16x8_t af = F8 ? gg_f8x8_to_bf16x8(wv[2 * (j & 1)], wv[2 * (j & 1) + 1]) : __builtin_bit_cast(bf16x8_t, wv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[nt], c[t][nt], 0, 0, 0);
      }
      if (REFILL && (j % WS) == WS - 1) {                       // the tile just used up
#pragma unroll
        for (int t = 0; t < T; ++t)
          wr[slot_base + j / WS][t] = ld_nt(reinterpret_cast<const u32x4*>(wrow[t] + (size_t)(((q + 2) * PH + j) / WS) * 1024));
      }
    }
    __syncthreads();   // the loader has parked the next phase; everybody is done reading this one
  };
  auto close_chain = [&]() {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { tot[t][nt] += c[t][nt]; c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  };
  constexpr bx_flag<true> yes{};
  constexpr bx_flag<false> no{};
  for (int q = 0; q + 2 < NPH; q += 2) {                        // both phases have two phases after them
    phase(q, no, yes);
    if (CHP == 1) close_chain();
    phase(q + 1, yes, yes);
    close_chain();
  }
  phase(NPH - 2, no, no);
  if (CHP == 1) close_chain();
  phase(NPH - 1, yes, no);
  close_chain();
  if (g >= groups) return;
  // C/D layout: lane holds rows (lane >> 4) * 4 + r, column (slot) nt * 16 + (lane & 15) of each tile
  gg_finish_unit<EPI, T, F8>(a, g, tot, lane);
}

static int g_cu_count = 0;
static int cu_count() {
  if (!g_cu_count) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_cu_count = n;
  }
  return g_cu_count;
}
// ------------------------------------------------------------------------------------------------------------------------
// k_gemv_bl — k_gemv_bx with the WEIGHTS decoupled from the compute waves: a loader wave streams both operands into LDS rings by
// LDS-DMA and the compute waves only ever read LDS.
//
// What round 3's persistent-layer probe measured (profiles/r03_engine2_probe.txt): one LDS-DMA loader wave per CU that runs a few
// fills ahead of v_dot2c consumers, ring + LDS flags instead of barriers, streams 6.4-6.5 TB/s — the batched MFMA GEMVs sit at
// 3.5-4.9 TB/s.  They all stop the weight stream at synchronisation points: k_gemv_bx refills its register ring as it is consumed
// and meets the x loader at a barrier every 8 k-steps, so a wave that waits issues nothing.  Here
//   * ONE loader wave per block issues everything: per phase of 4 k-steps the 16 x fragments of the 64 slots (L2 hits) and the
//     8 weight tiles of each unit (`nt`, HBM) — fragment-major tiles are 1 KiB contiguous = one `global_load_lds_dwordx4`, four
//     consecutive k-steps = one M0 + four instruction offsets (the offset moves the LDS destination with the global source:
//     tools/probe/glds_offset_probe.hip) — into rings of R = 3 phases, two phases in flight (counted vmcnt), no VGPR staging;
//   * the compute waves (one unit = the two paired row tiles, over the FULL K, as in k_gemv_bx) poll an LDS word for "phase p has
//     landed", read A and B fragments with conflict-free ds_read_b128, and publish "phase p consumed"; nobody meets at a barrier;
//   * chains and epilogue exactly as k_gemv_bx: BIT-IDENTICAL to k_gemv_b / k_gemv_bx (tested).
template <bool NONTEMPORAL>
__device__ __forceinline__ void glds_run4(const void* g, unsigned lds_byte) {   // 4 consecutive 1 KiB pieces, global and LDS alike
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);
  if (NONTEMPORAL)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\tglobal_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, off offset:3072 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds_run2_nt(const void* g, unsigned lds_byte) {   // two consecutive pieces (an fp8 pair-tile run of one phase)
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\tglobal_load_lds_dwordx4 %1, off offset:1024 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned bl_ld(unsigned off) {      // LDS control words by explicit ds_read / ds_write (a volatile access through
  unsigned v;                                                   // a generic pointer compiles to FLAT + vmcnt(0): it would drain the DMA queue)
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return v;
}
__device__ __forceinline__ void bl_st(unsigned off, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(off), "v"(v) : "memory"); }
__device__ __forceinline__ void bl_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// a bounded hand-off wait ran out (preemption, a debugger, a protocol bug): the kernel goes on with whatever the ring holds, so the
// step's tokens are garbage — counted in a device word the step's D2H copy carries, and dtk_decode_batch_wait fails on it
__device__ __forceinline__ void bl_timeout(unsigned* err) { if (err) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// F8: the weights come from the fp8 pair-tiled copy (1 KiB = 16 rows x 64 k = two k-steps): half the DMA pieces and ring bytes, widened to
// bf16 in registers and fed to the same MFMAs in the same k order, the per-row power-of-two scale on the finished sum — bit-identical
// to the fp8 k_gemv_bx / k_gemv_b kernels
// XW: the x fragments do NOT go through the LDS-DMA queue.  Leaving parts out showed what bounds the kernel (DESIGN §3.1b,
// profiles/r03_loader_kernel_experiments.txt): every LDS-DMA piece costs the CU ~25-32 ns whichever wave issues it, so a phase of
// 16 x + 24 weight pieces takes ~1 us — and 16 of the 40 were x from L2.  With XW an extra wave brings x in with ordinary 16-byte
// loads (one phase ahead, in registers) and stores it to the ring with ds_write_b128; the DMA queue carries weights only.
// Q3 (QKV of an MHA model, NC = 2): the role has (H + KVH) * 4 RoPE pair units and KVH * 8 V row tiles that need no partner — 1.5
// pair units per CU for ds-7b, which no whole number of units per block balances.  Block b owns pair unit b of the q / k sections
// (wave 0) AND V row tile b (wave 1): 256 blocks x 3 row tiles, every CU busy.
// launch bound 2: a budget of 256 VGPRs keeps the MFMA accumulators in VGPRs; with 512 (one wave per SIMD) the compiler puts
// them in AGPRs and copies all 32 to and fro every phase for the chain adds.
// LW loader waves take alternate phases, R = ring depth in phases.  A loader issues phase p and then waits for its previous phase to
// land before it looks at the next slot, so ONE loader with a ring of 3 keeps two phases in flight and the kernel runs at one phase
// per (memory latency / 2) whatever the phase holds — ~1 us under a saturated stream, for 28 KiB (Q3) as for 40 (gate/up).  More
// phases in flight need a deeper ring (LDS: R x phase bytes) and, because a wave's vmcnt counts at most 63 outstanding pieces, a
// second loader.
template <int EPI, int NC, int CHP4, bool F8 = false, int XW = 0, bool Q3 = false, int LW = 1, int R = 3>     // NC compute waves (units) per block; CHP4 = phases per chain (a k_gemv_b wave slice = 4 * CHP4 k-steps)
__global__ __launch_bounds__((NC + LW + XW) * 64, 2) void k_gemv_bl(GemvBArgs a) {   // XW = 0 / 1 / 2 x waves (2: alternate phases, i.e. two phases of lead each)
  static_assert(!Q3 || (EPI == EPI_QKV && NC == 2), "Q3 is the QKV role with a pair wave and a V wave");
  constexpr int T = 2, NT = 4, PH = 4;
  constexpr int WT = F8 ? PH / 2 : PH;                           // 1 KiB weight tiles per row tile and phase
  constexpr int TILES = Q3 ? 3 : NC * T;                         // weight row tiles per block
  constexpr unsigned XPH = PH * NT * 1024u;                      // x bytes of one phase (16 KiB)
  constexpr unsigned WPH = TILES * WT * 1024u;                   // weight bytes of one phase (4 KiB per row tile; fp8: 2)
  constexpr unsigned OFF_W = R * XPH, OFF_FILLED = OFF_W + R * WPH, OFF_FILLED_X = OFF_FILLED + 4 * LW, OFF_DONE = OFF_FILLED_X + 8;
  constexpr int PIECES = (XW ? 0 : NT * PH) + TILES * WT;        // LDS-DMA instructions per phase
  constexpr unsigned SPIN = 1u << 22;                            // bounded waits: a protocol error must not hang the chip
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the kernel's only LDS object (LDS address 0)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = a.K >> 5, NPH = nsteps / PH;                // the launcher guarantees K = 32 * 8 * 4 * CHP4
  if (threadIdx.x == 0) { for (unsigned o = 0; o < 4u * (NC + LW + 2); o += 4) bl_st(OFF_FILLED + o, 0u); bl_drain(); }
  __syncthreads();
  const int groups = Q3 ? (a.H + a.KVH) * 4 : gg_groups<EPI, T>(a.N, a.ff, a.H, a.KVH);

  auto wait_slot_free = [&](int p) {        // the ring slot of phase p still holds phase p - R: every compute wave must have released it
    if (p < R) return;
    unsigned spins = 0;
    for (; spins < SPIN; ++spins) {
      unsigned lo = bl_ld(OFF_DONE);
#pragma unroll
      for (int c = 1; c < NC; ++c) lo = min(lo, bl_ld(OFF_DONE + 4u * c));
      if (lo + R > (unsigned)p) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (spins == SPIN) bl_timeout(a.err);
  };

  if (wave >= NC && wave < NC + LW) {
    // ---- loader wave l (LDS-DMA): phases l, l + LW, ...
    const int l = wave - NC;
    const unsigned char* xsrc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xsrc[nt] = reinterpret_cast<const unsigned char*>(a.X) + ((size_t)nt * nsteps * 512 + lane * 8) * 2;
    const unsigned char* wsrc[TILES];
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
      int tn;
      if (Q3) {
        tn = j < 2 ? gg_tile_row0<EPI, T>(a, blockIdx.x, j) >> 4 : (a.H + a.KVH) * 8 + (int)blockIdx.x;
      } else {
        const int g = blockIdx.x * NC + j / T, gc = g < groups ? g : groups - 1;    // a surplus unit streams valid memory and stores nothing
        tn = gg_tile_row0<EPI, T>(a, gc, j % T) >> 4;
      }
      const int tn_max = ((a.N + 15) >> 4) - 1;
      if (tn > tn_max) tn = tn_max;
      wsrc[j] = F8 ? a.W8 + ((size_t)tn * (nsteps >> 1) * 64 + lane) * 16
                   : reinterpret_cast<const unsigned char*>(a.W) + ((size_t)tn * nsteps * 64 + lane) * 16;
    }
    unsigned slot = (unsigned)l % R;
    int own = 0;                            // own phases issued so far
    for (int p = l; p < NPH; p += LW, ++own) {
      wait_slot_free(p);
      const size_t adv = (size_t)p * PH * 1024;
      if (!XW) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) glds_run4<false>(xsrc[nt] + adv, slot * XPH + (unsigned)nt * PH * 1024u);
      }
#pragma unroll
      for (int j = 0; j < TILES; ++j) {
        if (F8) glds_run2_nt(wsrc[j] + (size_t)p * WT * 1024, OFF_W + slot * WPH + (unsigned)j * WT * 1024u);
        else glds_run4<true>(wsrc[j] + adv, OFF_W + slot * WPH + (unsigned)j * PH * 1024u);
      }
      if (own >= 1) {                       // this wave's previous phase has landed when only this phase's loads are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
        bl_st(OFF_FILLED + 4u * (unsigned)l, (unsigned)own);
      }
      slot = (slot + LW) % R;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bl_st(OFF_FILLED + 4u * (unsigned)l, (unsigned)own);
    return;
  }

  if (XW && wave >= NC + LW) {
    // ---- x wave xi: phases xi, xi + XW, ...; its next phase is loaded into registers while the current one is stored to the ring
    const int xi = wave - (NC + LW);
    const unsigned char* xsrc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xsrc[nt] = reinterpret_cast<const unsigned char*>(a.X) + ((size_t)nt * nsteps * 512 + lane * 8) * 2;
    u32x4 bufA[NT * PH], bufB[NT * PH];
    auto fetch = [&](u32x4 (&buf)[NT * PH], int p) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < PH; ++j) buf[nt * PH + j] = *reinterpret_cast<const u32x4*>(xsrc[nt] + ((size_t)p * PH + j) * 1024);
    };
    auto put = [&](const u32x4 (&buf)[NT * PH], int p, int own) {
      wait_slot_free(p);
      unsigned char* dst = smem + (unsigned)(p % R) * XPH + lane * 16;
#pragma unroll
      for (int i = 0; i < NT * PH; ++i) *reinterpret_cast<u32x4*>(dst + (size_t)i * 1024) = buf[i];
      bl_drain();
      bl_st(OFF_FILLED_X + 4u * (unsigned)xi, (unsigned)own);
    };
    constexpr int XS = XW > 0 ? XW : 1;
    if (xi < NPH) fetch(bufA, xi);
    for (int p = xi, own = 0; p < NPH; p += 2 * XS, own += 2) {
      if (p + XS < NPH) fetch(bufB, p + XS);
      put(bufA, p, own + 1);
      if (p + 2 * XS < NPH) fetch(bufA, p + 2 * XS);
      if (p + XS < NPH) put(bufB, p + XS, own + 2);
    }
    return;
  }

  // ---- compute waves: one unit each over the full K, operands from LDS
  const int g = Q3 ? (int)blockIdx.x : blockIdx.x * NC + wave;
  const int myT = (Q3 && wave == 1) ? 1 : T;                  // Q3: wave 1 owns the single V row tile (ring tile 2)
  f32x4 tot[T][NT], c[T][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { tot[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  unsigned slot = 0;
  for (int p = 0; p < NPH; ++p) {
    unsigned spins = 0;
    for (; spins < SPIN; ++spins) {
      bool ok = bl_ld(OFF_FILLED + 4u * (unsigned)(p % LW)) > (unsigned)(p / LW);
      if (XW) ok = ok && bl_ld(OFF_FILLED_X + 4u * (unsigned)(p % (XW > 0 ? XW : 1))) > (unsigned)(p / (XW > 0 ? XW : 1));
      if (ok) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (spins == SPIN) bl_timeout(a.err);
    const unsigned char* xb = smem + slot * XPH + lane * 16;
    const unsigned char* wb = smem + OFF_W + slot * WPH + (unsigned)wave * T * WT * 1024u + lane * 16;
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      bf16x8_t xf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xf[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(xb + (size_t)(nt * PH + j) * 1024));
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (Q3 && t >= myT) continue;
        bf16x8_t af;
        if (F8) {
          const u32x4 wv = *reinterpret_cast<const u32x4*>(wb + (size_t)(t * WT + j / 2) * 1024);
          af = gg_f8x8_to_bf16x8(wv[2 * (j & 1)], wv[2 * (j & 1) + 1]);
        } else {
          af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(wb + (size_t)(t * PH + j) * 1024));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[nt], c[t][nt], 0, 0, 0);
      }
    }
    if ((p + 1) % CHP4 == 0) {              // a k_gemv_b wave slice is complete: slice sums are added in slice order
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { tot[t][nt] += c[t][nt]; c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    bl_drain();                             // every fragment read of this phase has returned
    if (lane == 0) bl_st(OFF_DONE + 4u * (unsigned)wave, (unsigned)p + 1u);
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  if (g >= groups) return;
  if (Q3 && wave == 1) {                    // the V row tile of this block: rows (H + KVH) * 128 + 16 b .. + 15 = dims (b & 7) * 16 .. of V head b >> 3
    const int b = blockIdx.x, head = b >> 3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + (lane & 15);
      if (!a.bs->active[n]) continue;
      bf16_t* dst = a.vcache + (size_t)n * a.kv_slot_stride + ((size_t)head * a.T_max + a.st[n].pos) * 128 + (b & 7) * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = tot[0][nt][r];
        if (F8) v *= a.wscale[(a.H + a.KVH) * 128 + b * 16 + (lane >> 4) * 4 + r];
        dst[r] = f2bf(rbf(v));
      }
    }
    return;
  }
  gg_finish_unit<EPI, T, F8>(a, g, tot, lane);
}

static int g_gemv_xw = -1;                      // x fragments by an extra wave's ordinary loads + ds_write instead of LDS-DMA pieces (k_gemv_bl, k_gemv_bkl)
void set_gemv_xw(int v) { g_gemv_xw = v; }
static int gemv_xw() {                           // 0: x by LDS-DMA, 1 / 2: that many x waves
  if (g_gemv_xw < 0) { const char* e = getenv("DTK_GEMV_XW"); g_gemv_xw = e ? atoi(e) : 0; }
  return g_gemv_xw > 2 ? 2 : g_gemv_xw;
}
template <int EPI, int NC, int CHP4, int XW>
static void launch_bl_one_xw(const GemvBArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (4 * 4 * 1024) + 3 * (NC * 2 * 4 * 1024) + 4 * (NC + 3) + 12;        // (the fp8 kernel needs less; one size for both)
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) {
    DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bl<EPI, NC, CHP4, false, XW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bl<EPI, NC, CHP4, true, XW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  const int groups = gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH);
  constexpr int threads = (NC + 1 + XW) * 64;
  if (a.W8) hipLaunchKernelGGL((k_gemv_bl<EPI, NC, CHP4, true, XW>), dim3((groups + NC - 1) / NC), dim3(threads), lds, s, a);
  else hipLaunchKernelGGL((k_gemv_bl<EPI, NC, CHP4, false, XW>), dim3((groups + NC - 1) / NC), dim3(threads), lds, s, a);
}
template <int EPI, int NC, int CHP4>
static void launch_bl_one(const GemvBArgs& a, hipStream_t s) {
  switch (gemv_xw()) {
    case 1: launch_bl_one_xw<EPI, NC, CHP4, 1>(a, s); break;
    case 2: launch_bl_one_xw<EPI, NC, CHP4, 2>(a, s); break;
    default: launch_bl_one_xw<EPI, NC, CHP4, 0>(a, s);
  }
}
// QKV of an MHA model as a pair unit + a V row tile per block (Q3 above).  Covers H == KVH, K = 2048 / 4096.
static int g_gemv_loaders = -1;                 // loader waves of the Q3 kernel: 1 (ring of 3 phases) or 2 (ring of 5: four 28 KiB phases in flight)
void set_gemv_loaders(int v) { g_gemv_loaders = v; }
static int gemv_loaders() {
  if (g_gemv_loaders < 0) { const char* e = getenv("DTK_GEMV_LOADERS"); g_gemv_loaders = e ? atoi(e) : 1; }
  return g_gemv_loaders >= 2 ? 2 : 1;
}
template <int CHP4, int XW, int LW, int R>
static void launch_bl_q3_xw(const GemvBArgs& a, hipStream_t s) {
  constexpr int lds = R * (4 * 4 * 1024) + R * (3 * 4 * 1024) + 4 * (2 + LW + 2) + 12;
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) {
    DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bl<EPI_QKV, 2, CHP4, false, XW, true, LW, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bl<EPI_QKV, 2, CHP4, true, XW, true, LW, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  const int blocks = (a.H + a.KVH) * 4;
  constexpr int threads = (2 + LW + XW) * 64;
  if (a.W8) hipLaunchKernelGGL((k_gemv_bl<EPI_QKV, 2, CHP4, true, XW, true, LW, R>), dim3(blocks), dim3(threads), lds, s, a);
  else hipLaunchKernelGGL((k_gemv_bl<EPI_QKV, 2, CHP4, false, XW, true, LW, R>), dim3(blocks), dim3(threads), lds, s, a);
}
template <int CHP4>
static void launch_bl_q3(const GemvBArgs& a, hipStream_t s) {
  const bool two = gemv_loaders() == 2;
  switch (gemv_xw()) {
    case 1: if (two) launch_bl_q3_xw<CHP4, 1, 2, 5>(a, s); else launch_bl_q3_xw<CHP4, 1, 1, 3>(a, s); break;
    case 2: if (two) launch_bl_q3_xw<CHP4, 2, 2, 5>(a, s); else launch_bl_q3_xw<CHP4, 2, 1, 3>(a, s); break;
    default: if (two) launch_bl_q3_xw<CHP4, 0, 2, 5>(a, s); else launch_bl_q3_xw<CHP4, 0, 1, 3>(a, s);
  }
}
template <int EPI, int CHP4>
static bool launch_bl_units(int units, const GemvBArgs& a, hipStream_t s) {
  switch (units) {
    case 1: case 2: launch_bl_one<EPI, 2, CHP4>(a, s); return true;
    case 3: launch_bl_one<EPI, 3, CHP4>(a, s); return true;
    case 4: launch_bl_one<EPI, 4, CHP4>(a, s); return true;
    default: return false;
  }
}
// weight loads the compiler does not count (its waitcnt pass merges the ring's loads of earlier loop iterations into "wait for
// nearly everything": vmcnt(7) where 30 loads may stay outstanding) — issued and waited for by hand
__device__ __forceinline__ void br_load_nt(u32x4& dst, const void* src) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(src) : "memory"); }
template <int N>
__device__ __forceinline__ void br_wait(u32x4& a0, u32x4& a1) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(N)); }   // ties the two tiles to the wait
// k_gemv_br — k_gemv_bl with the WEIGHTS back in registers.  What bounds k_gemv_bl is the rate at which one CU's LDS-DMA path lands
// bytes (~25 GB/s from HBM: DESIGN §3.1b); the register path has no such cap.  Here
//   * the loader wave streams the x fragments only (16 KiB per phase of 4 k-steps, L2 hits) into a ring of R = 6 phases, three
//     phases in flight (counted vmcnt), paced by the compute waves' DONE words as in k_gemv_bl;
//   * every compute wave (one unit = two paired row tiles over the full K) loads its OWN weight tiles with non-temporal 16-byte
//     loads into a register ring of WD = 4 phases (32 KiB in flight per wave) and refills a ring slot right after the MFMAs that
//     consumed it; nobody meets at a barrier, so a wave that waits for x keeps its weight loads in flight;
//   * chains and epilogue exactly as k_gemv_bl / k_gemv_bx: BIT-IDENTICAL to k_gemv_b (tested).
// SHIPPED FOR fp8 WEIGHTS AT K = 4096 ONLY: with bf16 weights it is slower than k_gemv_bl (gate/up 35.4 vs 32.6 us), and the K = 2048
// instantiations (CHP4 = 2) need 254 VGPRs + 68 bytes of scratch — a spilled ring register is stored before its hand-issued load has
// landed, which is how the first version faulted on ds-1.3b.  The launcher admits fp8, K = 4096 (170 VGPRs, no scratch).
// WD = phases of the register ring.  A phase of fp8 weights is half the bytes of a bf16 phase, so WD = 4 leaves an fp8 wave with
// 16 KiB in flight (48 KiB per CU with three compute waves: 3.5 TB/s on gate/up, half the bytes of the bf16 kernel in 3/4 of its
// time); WD = 8 restores the 32 KiB per wave of the bf16 form — and measured slower (launcher note below): kept as an experiment.
template <int EPI, int NC, int CHP4, bool F8 = false, int WD = 4>
__global__ __launch_bounds__((NC + 1) * 64, WD == 8 ? 1 : 2) void k_gemv_br(GemvBArgs a) {   // (the 96 KiB x ring admits one block per CU anyway; WD = 8 needs > 256 registers)
  constexpr int T = 2, NT = 4, PH = 4, R = 6, XD = 3;
  static_assert(WD == 4 || WD == 8, "ring depth");
  constexpr int WT = F8 ? PH / 2 : PH;                           // 1 KiB weight tiles per row tile and phase
  constexpr unsigned XPH = PH * NT * 1024u;
  constexpr unsigned OFF_FILLED = R * XPH, OFF_DONE = OFF_FILLED + 4;
  constexpr int XP = NT * PH;                                    // LDS-DMA pieces per phase
  constexpr unsigned SPIN = 1u << 22;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = a.K >> 5, NPH = nsteps / PH;                // the launcher guarantees K = 32 * 8 * 4 * CHP4 (NPH = 16 or 32)
  if (threadIdx.x == 0) { for (unsigned o = 0; o < 4u * (NC + 1); o += 4) bl_st(OFF_FILLED + o, 0u); bl_drain(); }
  __syncthreads();
  const int groups = gg_groups<EPI, T>(a.N, a.ff, a.H, a.KVH);

  if (wave == NC) {
    // ---- loader wave: x only
    const unsigned char* xsrc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xsrc[nt] = reinterpret_cast<const unsigned char*>(a.X) + ((size_t)nt * nsteps * 512 + lane * 8) * 2;
    unsigned slot = 0;
    for (int p = 0; p < NPH; ++p) {
      if (p >= R) {
        unsigned spins = 0;
        for (; spins < SPIN; ++spins) {
          unsigned lo = bl_ld(OFF_DONE);
#pragma unroll
          for (int c = 1; c < NC; ++c) lo = min(lo, bl_ld(OFF_DONE + 4u * c));
          if (lo + R > (unsigned)p) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (spins == SPIN) bl_timeout(a.err);
      }
      const size_t adv = (size_t)p * PH * 1024;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) glds_run4<false>(xsrc[nt] + adv, slot * XPH + (unsigned)nt * PH * 1024u);
      if (p >= XD - 1) {                    // XD phases in flight: phase p - (XD - 1) has landed when only the last XD - 1 phases are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((XD - 1) * XP) : "memory");
        bl_st(OFF_FILLED, (unsigned)(p - (XD - 2)));
      }
      slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bl_st(OFF_FILLED, (unsigned)NPH);
    return;
  }

  // ---- compute waves
  const int g = blockIdx.x * NC + wave, gc = g < groups ? g : groups - 1;     // a surplus wave streams valid memory and stores nothing
  const unsigned char* wrow[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    int tn = gg_tile_row0<EPI, T>(a, gc, t) >> 4;
    const int tn_max = ((a.N + 15) >> 4) - 1;
    if (tn > tn_max) tn = tn_max;
    wrow[t] = F8 ? a.W8 + ((size_t)tn * (nsteps >> 1) * 64 + lane) * 16
                 : reinterpret_cast<const unsigned char*>(a.W) + ((size_t)tn * nsteps * 64 + lane) * 16;
  }
  u32x4 wr[WD][T][WT];
#pragma unroll
  for (int b = 0; b < WD; ++b)
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) br_load_nt(wr[b][t][i], wrow[t] + (size_t)(b * WT + i) * 1024);      // issue order: phase, tile, row tile
  f32x4 tot[T][NT], c[T][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { tot[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  unsigned slot = 0;
  auto group = [&](int p0, auto refill_tag) {       // WD phases on ring slots 0 .. WD - 1; REFILL: each used-up tile is reloaded for phase p + WD
    constexpr bool REFILL = decltype(refill_tag)::value;
#pragma unroll
    for (int b = 0; b < WD; ++b) {
      const int p = p0 + b;
      unsigned spins = 0;
      for (; spins < SPIN; ++spins) {
        if (bl_ld(OFF_FILLED) > (unsigned)p) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (spins == SPIN) bl_timeout(a.err);
      const unsigned char* xb = smem + slot * XPH + lane * 16;
#pragma unroll
      for (int j = 0; j < PH; ++j) {
        bf16x8_t xf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xf[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(xb + (size_t)(nt * PH + j) * 1024));
        // loads issued after tile i of this phase: the rest of the phase, the WD - 1 later phases, and this phase's refills so far —
        // T * ((WT - 1 - i) + (WD - 1) * WT + i) = T * (WD * WT - 1) with refills; without, the later phases are b + 1 .. WD - 1 only
        if (!F8 || !(j & 1)) {
          constexpr int dummy = 0; (void)dummy;
          const int i = F8 ? j / 2 : j;
          if (REFILL) br_wait<T * (WD * WT - 1)>(wr[b][0][i], wr[b][1][i]);
          else {
            switch ((WD - 1 - b) * WT + (WT - 1 - i)) {        // compile-time after unrolling (b, j are unrolled indices)
#define BRW(N) case N: br_wait<T * N>(wr[b][0][i], wr[b][1][i]); break;
              BRW(0) BRW(1) BRW(2) BRW(3) BRW(4) BRW(5) BRW(6) BRW(7) BRW(8) BRW(9) BRW(10) BRW(11) BRW(12) BRW(13) BRW(14) BRW(15)
#undef BRW
            }
          }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
          bf16x8_t af;
          if (F8) { const u32x4 wv = wr[b][t][j / 2]; af = gg_f8x8_to_bf16x8(wv[2 * (j & 1)], wv[2 * (j & 1) + 1]); }
          else af = __builtin_bit_cast(bf16x8_t, wr[b][t][j]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) c[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[nt], c[t][nt], 0, 0, 0);
        }
        if (REFILL && (!F8 || (j & 1))) {   // the tile just used up
          const int i = F8 ? j / 2 : j;
#pragma unroll
          for (int t = 0; t < T; ++t) br_load_nt(wr[b][t][i], wrow[t] + (size_t)((p + WD) * WT + i) * 1024);
        }
      }
      if ((p + 1) % CHP4 == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { tot[t][nt] += c[t][nt]; c[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
      bl_drain();
      if (lane == 0) bl_st(OFF_DONE + 4u * (unsigned)wave, (unsigned)p + 1u);
      slot = slot + 1 == R ? 0 : slot + 1;
    }
  };
  constexpr bx_flag<true> yes{};
  constexpr bx_flag<false> no{};
  for (int p0 = 0; p0 + WD < NPH; p0 += WD) group(p0, yes);      // NPH (16 or 32) is a multiple of WD
  group(NPH - WD, no);
  if (g >= groups) return;
  gg_finish_unit<EPI, T, F8>(a, g, tot, lane);
}
// false = this instantiation must not run: its weight loads are issued by hand, so a register the compiler SPILLS would be stored
// before its load has landed (how the K = 2048 instantiations faulted).  The code object says whether it spills: any private
// (scratch) bytes per thread disqualify the kernel, and the caller falls back to k_gemv_bx / k_gemv_b.
// fp8 ring depth: 4 (default) | 8.  Measured (profiles/r04_batch64_fp8_wd8_kernel_stats.csv): the deeper ring is SLOWER — gate/up
// 30.1 vs 25.7 us, qkv 31.1 vs 24.6 — what bounds these kernels is not the bytes in flight but the compute waves' own instruction
// stream (one wave per SIMD: SQ_ACTIVE_INST + SQ_WAIT_INST = 3/4 of a compute wave's cycles, profiles/r04_batch64_fp8_pmc_sq.csv)
static int g_br_wd = 4;
void set_gemv_br_wd(int v) { g_br_wd = v == 8 ? 8 : 4; }
template <int EPI, int NC, int CHP4, bool F8, int WD>
static bool br_usable(int lds) {
  static int usable = -1;
  if (usable < 0) {
    const void* fn = reinterpret_cast<const void*>(&k_gemv_br<EPI, NC, CHP4, F8, WD>);
    hipFuncAttributes fa;
    const bool ok = hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.localSizeBytes == 0
                    && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    usable = ok ? 1 : 0;
  }
  return usable == 1;
}
template <int EPI, int NC, int CHP4>
static bool launch_br_one(const GemvBArgs& a, hipStream_t s) {
  constexpr int lds = 6 * (4 * 4 * 1024) + 4 * (NC + 1) + 12;
  const int groups = gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH);
  const dim3 grid((groups + NC - 1) / NC), block((NC + 1) * 64);
  if (a.W8) {
    if (NC <= 3) if (g_br_wd == 8 && br_usable<EPI, (NC <= 3 ? NC : 3), CHP4, true, 8>(lds)) { hipLaunchKernelGGL((k_gemv_br<EPI, (NC <= 3 ? NC : 3), CHP4, true, 8>), grid, block, lds, s, a); return true; }   // (5 waves of > 256 registers do not fit: NC = 4 keeps WD = 4)
    if (!br_usable<EPI, NC, CHP4, true, 4>(lds)) return false;
    hipLaunchKernelGGL((k_gemv_br<EPI, NC, CHP4, true, 4>), grid, block, lds, s, a);
    return true;
  }
  if (!br_usable<EPI, NC, CHP4, false, 4>(lds)) return false;
  hipLaunchKernelGGL((k_gemv_br<EPI, NC, CHP4, false, 4>), grid, block, lds, s, a);
  return true;
}
template <int EPI, int CHP4>
static bool launch_br_units(int units, const GemvBArgs& a, hipStream_t s) {
  switch (units) {
    case 1: case 2: return launch_br_one<EPI, 2, CHP4>(a, s);
    case 3: return launch_br_one<EPI, 3, CHP4>(a, s);
    case 4: return launch_br_one<EPI, 4, CHP4>(a, s);
    default: return false;
  }
}
// ------------------------------------------------------------------------------------------------------------------------
// k_gemv_bc — the rows >> d roles at 64 slots with the compute waves split by COLUMN tile (round 6).
//
// Every 64-slot kernel above gives a compute wave a unit (two paired row tiles) and all four 16-slot column tiles, so the x fragments
// of all 64 slots — 16 KiB per phase of 4 k-steps, 512 KiB over K = 4096 — must reach every wave of the block: through an LDS ring
// filled by LDS-DMA.  That ring is what the kernels sit on: a CU's LDS-DMA path lands ~25 GB/s (40 with L2 hits: the guide's
// ldsdma-fill row, HISTORY 3.1b), 512 KiB of x per block therefore cost 13-20 us whatever the weights weigh — qkv takes 27.5 us with
// fp8 weights (50 MB) and 29.4 with bf16 (100 MB), gate/up 29.4 / 34.8.  Here the operands swap paths:
//   * compute wave w owns COLUMN tile w (slots 16 w .. 16 w + 15) of ALL the block's row tiles: its B operand is its own 1 KiB per
//     k-step of the fragment-major x (128 KiB over K = 4096), streamed from L2 straight into a register ring of XD = 4 phases with
//     ordinary 16-byte loads — no LDS, no DMA, nothing shared; the four waves together read x once per block as before;
//   * the WEIGHTS, which all four waves need, go through the LDS ring: one loader wave, LDS-DMA, U x 2 row tiles per phase — 8 KiB
//     (fp8, U = 2) .. 24 KiB (bf16, U = 3) instead of 16 KiB of x plus the weights — and every compute wave reads all of them
//     with conflict-free ds_read_b128;
//   * flags as in k_gemv_bl (FILLED by the loader's counted vmcnt, DONE per compute wave), no barrier in the k loop.
// Per accumulator the same MFMAs on the same operands in the same k order, chains closed at the same k_gemv_b slice boundaries,
// slice sums added in slice order, the epilogue of gg_finish_unit per column tile: BIT-IDENTICAL to k_gemv_b and its twins (tested).
template <int N>
__device__ __forceinline__ void bc_wait(u32x4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }   // ties the fragment to the wait
__device__ __forceinline__ void bc_load(u32x4& dst, const void* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); }
// How far the loader runs ahead (round 6, lease B: the first version kept k_gemv_bl's "issue phase p, wait for phase p - 1" and was
// exactly as fast as k_gemv_bl / k_gemv_br — 1 us per phase whatever the phase held): with TWO phases in flight a CU has 2 x 8 KiB
// (fp8 qkv) of weights outstanding against ~2 us of loaded HBM latency = 8 GB/s per CU, 1.5 TB/s for the chip.  The weight pieces of a
// phase are few here (x no longer rides the DMA queue), so the loader keeps LEAD phases outstanding — as many as a wave's vmcnt
// counter (63 operations) covers, i.e. up to 56 KiB per CU — in a ring of LEAD + 2 slots.
template <int U, bool F8> struct bc_shape {
  static constexpr int PIECES = U * 2 * (F8 ? 2 : 4);
  static constexpr int LEAD = 56 / PIECES < 2 ? 2 : (56 / PIECES > 8 ? 8 : 56 / PIECES);
  static constexpr int R = LEAD + 2;
  static constexpr int LDS = R * PIECES * 1024 + 4 * 5 + 12;
};
__device__ __forceinline__ unsigned bc_min_done(unsigned off) {      // the four compute waves' DONE words in ONE LDS round trip
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return min(min(v[0], v[1]), min(v[2], v[3]));
}
template <int EPI, int U, int CHP4, bool F8 = false>
__global__ __launch_bounds__(5 * 64, 2) void k_gemv_bc(GemvBArgs a) {
  constexpr int T = 2, NC = 4, PH = 4, XD = 4;
  constexpr int WT = F8 ? PH / 2 : PH;                           // 1 KiB weight pieces per row tile and phase
  constexpr int TILES = U * T;
  constexpr int LEAD = bc_shape<U, F8>::LEAD, R = bc_shape<U, F8>::R;
  constexpr unsigned WPH = TILES * WT * 1024u;                   // weight bytes of one phase
  constexpr unsigned OFF_DONE = R * WPH, OFF_FILLED = OFF_DONE + 16;      // (DONE: 16-byte aligned, read as one b128)
  constexpr int PIECES = TILES * WT;
  static_assert(PIECES == bc_shape<U, F8>::PIECES, "shape");
  constexpr unsigned SPIN = 1u << 22;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = a.K >> 5, NPH = nsteps / PH;                // the launcher guarantees K = 32 * 8 * 4 * CHP4 (NPH = 16 or 32)
  if (threadIdx.x == 0) { for (unsigned o = 0; o < 4u * (NC + 1); o += 4) bl_st(OFF_DONE + o, 0u); bl_drain(); }
  __syncthreads();
  const int groups = gg_groups<EPI, T>(a.N, a.ff, a.H, a.KVH);

  if (wave == NC) {
    // ---- loader wave: the weight tiles of the block's U units, phase by phase, by LDS-DMA.  What the flag skeleton alone costs was
    // half of the first version's time (profiles/r06d_bc_probe.txt: 13-16 us per launch with every load and MFMA left out): the loader
    // re-read the four DONE words one LDS round trip at a time before EVERY phase.  Now one b128 read, and only when the cached
    // minimum no longer proves the slot free.
    const unsigned char* wsrc[TILES];
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
      const int g = blockIdx.x * U + j / T, gc = g < groups ? g : groups - 1;      // a surplus unit streams valid memory and stores nothing
      int tn = gg_tile_row0<EPI, T>(a, gc, j % T) >> 4;
      const int tn_max = ((a.N + 15) >> 4) - 1;
      if (tn > tn_max) tn = tn_max;
      wsrc[j] = F8 ? a.W8 + ((size_t)tn * (nsteps >> 1) * 64 + lane) * 16
                   : reinterpret_cast<const unsigned char*>(a.W) + ((size_t)tn * nsteps * 64 + lane) * 16;
    }
    unsigned slot = 0, done = 0;
    for (int p = 0; p < NPH; ++p) {
      if (done + R <= (unsigned)p) {          // slot p % R still holds phase p - R: every compute wave must have released it
        unsigned spins = 0;
        for (; spins < SPIN; ++spins) {
          done = bc_min_done(OFF_DONE);
          if (done + R > (unsigned)p) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (spins == SPIN) bl_timeout(a.err);
      }
#pragma unroll
      for (int j = 0; j < TILES; ++j) {
        if (F8) glds_run2_nt(wsrc[j] + (size_t)p * WT * 1024, slot * WPH + (unsigned)j * WT * 1024u);
        else glds_run4<true>(wsrc[j] + (size_t)p * PH * 1024, slot * WPH + (unsigned)j * PH * 1024u);
      }
      if (p >= LEAD - 1) {                    // at most LEAD - 1 phases stay outstanding: phases 0 .. p - (LEAD - 1) have landed
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((LEAD - 1) * PIECES > 63 ? 63 : (LEAD - 1) * PIECES) : "memory");
        bl_st(OFF_FILLED, (unsigned)(p - (LEAD - 2)));
      }
      slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bl_st(OFF_FILLED, (unsigned)NPH);
    return;
  }

  // ---- compute wave = column tile `wave`
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(a.X) + ((size_t)wave * nsteps * 512 + lane * 8) * 2;
  u32x4 xr[XD][PH];
#pragma unroll
  for (int b = 0; b < XD; ++b)
#pragma unroll
    for (int j = 0; j < PH; ++j) bc_load(xr[b][j], xsrc + (size_t)(b * PH + j) * 1024);
  // what the epilogue reads (active flags, positions, row scales, RoPE entries): requested now, behind the first x fragments, used
  // after the last MFMA (4.8 us per launch when it was two or three dependent round trips at the end)
  gg_pre<EPI, T, F8> pre[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int g = blockIdx.x * U + u;
    gg_pre_load<EPI, T, F8>(a, g < groups ? g : groups - 1, lane, wave, pre[u]);
  }
  f32x4 tot[TILES], c[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) { tot[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; c[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  unsigned slot = 0, filled = 0;
  auto group = [&](int p0, auto refill_tag) {       // XD phases on the x ring's slots 0 .. XD - 1; REFILL: a used-up fragment is reloaded for phase p + XD
    constexpr bool REFILL = decltype(refill_tag)::value;
#pragma unroll
    for (int b = 0; b < XD; ++b) {
      const int p = p0 + b;
      if (filled <= (unsigned)p) {            // (the loader runs LEAD phases ahead: one look usually covers several phases)
        unsigned spins = 0;
        for (; spins < SPIN; ++spins) {
          filled = bl_ld(OFF_FILLED);
          if (filled > (unsigned)p) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (spins == SPIN) bl_timeout(a.err);
      }
      const unsigned char* wb = smem + slot * WPH + lane * 16;
      u32x4 wv[TILES];
#pragma unroll
      for (int j = 0; j < PH; ++j) {
        // loads issued after fragment (b, j): with refills always XD * PH - 1; in the last group the rest of this phase and the later ones
        if (REFILL) bc_wait<XD * PH - 1>(xr[b][j]);
        else {
          switch ((XD - 1 - b) * PH + (PH - 1 - j)) {        // compile-time after unrolling
#define BCW(N) case N: bc_wait<N>(xr[b][j]); break;
            BCW(0) BCW(1) BCW(2) BCW(3) BCW(4) BCW(5) BCW(6) BCW(7) BCW(8) BCW(9) BCW(10) BCW(11) BCW(12) BCW(13) BCW(14) BCW(15)
#undef BCW
          }
        }
        const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, xr[b][j]);
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          bf16x8_t af;
          if (F8) {
            if (!(j & 1)) wv[t] = *reinterpret_cast<const u32x4*>(wb + (size_t)(t * WT + j / 2) * 1024);
            af = gg_f8x8_to_bf16x8(wv[t][2 * (j & 1)], wv[t][2 * (j & 1) + 1]);
          } else {
            af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(wb + (size_t)(t * PH + j) * 1024));
          }
          c[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf, c[t], 0, 0, 0);
        }
        if (REFILL) bc_load(xr[b][j], xsrc + (size_t)((p + XD) * PH + j) * 1024);
      }
      if ((p + 1) % CHP4 == 0) {              // a k_gemv_b wave slice is complete: slice sums are added in slice order
#pragma unroll
        for (int t = 0; t < TILES; ++t) { tot[t] += c[t]; c[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
      bl_drain();                             // every weight fragment of this phase has been read
      if (lane == 0) bl_st(OFF_DONE + 4u * (unsigned)wave, (unsigned)p + 1u);
      slot = slot + 1 == R ? 0 : slot + 1;
    }
  };
  constexpr bx_flag<true> yes{};
  constexpr bx_flag<false> no{};
  for (int p0 = 0; p0 + XD < NPH; p0 += XD) group(p0, yes);      // NPH (16 or 32) is a multiple of XD
  group(NPH - XD, no);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int g = blockIdx.x * U + u;
    if (g >= groups) break;
    const f32x4 unit[T] = {tot[u * T], tot[u * T + 1]};
    gg_pre_store<EPI, T, F8>(a, g, unit, lane, wave, pre[u]);
  }
}
// false = this instantiation must not run (see br_usable: hand-issued loads and a spilling compiler do not mix)
template <int EPI, int U, int CHP4, bool F8>
static bool bc_usable(int lds) {
  static int usable = -1;
  if (usable < 0) {
    const void* fn = reinterpret_cast<const void*>(&k_gemv_bc<EPI, U, CHP4, F8>);
    hipFuncAttributes fa;
    const bool ok = hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.localSizeBytes == 0
                    && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    usable = ok ? 1 : 0;
  }
  return usable == 1;
}
template <int EPI, int U, int CHP4>
static bool launch_bc_one(const GemvBArgs& a, hipStream_t s) {
  const int groups = gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH);
  const dim3 grid((groups + U - 1) / U), block(5 * 64);
  if (a.W8) {
    constexpr int lds = bc_shape<U, true>::LDS;
    if (!bc_usable<EPI, U, CHP4, true>(lds)) return false;
    hipLaunchKernelGGL((k_gemv_bc<EPI, U, CHP4, true>), grid, block, lds, s, a);
  } else {
    constexpr int lds = bc_shape<U, false>::LDS;
    if (!bc_usable<EPI, U, CHP4, false>(lds)) return false;
    hipLaunchKernelGGL((k_gemv_bc<EPI, U, CHP4, false>), grid, block, lds, s, a);
  }
  return true;
}
template <int EPI, int CHP4>
static bool launch_bc_units(int units, const GemvBArgs& a, hipStream_t s) {
  switch (units) {
    case 1: return launch_bc_one<EPI, 1, CHP4>(a, s);
    case 2: return launch_bc_one<EPI, 2, CHP4>(a, s);
    case 3: return launch_bc_one<EPI, 3, CHP4>(a, s);
    default: return launch_bc_one<EPI, 4, CHP4>(a, s);
  }
}
static int g_gemv_bc = -1;
void set_gemv_bc(int v) { g_gemv_bc = v; }
// variant: 0 off; bit 0 qkv, bit 1 gate/up, bit 2 lm_head, bits 4..6 = forced units per block (0 = one CU's share, at most 4); 128 (default) =
// by measurement (profiles/r06f_step_bench.txt, 64 slots): qkv always (cl-7b fp8 3.81 -> 3.67 ms per step, ds-7b 4.05 -> 3.96, ds-1.3b
// 1.78 -> 1.73), gate/up with bf16 weights (ds-1.3b 1.73 -> 1.65; ds-7b neutral; fp8: k_gemv_br's register path stays ahead, 29.4 vs
// 31.9 us), lm_head never (neutral).  false = not covered (fewer than 49 slots, N = d roles, K other than 2048 / 4096, a ragged ff /
// vocabulary): the caller goes on to k_gemv_bl / ...
bool launch_gemv_bc(int epi, const GemvBArgs& a, hipStream_t s) {
  if (g_gemv_bc < 0) { const char* e = getenv("DTK_GEMV_BC"); g_gemv_bc = e ? atoi(e) : 128; }
  if (g_gemv_bc <= 0 || a.nt < 3) return false;
  if (epi != EPI_QKV && epi != EPI_SWIGLU && epi != EPI_LOGITS) return false;
  const int roles = (g_gemv_bc & 128) ? (a.W8 ? 1 : 3) : (g_gemv_bc & 7);
  if (!(roles & (epi == EPI_QKV ? 1 : (epi == EPI_SWIGLU ? 2 : 4)))) return false;
  if (a.K != 4096 && a.K != 2048) return false;
  if (epi == EPI_SWIGLU && (a.ff & 15)) return false;
  if (epi == EPI_LOGITS && (a.N & 31)) return false;
  const int groups = epi == EPI_QKV ? gg_groups<EPI_QKV, 2>(a.N, a.ff, a.H, a.KVH)
                   : epi == EPI_SWIGLU ? gg_groups<EPI_SWIGLU, 2>(a.N, a.ff, a.H, a.KVH) : gg_groups<EPI_LOGITS, 2>(a.N, a.ff, a.H, a.KVH);
  int units = (groups + cu_count() - 1) / cu_count();
  if (units > 4) units = 4;
  if ((g_gemv_bc >> 4) & 7) units = (g_gemv_bc >> 4) & 7;
  if (units > 4) units = 4;
#define BC(E) (a.K == 4096 ? launch_bc_units<E, 4>(units, a, s) : launch_bc_units<E, 2>(units, a, s))
  if (epi == EPI_QKV) return BC(EPI_QKV);
  if (epi == EPI_SWIGLU) return BC(EPI_SWIGLU);
  return BC(EPI_LOGITS);
#undef BC
}

static int g_gemv_bl = -1;
void set_gemv_bl(int v) { g_gemv_bl = v; }
// variant bit 0: gate/up + lm_head, bit 1: qkv (2 units per block), bit 2: also with fp8 weights.  false = not covered (fp8 weights, fewer than 33 slots, N = d roles,
// K other than 2048 / 4096, more than 4 units per CU): the caller goes on to k_gemv_bx / k_gemv_b
bool launch_gemv_bl(int epi, const GemvBArgs& a, hipStream_t s) {
  if (g_gemv_bl < 0) { const char* e = getenv("DTK_GEMV_BL"); g_gemv_bl = e ? atoi(e) : 33; }    // default: bit 0 = gate/up + lm_head (64-slot step 4.49 -> 4.35 ms; qkv has too few units per CU: 4.64), bit 5 = k_gemv_br for fp8 weights
  if (g_gemv_bl <= 0 || a.nt < 3) return false;
  // bit 6 (experiment): bf16 qkv through k_gemv_br too — its 1.5 units per CU leave k_gemv_bl one loader wave for two streams
  if (((a.W8 && (g_gemv_bl & 32)) || (!a.W8 && (g_gemv_bl & 64) && epi == EPI_QKV)) && a.K == 4096 && (epi == EPI_QKV || epi == EPI_SWIGLU || epi == EPI_LOGITS)
      && !(epi == EPI_SWIGLU && (a.ff & 15)) && !(epi == EPI_LOGITS && (a.N & 31))) {
    // fp8 weights through registers (k_gemv_br): gate/up 28.9 -> 25.5 us, qkv 28.1 -> 24.8 (profiles/r03_loader_kernel_experiments.txt)
    const int groups = epi == EPI_QKV ? gg_groups<EPI_QKV, 2>(a.N, a.ff, a.H, a.KVH)
                     : epi == EPI_SWIGLU ? gg_groups<EPI_SWIGLU, 2>(a.N, a.ff, a.H, a.KVH) : gg_groups<EPI_LOGITS, 2>(a.N, a.ff, a.H, a.KVH);
    const int units = (groups + cu_count() - 1) / cu_count();
    if (epi == EPI_QKV) { if (launch_br_units<EPI_QKV, 4>(units, a, s)) return true; }
    else if (epi == EPI_SWIGLU) { if (launch_br_units<EPI_SWIGLU, 4>(units, a, s)) return true; }
    else if (launch_br_units<EPI_LOGITS, 4>(units, a, s)) return true;
  }
  if (a.W8 && !(g_gemv_bl & 4)) return false;     // fp8 weights: bit 2 (measured neutral against the fp8 k_gemv_bx: 4.01 vs 4.04 ms per 64-slot step)
  if (epi != EPI_QKV && epi != EPI_SWIGLU && epi != EPI_LOGITS) return false;
  if (epi == EPI_QKV ? !(g_gemv_bl & (2 | 8 | 16)) : !(g_gemv_bl & 1)) return false;
  if (a.K != 4096 && a.K != 2048) return false;
  if (epi == EPI_QKV && (g_gemv_bl & (8 | 16))) {      // bit 3: a pair unit + a V row tile per block for MHA models whose q / k pair count fills the chip; bit 4: for any MHA model (tests)
    if (a.H == a.KVH && a.N == (a.H + 2 * a.KVH) * 128 && ((g_gemv_bl & 16) || (a.H + a.KVH) * 4 * 4 >= cu_count() * 3)) {
      if (a.K == 4096) launch_bl_q3<4>(a, s); else launch_bl_q3<2>(a, s);
      return true;
    }
    if (!(g_gemv_bl & 2)) return false;
  }
  if (epi == EPI_SWIGLU && (a.ff & 15)) return false;
  if (epi == EPI_LOGITS && (a.N & 31)) return false;
  const int groups = epi == EPI_QKV ? gg_groups<EPI_QKV, 2>(a.N, a.ff, a.H, a.KVH)
                   : epi == EPI_SWIGLU ? gg_groups<EPI_SWIGLU, 2>(a.N, a.ff, a.H, a.KVH) : gg_groups<EPI_LOGITS, 2>(a.N, a.ff, a.H, a.KVH);
  const int units = (groups + cu_count() - 1) / cu_count();
#define BL(E) (a.K == 4096 ? launch_bl_units<E, 4>(units, a, s) : launch_bl_units<E, 2>(units, a, s))
  if (epi == EPI_QKV) return BL(EPI_QKV);
  if (epi == EPI_SWIGLU) return BL(EPI_SWIGLU);
  return BL(EPI_LOGITS);
#undef BL
}

template <int EPI, int UNITS, int CHP>
static void launch_bx_one(const GemvBArgs& a, hipStream_t s) {
  constexpr int lds = 2 * 8 * 4 * 1024;
  static unsigned long long attr_set = 0;
  if (dtk_lds_attr_todo(attr_set)) {
    DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bx<EPI, UNITS, CHP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bx<EPI, UNITS, CHP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  const int groups = gg_groups<EPI, 2>(a.N, a.ff, a.H, a.KVH);
  if (a.W8) hipLaunchKernelGGL((k_gemv_bx<EPI, UNITS, CHP, true>), dim3((groups + UNITS - 1) / UNITS), dim3((UNITS + 1) * 64), lds, s, a);
  else hipLaunchKernelGGL((k_gemv_bx<EPI, UNITS, CHP, false>), dim3((groups + UNITS - 1) / UNITS), dim3((UNITS + 1) * 64), lds, s, a);
}
template <int EPI, int CHP>
static bool launch_bx_units(int units, const GemvBArgs& a, hipStream_t s) {
  switch (units) {
    case 1: case 2: launch_bx_one<EPI, 2, CHP>(a, s); return true;   // (one unit per block would share nothing)
    case 3: launch_bx_one<EPI, 3, CHP>(a, s); return true;
    case 4: launch_bx_one<EPI, 4, CHP>(a, s); return true;
    default: return false;                                           // more than 4 units per CU (128 k-token vocabularies): k_gemv_b
  }
}
// variant > 0 = on; `units` waves per block = units per CU (variant 2..4 force that many: tuning).  false = not covered (fewer
// than 49 slots, N = d roles, K other than 2048 / 4096, more than 4 units per CU): the caller uses k_gemv_b
bool launch_gemv_bx(int epi, int variant, const GemvBArgs& a, hipStream_t s) {
  if (a.nt < 3 || variant <= 0) return false;
  if (epi != EPI_QKV && epi != EPI_SWIGLU && epi != EPI_LOGITS) return false;
  if (a.K != 4096 && a.K != 2048) return false;
  if (epi == EPI_SWIGLU && (a.ff & 15)) return false;
  if (epi == EPI_LOGITS && (a.N & 31)) return false;
  const int groups = epi == EPI_QKV ? gg_groups<EPI_QKV, 2>(a.N, a.ff, a.H, a.KVH)
                   : epi == EPI_SWIGLU ? gg_groups<EPI_SWIGLU, 2>(a.N, a.ff, a.H, a.KVH) : gg_groups<EPI_LOGITS, 2>(a.N, a.ff, a.H, a.KVH);
  int units = (groups + cu_count() - 1) / cu_count();
  // measured at 64 slots, ds-7b (profiles/r02_batch64_bx_kernel_stats.csv vs r02_batch64_kernel_stats.csv): gate/up (3 units per CU)
  // 48.7 -> 36.3 us, lm_head (4) 60.3 -> 49.5, qkv (1.5 -> 2 units: 192 blocks of 2 compute waves) 28.1 -> 31.3: with fewer than
  // 3 units a CU has too few weight streams in flight, k_gemv_b keeps those roles
  if (variant == 1 && units < 3) return false;
  if (variant >= 2 && variant <= 4) units = variant;
#define BX(E) (a.K == 4096 ? launch_bx_units<E, 2>(units, a, s) : launch_bx_units<E, 1>(units, a, s))
  if (epi == EPI_QKV) return BX(EPI_QKV);
  if (epi == EPI_SWIGLU) return BX(EPI_SWIGLU);
  return BX(EPI_LOGITS);
#undef BX
}


// ------------------------------------------------------------------------------------------------------------------------
// k_gemv_bkp + k_resid_norm_b — the N = d roles at 33..64 slots as TWO launches: K split over the CUs of a row group, the
// partials met by the NEXT kernel of the step instead of inside this one.
//
// k_gemv_bk (above) has the right traffic shape — x / 8 — and lost to its in-kernel exchange: publish -> drain -> ticket ->
// read-back -> residual read-modify-write is ~8 us of dependent memory round trips at the tail of every row tile
// (profiles/r02_batch64_bk_kernel_stats.csv).  The step already HAS a kernel right behind each N = d role: the RMSNorm of the
// residual stream it has just updated (post_attention_layernorm after o_proj, the next layer's input_layernorm — or the final
// norm — after down), a 64-block launch that reads every x element anyway.  So:
//   * k_gemv_bkp = k_gemv_bk's compute, bit for bit (same K slices, same MFMA chains), whose waves simply STORE their
//     16 x 64 fp32 partial, slot-major ([slice][slot][row]: a lane's 4 rows are one 16-byte store, kernel B reads rows
//     contiguously) — no atomics, no tickets, no fences: the kernel boundary is the hand-off;
//   * k_resid_norm_b (one block per slot) adds the 8 partials of every row in slice order (k_gemv_b's LDS reduction order),
//     applies the residual epilogue (x += bf16(sum), HF rounding), and normalises the updated row in the same pass — sum of
//     squares in k_rmsnorm_b's exact order (chunk c by "virtual thread" c mod 256, chained through LDS across the block's
//     256-thread rounds), so x, the scale and the fragment-major xn are BIT-IDENTICAL to k_gemv_b<RESID> + k_rmsnorm_b.
// Launch count per layer is unchanged (7); the N = d kernels stop reading all of x in every block (down: 360 -> 45 MB through
// L2) and lose their reduction + read-modify-write tail; the norm kernel reads 8 MB of partials (L2 / MALL resident) more.
template <int TPG>
__global__ __launch_bounds__((TPG + 1) * 64) void k_gemv_bkp(GemvBArgs a) {
  constexpr int NT = 4, PH = 8, FR = PH * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x FR KiB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = (a.K + 31) >> 5;
  const int per = (nsteps + 7) >> 3;
  const int b = blockIdx.x, idx = b >> 3;
  const int rgs_per_xcd = (int)(gridDim.x >> 6);                // grid = 8 XCDs x rgs_per_xcd row groups x 8 slices
  const int rg = (b & 7) * rgs_per_xcd + (idx >> 3), ks = idx & 7;
  const int s0 = min(nsteps, ks * per), s1 = min(nsteps, s0 + per);
  const int Lc = s1 - s0;                                       // >= 1 (launcher)
  const int nph = (Lc + PH - 1) / PH;

  if (wave == TPG) {   // ---- loader wave: the slice's x fragments, one phase (8 k-steps x 4 column tiles = 32 KiB) at a time
    const bf16_t* xlane = a.X + lane * 8;
    auto src = [&](int q, int f) { return xlane + ((size_t)(f / PH) * nsteps + min(s0 + q * PH + (f % PH), s1 - 1)) * 512; };
    u32x4 xr[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) xr[f] = *reinterpret_cast<const u32x4*>(src(0, f));
#pragma unroll
    for (int f = 0; f < FR; ++f) *reinterpret_cast<u32x4*>(smem + (size_t)f * 1024 + lane * 16) = xr[f];
    if (nph > 1) {
#pragma unroll
      for (int f = 0; f < FR; ++f) xr[f] = *reinterpret_cast<const u32x4*>(src(1, f));
    }
    __syncthreads();
    for (int q = 0; q < nph; ++q) {
      if (q + 1 < nph) {
        unsigned char* xn = smem + (size_t)((q + 1) & 1) * FR * 1024 + lane * 16;
#pragma unroll
        for (int f = 0; f < FR; ++f) *reinterpret_cast<u32x4*>(xn + (size_t)f * 1024) = xr[f];
      }
      if (q + 2 < nph) {
#pragma unroll
        for (int f = 0; f < FR; ++f) xr[f] = *reinterpret_cast<const u32x4*>(src(q + 2, f));
      }
      __syncthreads();
    }
    return;
  }

  // ---- compute waves: wave w owns row tile rg * TPG + w over this block's K slice.  The weights sit in a register ring of TWO
  // phases (16 k-steps, 16 KiB per wave, 128 KiB per CU in flight — k_gemv_bk's ring was one phase): phase p computes from one half
  // while the other half already holds phase p + 1 and this half is refilled for phase p + 2 as its k-steps are used up.  o_proj
  // (16 k-steps per slice) has its whole slice in flight before the first MFMA.  Same chains, same order: bit-identical partials.
  const int tn = rg * TPG + wave;
  const unsigned char* wrow = reinterpret_cast<const unsigned char*>(a.W) + ((size_t)tn * nsteps * 64 + lane) * 16 + (size_t)s0 * 1024;
  u32x4 wa[PH], wb[PH];
#pragma unroll
  for (int i = 0; i < PH; ++i) wa[i] = ld_nt(reinterpret_cast<const u32x4*>(wrow + (size_t)min(i, Lc - 1) * 1024));
  if (nph > 1) {
#pragma unroll
    for (int i = 0; i < PH; ++i) wb[i] = ld_nt(reinterpret_cast<const u32x4*>(wrow + (size_t)min(PH + i, Lc - 1) * 1024));
  }
  f32x4 c[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) c[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  auto phase = [&](int p, u32x4 (&w)[PH], auto refill_tag) {
    constexpr bool REFILL = decltype(refill_tag)::value;
    const unsigned char* xb = smem + (size_t)(p & 1) * FR * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      bf16x8_t xf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xf[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(xb + (size_t)(nt * PH + j) * 1024));
      u32x4 wv = w[j];
      if (p * PH + j >= Lc) wv = (u32x4){0u, 0u, 0u, 0u};          // k-steps past the end of the slice contribute nothing
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, wv);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) c[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[nt], c[nt], 0, 0, 0);
      if (REFILL) w[j] = ld_nt(reinterpret_cast<const u32x4*>(wrow + (size_t)min((p + 2) * PH + j, Lc - 1) * 1024));
    }
    __syncthreads();   // the loader has parked the next phase; everybody is done reading this one
  };
  constexpr bx_flag<true> yes{};
  constexpr bx_flag<false> no{};
  int p = 0;
  for (; p + 3 < nph; p += 2) { phase(p, wa, yes); phase(p + 1, wb, yes); }     // both phases have a phase two steps on
  const int rest = nph - p;                                                      // 1, 2 or 3 phases left
  if (rest == 3) { phase(p, wa, yes); phase(p + 1, wb, no); phase(p + 2, wa, no); }
  else if (rest == 2) { phase(p, wa, no); phase(p + 1, wb, no); }
  else phase(p, wa, no);
  // ---- the partial of (slice ks, tile tn): lane holds rows tn*16 + (lane>>4)*4 + 0..3 of slot nt*16 + (lane&15)
  float* out = a.kpart + ((size_t)ks * 64 + (lane & 15)) * a.N + tn * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(out + (size_t)nt * 16 * a.N) = c[nt];
}

// k_gemv_bkl — k_gemv_bkp with both operands through LDS rings filled by a loader wave (the k_gemv_bl treatment for the N = d
// roles): block (row group, K slice) = TPG compute waves, one row tile each, + one loader wave that streams the slice's x
// fragments (16 KiB per phase of 4 k-steps) and the TPG weight tiles (4 KiB each per phase) by LDS-DMA, two phases in flight, ring
// of 3.  Same K slices, same MFMA chains, same partial layout as k_gemv_bkp: bit-identical.  A slice that is not a whole number
// of phases (down: 43 k-steps) ends in a phase whose surplus k-steps are fetched clamped and not multiplied.
__device__ __forceinline__ void glds16_any(const void* gsrc, unsigned lds_byte, bool nontemporal) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);
  if (nontemporal)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// F8: the weights are the fp8 pair tiles (1 KiB = 16 rows x two k-steps): half the bytes per phase, widened to bf16 in registers
// and fed to the same MFMA chain; the per-row 2^e scale multiplies the partial (exact), so the 8 partials + the residual that
// k_resid_norm_b adds are bit for bit what k_gemv_b<RESID, fp8> reduces through LDS.  A K slice that starts at an odd k-step
// (down: 43 k-steps per slice) fetches the pair tile it shares with its left neighbour and skips that neighbour's k-step.
template <int TPG, int XW = 0, bool F8 = false>       // XW x waves: x by ordinary loads + ds_write_b128 (see k_gemv_bl)
__global__ __launch_bounds__((TPG + 1 + XW) * 64) void k_gemv_bkl(GemvBArgs a) {
  constexpr int NT = 4, PH = 4, R = 3;
  constexpr int WT = F8 ? PH / 2 : PH;                           // 1 KiB weight pieces per row tile and phase
  constexpr unsigned XPH = PH * NT * 1024u, WPH = TPG * WT * 1024u;
  constexpr unsigned OFF_W = R * XPH, OFF_FILLED = OFF_W + R * WPH, OFF_FILLED_X = OFF_FILLED + 4, OFF_DONE = OFF_FILLED + 12;
  constexpr int PIECES = (XW ? 0 : NT) * PH + TPG * WT;
  constexpr unsigned SPIN = 1u << 22;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = (a.K + 31) >> 5;
  const int per = (nsteps + 7) >> 3;
  const int b = blockIdx.x, idx = b >> 3;
  const int rgs_per_xcd = (int)(gridDim.x >> 6);                // grid = 8 XCDs x rgs_per_xcd row groups x 8 slices
  const int rg = (b & 7) * rgs_per_xcd + (idx >> 3), ks = idx & 7;
  const int s0 = min(nsteps, ks * per), s1 = min(nsteps, s0 + per);
  const int e0 = F8 ? (s0 & ~1) : s0;                           // first k-step FETCHED (fp8 pair tiles start at even k-steps)
  const int Lc = s1 - e0;                                       // >= 1 (launcher); k-steps e0 .. e0 + skip - 1 belong to the left neighbour
  const int skip = s0 - e0;
  const int nph = (Lc + PH - 1) / PH;
  if (threadIdx.x == 0) { for (unsigned o = 0; o < 4u * (TPG + 3); o += 4) bl_st(OFF_FILLED + o, 0u); bl_drain(); }
  __syncthreads();
  auto wait_slot_free = [&](int p) {
    if (p < R) return;
    unsigned spins = 0;
    for (; spins < SPIN; ++spins) {
      unsigned lo = bl_ld(OFF_DONE);
#pragma unroll
      for (int c = 1; c < TPG; ++c) lo = min(lo, bl_ld(OFF_DONE + 4u * c));
      if (lo + R > (unsigned)p) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (spins == SPIN) bl_timeout(a.err);
  };

  if (wave == TPG) {   // ---- loader wave
    const unsigned char* xsrc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xsrc[nt] = reinterpret_cast<const unsigned char*>(a.X) + (((size_t)nt * nsteps + e0) * 512 + lane * 8) * 2;
    const int upr = F8 ? (nsteps + 1) >> 1 : nsteps;            // 1 KiB pieces per row tile (fp8: pair tiles)
    const unsigned char* wbase = F8 ? a.W8 + (((size_t)rg * TPG * upr + (e0 >> 1)) * 64 + lane) * 16       // tile rg*TPG + w at + w * upr KiB
                                    : reinterpret_cast<const unsigned char*>(a.W) + (((size_t)rg * TPG * nsteps + s0) * 64 + lane) * 16;
    const int last_u = F8 ? ((s1 - 1) >> 1) - (e0 >> 1) : Lc - 1;   // last piece of the slice, relative to wbase
    unsigned slot = 0;
    for (int p = 0; p < nph; ++p) {
      wait_slot_free(p);
      const size_t adv = (size_t)p * PH * 1024;
      if ((p + 1) * PH <= Lc) {               // a whole phase inside the slice: runs of four consecutive pieces
        if (!XW) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) glds_run4<false>(xsrc[nt] + adv, slot * XPH + (unsigned)nt * PH * 1024u);
        }
        if (F8) {
#pragma unroll
          for (int w = 0; w < TPG; ++w)
#pragma unroll
            for (int i = 0; i < WT; ++i)
              glds16_any(wbase + ((size_t)w * upr + (size_t)(p * WT + i)) * 1024, OFF_W + slot * WPH + (unsigned)(w * WT + i) * 1024u, true);
        } else {
#pragma unroll
          for (int w = 0; w < TPG; ++w) glds_run4<true>(wbase + (size_t)w * nsteps * 1024 + adv, OFF_W + slot * WPH + (unsigned)w * PH * 1024u);
        }
      } else {                                // the ragged last phase: piece by piece, clamped to the slice's last k-step
#pragma unroll
        for (int j = 0; j < PH; ++j) {
          const size_t kk = (size_t)min(p * PH + j, Lc - 1) * 1024;
          if (!XW) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) glds16_any(xsrc[nt] + kk, slot * XPH + (unsigned)(nt * PH + j) * 1024u, false);
          }
          if (!F8) {
#pragma unroll
            for (int w = 0; w < TPG; ++w) glds16_any(wbase + (size_t)w * nsteps * 1024 + kk, OFF_W + slot * WPH + (unsigned)(w * PH + j) * 1024u, true);
          }
        }
        if (F8) {
#pragma unroll
          for (int w = 0; w < TPG; ++w)
#pragma unroll
            for (int i = 0; i < WT; ++i)
              glds16_any(wbase + ((size_t)w * upr + (size_t)min(p * WT + i, last_u)) * 1024, OFF_W + slot * WPH + (unsigned)(w * WT + i) * 1024u, true);
        }
      }
      if (p >= 1) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
        bl_st(OFF_FILLED, (unsigned)p);
      }
      slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bl_st(OFF_FILLED, (unsigned)nph);
    return;
  }

  if (XW && wave > TPG) {   // ---- x wave xi: phases xi, xi + XW, ...; the next own phase in registers while the current one is stored
    const int xi = wave - (TPG + 1);
    const unsigned char* xsrc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xsrc[nt] = reinterpret_cast<const unsigned char*>(a.X) + (((size_t)nt * nsteps + e0) * 512 + lane * 8) * 2;
    u32x4 bufA[NT * PH], bufB[NT * PH];
    auto fetch = [&](u32x4 (&buf)[NT * PH], int p) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < PH; ++j) buf[nt * PH + j] = *reinterpret_cast<const u32x4*>(xsrc[nt] + (size_t)min(p * PH + j, Lc - 1) * 1024);
    };
    auto put = [&](const u32x4 (&buf)[NT * PH], int p, int own) {
      wait_slot_free(p);
      unsigned char* dst = smem + (unsigned)(p % R) * XPH + lane * 16;
#pragma unroll
      for (int i = 0; i < NT * PH; ++i) *reinterpret_cast<u32x4*>(dst + (size_t)i * 1024) = buf[i];
      bl_drain();
      bl_st(OFF_FILLED_X + 4u * (unsigned)xi, (unsigned)own);
    };
    constexpr int XS = XW > 0 ? XW : 1;
    if (xi < nph) fetch(bufA, xi);
    for (int p = xi, own = 0; p < nph; p += 2 * XS, own += 2) {
      if (p + XS < nph) fetch(bufB, p + XS);
      put(bufA, p, own + 1);
      if (p + 2 * XS < nph) fetch(bufA, p + 2 * XS);
      if (p + XS < nph) put(bufB, p + XS, own + 2);
    }
    return;
  }

  // ---- compute waves: wave w owns row tile rg * TPG + w over this block's K slice
  const int tn = rg * TPG + wave;
  f32x4 c[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) c[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned slot = 0;
  for (int p = 0; p < nph; ++p) {
    unsigned spins = 0;
    for (; spins < SPIN; ++spins) {
      bool ok = bl_ld(OFF_FILLED) > (unsigned)p;
      if (XW) ok = ok && bl_ld(OFF_FILLED_X + 4u * (unsigned)(p % (XW > 0 ? XW : 1))) > (unsigned)(p / (XW > 0 ? XW : 1));
      if (ok) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (spins == SPIN) bl_timeout(a.err);
    const unsigned char* xb = smem + slot * XPH + lane * 16;
    const unsigned char* wb = smem + OFF_W + slot * WPH + (unsigned)wave * WT * 1024u + lane * 16;
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      if (p * PH + j < Lc && p * PH + j >= skip) {     // wave-uniform: k-steps outside the slice are not multiplied
        bf16x8_t af;
        if (F8) { const u32x4 wv = *reinterpret_cast<const u32x4*>(wb + (size_t)(j >> 1) * 1024); af = gg_f8x8_to_bf16x8(wv[2 * (j & 1)], wv[2 * (j & 1) + 1]); }
        else af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(wb + (size_t)j * 1024));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(xb + (size_t)(nt * PH + j) * 1024));
          c[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf, c[nt], 0, 0, 0);
        }
      }
    }
    bl_drain();
    if (lane == 0) bl_st(OFF_DONE + 4u * (unsigned)wave, (unsigned)p + 1u);
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  if (F8) {                                                      // per-row power-of-two scale: exact on the partial
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.wscale + tn * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) c[nt] *= sc;
  }
  float* out = a.kpart + ((size_t)ks * 64 + (lane & 15)) * a.N + tn * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(out + (size_t)nt * 16 * a.N) = c[nt];
}
static int g_gemv_bkl = -1;
void set_gemv_bkl(int v) { g_gemv_bkl = v; }
static bool launch_gemv_bkl(const GemvBArgs& a, hipStream_t s) {
  if (g_gemv_bkl < 0) { const char* e = getenv("DTK_GEMV_BKL"); g_gemv_bkl = e ? atoi(e) : 1; }   // default on: 64-slot step 4.35 -> 4.26 ms
  if (g_gemv_bkl <= 0) return false;
  const int xw = gemv_xw();
#define BKL_ATTR(TPG_, XW_) do { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bkl<TPG_, XW_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
                                 DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bkl<TPG_, XW_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); } while (0)
#define BKL_GO(TPG_, XW_) do { if (a.W8) hipLaunchKernelGGL((k_gemv_bkl<TPG_, XW_, true>), dim3(256), dim3((TPG_ + 1 + XW_) * 64), lds, s, a); \
                               else hipLaunchKernelGGL((k_gemv_bkl<TPG_, XW_, false>), dim3(256), dim3((TPG_ + 1 + XW_) * 64), lds, s, a); } while (0)
  if (((a.N + 15) >> 4) == 256) {
    constexpr int lds = 3 * (16 + 8 * 4) * 1024 + 4 * 11 + 12;
    static unsigned long long attr8 = 0;
    if (dtk_lds_attr_todo(attr8)) { BKL_ATTR(8, 0); BKL_ATTR(8, 1); BKL_ATTR(8, 2); }
    if (xw == 2) BKL_GO(8, 2); else if (xw == 1) BKL_GO(8, 1); else BKL_GO(8, 0);
  } else {
    constexpr int lds = 3 * (16 + 4 * 4) * 1024 + 4 * 7 + 12;
    static unsigned long long attr4 = 0;
    if (dtk_lds_attr_todo(attr4)) { BKL_ATTR(4, 0); BKL_ATTR(4, 1); BKL_ATTR(4, 2); }
    if (xw == 2) BKL_GO(4, 2); else if (xw == 1) BKL_GO(4, 1); else BKL_GO(4, 0);
  }
#undef BKL_GO
#undef BKL_ATTR
  return true;
}

// false = not covered (fewer than 33 slots, a tile count that is not 32 row groups of 4 or 8 tiles, a K that
// leaves one of the 8 slices empty, a width k_resid_norm_b does not handle): the caller uses k_gemv_b<RESID> + k_rmsnorm_b
static bool gemv_bkl_on() {
  if (g_gemv_bkl < 0) { const char* e = getenv("DTK_GEMV_BKL"); g_gemv_bkl = e ? atoi(e) : 1; }
  return g_gemv_bkl > 0;
}
bool resid_kparts_covers(const GemvBArgs& a) {
  if (a.nt < 3 || !a.kpart) return false;
  if (a.W8 && !gemv_bkl_on()) return false;      // fp8 weights: only the LDS-ring kernel reads the pair tiles
  const int ntiles = (a.N + 15) >> 4, nsteps = (a.K + 31) >> 5, per = (nsteps + 7) >> 3;
  if ((a.N & 15) || (a.K & 31) || 7 * per >= nsteps) return false;
  if (ntiles != 256 && ntiles != 128) return false;
  const int D8 = a.N >> 3;
  return D8 == 256 || D8 == 512 || D8 == 1024;
}
void launch_gemv_bkp(const GemvBArgs& a, hipStream_t s) {
  if (launch_gemv_bkl(a, s)) return;       // operands through LDS rings filled by a loader wave (option gemv_bkl)
  constexpr int lds = 2 * 8 * 4 * 1024;
  if (((a.N + 15) >> 4) == 256) {
    static unsigned long long attr8 = 0;
    if (dtk_lds_attr_todo(attr8)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bkp<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
    hipLaunchKernelGGL((k_gemv_bkp<8>), dim3(256), dim3(9 * 64), lds, s, a);
  } else {
    static unsigned long long attr4 = 0;
    if (dtk_lds_attr_todo(attr4)) { DTK_LDS_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_bkp<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
    hipLaunchKernelGGL((k_gemv_bkp<4>), dim3(256), dim3(5 * 64), lds, s, a);
  }
}

// One block per slot, D / 8 threads (thread c owns rows 8c .. 8c+7).  part: [8 slices][64 slots][D] fp32 (k_gemv_bkp);
// X: the residual streams [slot][ldx] (updated in place); Y: the normalised rows, fragment-major (the next GEMV's B operand).
template <int ROUNDS>
__global__ __launch_bounds__(ROUNDS * 256) void k_resid_norm_b(const float* part, bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int D, float eps,
                                                               const BatchState* bs, uint8_t* Y8, uint8_t* YS) {
  const int slot = blockIdx.x;
  if (!bs->active[slot]) return;
  __shared__ float chain[256];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, c = tid, round = tid >> 8;
  // every load of the thread goes out before the first use: 16 partial pieces + x + the norm weight = one memory round trip
  f32x4 p0[8], p1[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const f32x4* src = reinterpret_cast<const f32x4*>(part + ((size_t)ks * 64 + slot) * D + c * 8);
    p0[ks] = src[0]; p1[ks] = src[1];
  }
  u32x4* xrow = reinterpret_cast<u32x4*>(X + (size_t)slot * ldx) + c;
  const u32x4 xv = *xrow;
  const u32x4 g = reinterpret_cast<const u32x4*>(w)[c];
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float sum = 0.f;                                    // slice order 0..7 from zero: k_gemv_b's cross-wave reduction
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) sum += (e < 4 ? p0[ks][e] : p1[ks][e - 4]);
    const float res = (e & 1) ? pk_hi(xv[e >> 1]) : pk_lo(xv[e >> 1]);
    y[e] = rbf(res + rbf(sum));                         // HF: hidden = residual + proj(x), the projection a bf16 tensor
  }
  u32x4 yo;
#pragma unroll
  for (int e = 0; e < 4; ++e) yo[e] = pack2(y[2 * e], y[2 * e + 1]);
  *xrow = yo;
  // sum of squares in k_rmsnorm_b's order: its thread t folds chunk t, then chunk t + 256, ... into ONE accumulator
  float ss = 0.f;
  for (int r = 0; r < ROUNDS; ++r) {
    if (round == r) {
      if (r > 0) ss = chain[tid & 255];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ss = __builtin_fmaf(y[2 * e], y[2 * e], ss); ss = __builtin_fmaf(y[2 * e + 1], y[2 * e + 1], ss); }
      if (r + 1 < ROUNDS) chain[tid & 255] = ss;
    }
    if (r + 1 < ROUNDS) __syncthreads();
  }
  if (round == ROUNDS - 1) {
    ss = wave_sum(ss);
    if (lane == 0) red[(tid >> 6) & 3] = ss;
  }
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = pack2(pk_lo(g[e]) * rbf(y[2 * e] * inv), pk_hi(g[e]) * rbf(y[2 * e + 1] * inv));
  if (Y8) mx32_store8(Y8, YS, slot, c * 8, o);       // fp8 matrix-core step (kernels_batch_mx.hip): the same bf16 values as MXFP8
  else *reinterpret_cast<u32x4*>(Y + xtile_off(slot, c * 8, (D + 31) >> 5)) = o;
}
void launch_resid_norm_b(const float* part, bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int D, float eps, const BatchState* bs,
                         int nslots, hipStream_t s, uint8_t* Y8, uint8_t* YS) {
  const int rounds = (D >> 3) >> 8;
  if (rounds == 1) hipLaunchKernelGGL((k_resid_norm_b<1>), dim3(nslots), dim3(256), 0, s, part, X, ldx, w, Y, D, eps, bs, Y8, YS);
  else if (rounds == 2) hipLaunchKernelGGL((k_resid_norm_b<2>), dim3(nslots), dim3(512), 0, s, part, X, ldx, w, Y, D, eps, bs, Y8, YS);
  else hipLaunchKernelGGL((k_resid_norm_b<4>), dim3(nslots), dim3(1024), 0, s, part, X, ldx, w, Y, D, eps, bs, Y8, YS);
}

