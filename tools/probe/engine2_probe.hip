// engine2_probe.hip — the persistent decode layer the round-2 VERDICT asked for, measured stand-alone before it is built into
// the product: ONE launch for L ds-7b-shaped layers, per CU 1 LDS-DMA loader wave + 3 v_dot2c consumer waves, the weights
// streamed `global_load_lds_dwordx4 ... nt` into a 128 KiB LDS ring that RUNS AHEAD across phase edges, phase hand-offs as
// 8-byte {tag, data} granules (one sc1 store each, no flag, no fence, no grid barrier) swept by one consumer wave per CU
// (guides/MI355X_MICROARCH.md price list: engine-vs-launches, allgather, prefetch-credit, ldsdma-fill, nt-weights).
// Round 1's probe (engine_probe.hip) rejected a DIFFERENT design: grid barriers (5.9 us each) + register prefetch that the
// release fence drained.
//
// Phases per layer (batch 1 decode, GEMV only; the arithmetic is a stand-in with the product's data flow):
//   qkv   12288 x 4096   x -> y; the first 4096 outputs ("q") go on
//   attn  stand-in for attention: every CU needs all of q, produces 16 values of the 4096-vector (one all-gather edge, no weights)
//   o      4096 x 4096
//   gu    22016 x 4096   rows interleaved (2u = gate, 2u + 1 = up) -> act[u] = bf16(g) * bf16(u)
//   down   4096 x 11008
// Variant A = one kernel per phase in a hipGraph (what the product does: 5 launches per layer).  Variant C = the engine.  Both
// compute the same function with the same fp32 summation order per row, so the final vectors must be BIT-IDENTICAL (checked).
//   hipcc --offload-arch=gfx950 -O3 -o engine2_probe engine2_probe.hip && ./engine2_probe [layers] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ float dot8(const u32x4& a, const u32x4& b, float c) {
  c = dot2(a[0], b[0], c); c = dot2(a[1], b[1], c); c = dot2(a[2], b[2], c); c = dot2(a[3], b[3], c);
  return c;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__host__ __device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ __forceinline__ float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

#ifndef PROBE_D            // -DPROBE_D=2048 -DPROBE_FF=5504: the ds-1.3b layer (8 / 8 / 8 / 8 / 22 KB edges, 101 MB of weights) — VERDICT r3 item 4
#define PROBE_D 4096
#define PROBE_FF 11008
#endif
constexpr int D = PROBE_D, FF = PROBE_FF;
constexpr int NPHASE = 5;                       // qkv, attn, o, gu, down
enum { P_QKV = 0, P_ATTN = 1, P_O = 2, P_GU = 3, P_DOWN = 4 };
constexpr float SCALE = 1.0f;

struct LayerW { const uint16_t* w[NPHASE]; };   // row-major [N][K] bf16 (attn: null)
__host__ __device__ inline int phase_N(int p) { return p == P_QKV ? 3 * D : p == P_O ? D : p == P_GU ? 2 * FF : p == P_DOWN ? D : D; }
__host__ __device__ inline int phase_K(int p) { return p == P_DOWN ? FF : D; }

// ------------------------------------------------------------------------------------------------ shared epilogue math
// outputs are bf16; every variant applies exactly this
__device__ __forceinline__ uint16_t epi_plain(float acc) { return f2bf(acc * SCALE); }
__device__ __forceinline__ uint16_t epi_attn(uint16_t q) { return f2bf(bf2f(q) * 0.5f + 0.25f); }
__device__ __forceinline__ uint16_t epi_act(uint16_t g, uint16_t u) { return f2bf(bf2f(g) * bf2f(u) * 0.5f); }

// ================================================================================================ variant A: kernel per phase
constexpr int A_THREADS = 512, A_WAVES = A_THREADS / 64;
// x for the phase from plain bf16 vectors: in [K] (down: act[u] is formed from the gu outputs t[2u], t[2u+1] while loading)
__global__ __launch_bounds__(A_THREADS) void k_phase(const uint16_t* W, const uint16_t* xin, uint16_t* y, int N, int K, int pair_in) {
  __shared__ __attribute__((aligned(16))) uint16_t xs[FF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (pair_in) { for (int u = threadIdx.x; u < K; u += A_THREADS) xs[u] = epi_act(xin[2 * u], xin[2 * u + 1]); }
  else { for (int c = threadIdx.x; c < K / 8; c += A_THREADS) reinterpret_cast<u32x4*>(xs)[c] = reinterpret_cast<const u32x4*>(xin)[c]; }
  __syncthreads();
  const int K8 = K / 8;
  const int Wtot = gridDim.x * A_WAVES;
  for (int row = wave * gridDim.x + blockIdx.x; row < N; row += Wtot) {
    const u32x4* wr = reinterpret_cast<const u32x4*>(W + (size_t)row * K);
    float acc = 0.f;
    u32x4 wv[22];
#pragma unroll
    for (int i = 0; i < 22; ++i) { const int c = lane + 64 * i; if (c < K8) wv[i] = __builtin_nontemporal_load(wr + c); }
#pragma unroll
    for (int i = 0; i < 22; ++i) { const int c = lane + 64 * i; if (c < K8) acc = dot8(wv[i], reinterpret_cast<const u32x4*>(xs)[c], acc); }
    acc = wave_sum(acc);
    if (lane == 0) y[row] = epi_plain(acc);
  }
}
__global__ void k_attn_standin(const uint16_t* q, uint16_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D) out[i] = epi_attn(q[i]);
}

// ================================================================================================ variant C: the engine
constexpr int PIECE = 1024;                     // bytes of one LDS-DMA instruction (64 lanes x 16 B)
constexpr int RING_PIECES = 128;                // 128 KiB ring
constexpr int RING_BYTES = RING_PIECES * PIECE;
constexpr int SLOT = 16;                        // pieces per fill (16 KiB)
constexpr int XBUF_BYTES = FF * 2;              // 22016 B: the gathered input vector, handed from the sweeping wave to the others
constexpr int NCONS = 3;
constexpr unsigned SPIN_LIMIT = 1u << 20;     // ~1 s: a legitimate wait is tens of microseconds

struct EngineArgs {
  LayerW layers[8]; int nlayers;      // by value (kernarg, scalar loads): a vector load of a weight pointer would make the loader wait vmcnt(0) at every phase
  u64* vec[NPHASE];        // granule vectors: vec[p] = OUTPUT of phase p ({tag, 2 x bf16}); vec[P_DOWN] is also layer 0's input
  unsigned epoch0;         // tag of layer l, phase p = epoch0 + l * NPHASE + p + 1 (layer 0's input carries epoch0)
  unsigned* err;           // != 0: a bounded spin gave up (code)
  int depth;               // fills (16 KiB) the loader keeps in flight: 1..3
  uint16_t* final_out;     // plain bf16 [D]: the last layer's down output (for the A == C check)
  long long* trace;        // optional [ncu][phases][2] wall-clock stamps (100 MHz) of consumer wave 0: input vector ready, last item published
  int sweepers;            // consumer waves that sweep an input vector: 3 (a third each) or 1 (wave 0 sweeps all of it)
  int thin;                // 1: the loader keeps ONE fill in flight while a wave of its CU is sweeping (guide row gather-pass)
  int mode;                // timing experiments (wrong results): 1 = no edges (nobody waits for an input vector), 2 = no dot products (loader alone)
};

// control words in LDS, behind the ring and the x buffer (byte offsets from the start of the dynamic LDS = LDS address 0: the kernel
// has no static __shared__).  They are read and written with ds_read_b32 / ds_write_b32 in inline asm: a `volatile` access through
// a generic pointer compiles to FLAT loads with `s_waitcnt vmcnt(0)`, which would drain the loader's LDS-DMA queue at every poll.
constexpr unsigned OFF_FILLED = RING_BYTES + XBUF_BYTES;      // pieces landed in the ring (absolute count, advanced a fill at a time)
constexpr unsigned OFF_WPOS = OFF_FILLED + 4;                 // [3] absolute piece index below which consumer w no longer needs the ring
constexpr unsigned OFF_XREADY = OFF_FILLED + 16;              // phases (absolute count) whose input vector is in xbuf
constexpr unsigned OFF_ABORT = OFF_FILLED + 20;
constexpr unsigned OFF_XPARTS = OFF_FILLED + 24;             // sweep parts written into xbuf (3 per phase, absolute count)
constexpr unsigned OFF_XCOPIED = OFF_FILLED + 28;
constexpr unsigned OFF_GATHER = OFF_FILLED + 32;             // consumer waves of this CU that are sweeping right now (the loader thins its stream)            // consumer waves that have taken a phase's x into registers (3 per phase)
constexpr unsigned CTRL_BYTES = 48;

__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_byte) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);      // wave-uniform by construction; the "s" constraint wants a proof
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// one fill = 16 consecutive pieces (16 KiB of the weight stream) into 16 consecutive ring pieces: two asm statements of 8 pieces, ~4
// issue slots per piece (a per-piece C loop with its bookkeeping costs ~30 instructions per piece — more than the 0.64 us
// landing cadence of a fill allows one wave).  No instruction offsets: an `offset:` on global_load_lds moves the LDS destination
// as well as the global source (LDS address = M0 + offset + 16 * lane; the first version of this probe added it on top of the M0
// increments and scattered three pieces out of four), so every piece gets its own address register pair and M0 value.
__device__ __forceinline__ void glds_fill8_nt(const unsigned char* g, unsigned lds_byte) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);
  const unsigned char *g1 = g + 1024, *g2 = g + 2048, *g3 = g + 3072, *g4 = g + 4096, *g5 = g + 5120, *g6 = g + 6144, *g7 = g + 7168;
#define P1(G) "global_load_lds_dwordx4 " G ", off nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t" P1("%1") P1("%2") P1("%3") P1("%4") P1("%5") P1("%6") P1("%7") P1("%8") "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5), "v"(g6), "v"(g7), "s"(dst) : "memory");
#undef P1
}
__device__ __forceinline__ void glds_fill16_nt(const unsigned char* g, unsigned lds_byte) {
  glds_fill8_nt(g, lds_byte);
  glds_fill8_nt(g + 8192, lds_byte + 8192u);
}
__device__ __forceinline__ unsigned lds_ld(unsigned off) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return v;
}
__device__ __forceinline__ void lds_st(unsigned off, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(off), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add(unsigned off, unsigned v) { asm volatile("ds_add_u32 %0, %1" :: "v"(off), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// items of phase p owned by CU `cu`: [first, first + count) of N / rows_per_item items; an item = 2 rows (gu: 4 rows) -> ONE granule
__device__ __forceinline__ int rows_per_item(int p) { return p == P_GU ? 4 : 2; }
__device__ __forceinline__ void cu_items(int p, int cu, int ncu, int& first, int& count) {
  const int total = (p == P_ATTN ? D / 2 : phase_N(p) / rows_per_item(p));   // attn: one granule (2 values) per item, no weights
  const int base = total / ncu, rem = total % ncu;
  count = base + (cu < rem ? 1 : 0);
  first = cu * base + (cu < rem ? cu : rem);
}
__device__ __forceinline__ unsigned item_pieces_x2(int p) {   // ring bytes of one item / 512 (so that down's 43 KiB stays integral)
  return p == P_ATTN ? 0u : (unsigned)(rows_per_item(p) * phase_K(p) * 2 / 512);
}

// sweep granule rows [row0, row1) (a row = 64 consecutive granules, one per lane) of `vec` until every tag == tag: EVERY load of
// the range is in flight in each pass (the first version polled 16 loads at a time, chunk after chunk: two to six dependent
// round trips after the last producer had published); lane l keeps granule 64 * (row0 + j) + l in val[j]
template <int MAXJ>
__device__ __forceinline__ bool sweep_rows(const u64* vec, int row0, int row1, int G, unsigned tag, unsigned (&val)[MAXJ], int lane) {
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int g = lane + 64 * (row0 + j);
      if (row0 + j < row1 && g < G) {
        const u64 x = __hip_atomic_load(vec + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        val[j] = (unsigned)x;
        ok = ok && (unsigned)(x >> 32) == tag;
      }
    }
    if (__all(ok)) return true;
    if (spins > SPIN_LIMIT || lds_ld(OFF_ABORT)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// one row pair from the ring: rows at ring byte offsets r0, r0 + K * 2 (mod ring); x in registers (chunk c = lane + 64 i)
template <int NI>
__device__ __forceinline__ void row_pair(const unsigned char* ring, unsigned r0, int K8, int lane, const u32x4 (&xr)[22], float& a0, float& a1) {
  const unsigned rowb = (unsigned)K8 * 16u;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    if (c < K8) {
      unsigned o0 = r0 + 16u * (unsigned)c; if (o0 >= (unsigned)RING_BYTES) o0 -= RING_BYTES;
      unsigned o1 = o0 + rowb; if (o1 >= (unsigned)RING_BYTES) o1 -= RING_BYTES;
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(ring + o0);
      const u32x4 w1 = *reinterpret_cast<const u32x4*>(ring + o1);
      s0 = dot8(w0, xr[i], s0);
      s1 = dot8(w1, xr[i], s1);
    }
  }
  a0 = wave_sum(s0); a1 = wave_sum(s1);
}

__global__ __launch_bounds__(256) void k_engine(EngineArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  unsigned char* xbuf = smem + RING_BYTES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cu = blockIdx.x, ncu = gridDim.x;
  if (threadIdx.x == 0) { for (unsigned o = 0; o < CTRL_BYTES; o += 4) lds_st(OFF_FILLED + o, 0u); lds_drain(); }
  __syncthreads();
  const int nph_total = a.nlayers * NPHASE;

  if (wave == NCONS) {
    // ------------------------------------------------------------------------------------------ loader wave
    unsigned issued = 0;                    // pieces issued (absolute)
    unsigned fills_pending = 0;             // fills issued and not yet announced
    const unsigned depth = (unsigned)a.depth;
    auto wait_space = [&](unsigned upto, int p) -> bool {      // ring pieces [.., upto) must be free behind the slowest consumer
      for (unsigned spins = 0;; ++spins) {
        const unsigned lo = min(min(lds_ld(OFF_WPOS), lds_ld(OFF_WPOS + 4)), lds_ld(OFF_WPOS + 8));
        if (upto <= lo + RING_PIECES) return true;
        if (fills_pending) {                // the ring is full: nothing more to issue, so let everything in flight land and say so
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          fills_pending = 0;
          lds_st(OFF_FILLED, issued);
        }
        if (spins > SPIN_LIMIT || lds_ld(OFF_ABORT)) { if (lane == 0) { lds_st(OFF_ABORT, 1u); atomicMax(a.err, 100u + (unsigned)p); } return false; }
        __builtin_amdgcn_s_sleep(1);
      }
    };
    for (int ph = 0; ph < nph_total; ++ph) {
      const int l = ph / NPHASE, p = ph % NPHASE;
      if (p == P_ATTN) continue;
      int first, count; cu_items(p, cu, ncu, first, count);
      const unsigned npieces = (unsigned)count * item_pieces_x2(p) / 2u;       // (count * bytes) / 1024: integral for these shapes
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.layers[l].w[p]) + (size_t)first * rows_per_item(p) * phase_K(p) * 2 + lane * 16;
      unsigned q = 0;
      for (; q + SLOT <= npieces; q += SLOT) {            // full fills: `issued` is a multiple of 16 here (phases are padded to fills)
        if (!wait_space(issued + SLOT, p)) return;
        glds_fill16_nt(src + (size_t)q * PIECE, (issued % RING_PIECES) * PIECE);
        issued += SLOT; ++fills_pending;
        if (a.thin && lds_ld(OFF_GATHER) != 0u) {          // a sweep is queued behind this CU's own LDS-DMA burst: one fill at a time
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          fills_pending = 0;
          lds_st(OFF_FILLED, issued);
        } else if (fills_pending >= depth) {       // keep `depth` fills in flight: the oldest has landed when (depth - 1) * 16 loads are outstanding
          if (depth == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (depth == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
          --fills_pending;
          lds_st(OFF_FILLED, issued - fills_pending * SLOT);
        }
      }
      if (q < npieces) {                    // a phase whose stream is not a whole number of fills (down: 344 pieces): piecewise tail,
        if (!wait_space(issued + SLOT, p)) return;              // drained at once, and the ring position padded up to the next fill
        for (unsigned j = 0; q + j < npieces; ++j) glds16_nt(src + (size_t)(q + j) * PIECE, ((issued + j) % RING_PIECES) * PIECE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issued += SLOT; fills_pending = 0;
        lds_st(OFF_FILLED, issued);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_st(OFF_FILLED, issued);
    return;
  }

  // -------------------------------------------------------------------------------------------- consumer waves
  u32x4 xr[22];                             // this wave's copy of the phase input: chunk c = lane + 64 i
  unsigned stream_x2 = 0;                   // ring bytes / 512 streamed by this CU before the current phase
  for (int ph = 0; ph < nph_total; ++ph) {
    const int l = ph / NPHASE, p = ph % NPHASE;
    const unsigned tag_out = a.epoch0 + (unsigned)ph + 1u;
    const int K = p == P_ATTN ? D : phase_K(p), K8 = K / 8;
    // ---- the input vector: OUTPUT of the previous phase (layer 0, phase 0: vec[P_DOWN] tagged epoch0)
    const int psrc = (p + NPHASE - 1) % NPHASE;
    const unsigned tag_in = a.epoch0 + (unsigned)ph;
    if (a.mode & 1) {
      // timing experiment: no edge — the phase starts on whatever xbuf holds
    } else {
      // every consumer wave sweeps a third of the vector (rows of 64 granules), parks it in xbuf, and all three meet on LDS counters
      const int G = K / 2, rows = (G + 63) >> 6;
      const int nsw = a.sweepers == 1 ? 1 : NCONS;
      const int row0 = wave < nsw ? rows * wave / nsw : 0, row1 = wave < nsw ? rows * (wave + 1) / nsw : 0;
      auto park = [&](auto& val, auto nj) {     // sweep this wave's rows, wait until xbuf is free, park the values there
        constexpr int NJ = decltype(nj)::value;
        if (a.thin && lane == 0) lds_add(OFF_GATHER, 1u);
        const bool ok = sweep_rows<NJ>(a.vec[psrc], row0, row1, G, tag_in, val, lane);
        if (a.thin && lane == 0) lds_add(OFF_GATHER, 0xffffffffu);
        if (!ok) { if (lane == 0) { lds_st(OFF_ABORT, 1u); atomicMax(a.err, 200u + (unsigned)p); } return false; }
        for (unsigned spins = 0;; ++spins) {     // xbuf still holds the previous phase's x until all three waves have it in registers
          if (lds_ld(OFF_XCOPIED) >= (unsigned)NCONS * (unsigned)ph) break;
          if (spins > SPIN_LIMIT || lds_ld(OFF_ABORT)) { if (lane == 0) atomicMax(a.err, 250u + (unsigned)p); return false; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int g = lane + 64 * (row0 + j); if (row0 + j < row1 && g < G) reinterpret_cast<unsigned*>(xbuf)[g] = val[j]; }
        lds_drain();                          // the wave's LDS stores execute in order; the count goes last
        if (lane == 0) lds_add(OFF_XPARTS, 1u);
        return true;
      };
      if (nsw == 1) {
        if (wave == 0) { unsigned val[86]; if (!park(val, std::integral_constant<int, 86>())) return; }
      } else {
        unsigned val[29]; if (!park(val, std::integral_constant<int, 29>())) return;
      }
      for (unsigned spins = 0;; ++spins) {
        if (lds_ld(OFF_XPARTS) >= (unsigned)nsw * ((unsigned)ph + 1u)) break;
        if (spins > SPIN_LIMIT || lds_ld(OFF_ABORT)) { if (lane == 0) atomicMax(a.err, 300u + (unsigned)p); return; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
#pragma unroll
    for (int i = 0; i < 22; ++i) { const int c = lane + 64 * i; if (c < K8) xr[i] = reinterpret_cast<const u32x4*>(xbuf)[c]; }
    if (p != P_ATTN) { lds_drain(); if (lane == 0) lds_add(OFF_XCOPIED, 1u); }       // (attn reads xbuf in its items: counted after them)
    if (a.trace && wave == 0 && lane == 0) a.trace[((size_t)cu * nph_total + ph) * 2] = (long long)wall_clock64();

    int first, count; cu_items(p, cu, ncu, first, count);
    if (p == P_ATTN) {
      // stand-in: item = granule gi of the 4096-vector, computed from q (all of q was needed to get here)
      for (int it = wave; it < count; it += NCONS) {
        const int gi = first + it;
        if (lane == 0) {
          const unsigned qq = reinterpret_cast<const unsigned*>(xbuf)[gi];
          const unsigned o = (unsigned)epi_attn((uint16_t)qq) | ((unsigned)epi_attn((uint16_t)(qq >> 16)) << 16);
          __hip_atomic_store(a.vec[p] + gi, ((u64)tag_out << 32) | o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      lds_drain(); if (lane == 0) lds_add(OFF_XCOPIED, 1u);
      if (a.trace && wave == 0 && lane == 0) a.trace[((size_t)cu * nph_total + ph) * 2 + 1] = (long long)wall_clock64();
      continue;
    }
    const unsigned ipx2 = item_pieces_x2(p);
    for (int it = wave; it < count; it += NCONS) {
      const unsigned b0x2 = stream_x2 + (unsigned)it * ipx2;              // item start, bytes / 512
      const unsigned end_piece = (b0x2 + ipx2 + 1u) / 2u;                 // pieces that must have landed
      for (unsigned spins = 0;; ++spins) {
        if (lds_ld(OFF_FILLED) >= end_piece) break;
        if (spins > SPIN_LIMIT || lds_ld(OFF_ABORT)) { if (lane == 0) { lds_st(OFF_ABORT, 1u); atomicMax(a.err, 400u + (unsigned)p); } return; }
        __builtin_amdgcn_s_sleep(1);
      }
      const unsigned r0 = (unsigned)(((u64)b0x2 * 512ull) % (u64)RING_BYTES);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (a.mode & 2) { /* timing experiment: the ring is released unread */ }
      else if (K8 == 512) row_pair<8>(ring, r0, K8, lane, xr, a0, a1); else row_pair<22>(ring, r0, K8, lane, xr, a0, a1);
      if (p == P_GU && !(a.mode & 2)) {
        unsigned r1 = r0 + 2u * (unsigned)K8 * 16u; if (r1 >= (unsigned)RING_BYTES) r1 -= RING_BYTES;
        row_pair<8>(ring, r1, K8, lane, xr, a2, a3);
      }
      // release the item's ring bytes: this wave's next item starts 3 items on (or in the next phase)
      if (lane == 0) {                               // (the sums above consumed every ring read of this item)
        unsigned next_piece = 0x7fffffffu;           // nothing more to read: the ring is the loader's
        if (it + NCONS < count) next_piece = (stream_x2 + (unsigned)(it + NCONS) * ipx2) / 2u;
        else {                                       // this wave's first item of the next weight phase (attn has none)
          int nph = ph + 1;
          if (nph < nph_total && nph % NPHASE == P_ATTN) ++nph;
          if (nph < nph_total) next_piece = ((stream_x2 + (unsigned)count * ipx2 + 2u * SLOT - 1u) / (2u * SLOT) * (2u * SLOT) + (unsigned)wave * item_pieces_x2(nph % NPHASE)) / 2u;
        }
        lds_st(OFF_WPOS + 4u * (unsigned)wave, next_piece);
        const int gi = first + it;
        unsigned o;
        if (p == P_GU) o = (unsigned)epi_act(epi_plain(a0), epi_plain(a1)) | ((unsigned)epi_act(epi_plain(a2), epi_plain(a3)) << 16);
        else o = (unsigned)epi_plain(a0) | ((unsigned)epi_plain(a1) << 16);
        __hip_atomic_store(a.vec[p] + gi, ((u64)tag_out << 32) | o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == P_DOWN && l == a.nlayers - 1) reinterpret_cast<unsigned*>(a.final_out)[gi] = o;
      }
    }
    if (a.trace && wave == 0 && lane == 0) a.trace[((size_t)cu * nph_total + ph) * 2 + 1] = (long long)wall_clock64();
    stream_x2 = (stream_x2 + (unsigned)count * ipx2 + 2u * SLOT - 1u) / (2u * SLOT) * (2u * SLOT);   // phases are padded to whole fills (loader)
  }
}

// layer 0's input as granules tagged `tag`
__global__ void k_publish_input(u64* vec, const uint16_t* x, unsigned tag) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < D / 2) vec[g] = ((u64)tag << 32) | (unsigned)x[2 * g] | ((unsigned)x[2 * g + 1] << 16);
}
__global__ void k_fill(uint16_t* w, size_t n, uint32_t seed, float amp) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 0x9E3779B1u + seed * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    w[i] = f2bf(((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * amp);
  }
}

int main(int argc, char** argv) {
  const int nl = argc > 1 ? atoi(argv[1]) : 8;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  hipStream_t s; CK(hipStreamCreate(&s));
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  std::vector<LayerW> host(nl);
  size_t bytes = 0;
  for (int l = 0; l < nl; ++l)
    for (int p = 0; p < NPHASE; ++p) {
      host[l].w[p] = nullptr;
      if (p == P_ATTN) continue;
      const size_t n = (size_t)phase_N(p) * phase_K(p);
      uint16_t* w; CK(hipMalloc(&w, n * 2));
      const float amp = p == P_DOWN ? 0.0165f : 0.027f;          // unit gain for uniform(-amp, amp) rows of K terms
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, s, w, n, (uint32_t)(l * 8 + p + 1), amp);
      host[l].w[p] = w; bytes += n * 2;
    }
  uint16_t* x0; CK(hipMalloc(&x0, D * 2));
  hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, s, x0, (size_t)D, 777u, 1.0f);
  // variant A buffers
  uint16_t *bq, *bt, *bo, *bg, *bx[2], *final_a;     // q (3D), attn out, o out, gu out (2FF), layer outputs (ping-pong)
  CK(hipMalloc(&bq, 3 * D * 2)); CK(hipMalloc(&bt, D * 2)); CK(hipMalloc(&bo, D * 2)); CK(hipMalloc(&bg, 2 * FF * 2));
  CK(hipMalloc(&bx[0], D * 2)); CK(hipMalloc(&bx[1], D * 2)); CK(hipMalloc(&final_a, D * 2));
  CK(hipStreamSynchronize(s));
  printf("CUs %d, %d layers, weights %.2f GB\n", cus, nl, bytes / 1e9);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto fn, const char* name) {
    fn(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-72s %8.1f us/pass  %6.2f us/layer  %5.2f TB/s\n", name, ms * 1e3, ms * 1e3 / nl, bytes / (ms * 1e-3) / 1e12);
    return ms;
  };
  // ---- A: graph of per-phase kernels (5 launches per layer, as the product)
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int grid = cus * bpc;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    const uint16_t* xin = x0;
    for (int l = 0; l < nl; ++l) {
      hipLaunchKernelGGL(k_phase, dim3(grid), dim3(A_THREADS), 0, s, host[l].w[P_QKV], xin, bq, 3 * D, D, 0);
      hipLaunchKernelGGL(k_attn_standin, dim3(D / 256), dim3(256), 0, s, bq, bt);
      hipLaunchKernelGGL(k_phase, dim3(grid), dim3(A_THREADS), 0, s, host[l].w[P_O], bt, bo, D, D, 0);
      hipLaunchKernelGGL(k_phase, dim3(grid), dim3(A_THREADS), 0, s, host[l].w[P_GU], bo, bg, 2 * FF, D, 0);
      uint16_t* out = l == nl - 1 ? final_a : bx[l & 1];
      hipLaunchKernelGGL(k_phase, dim3(grid), dim3(A_THREADS), 0, s, host[l].w[P_DOWN], bg, out, D, FF, 1);
      xin = out;
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    char nm[96]; snprintf(nm, sizeof nm, "A graph of per-phase kernels, %d blocks x 8 waves", grid);
    time_it([&] { CK(hipGraphLaunch(ge, s)); }, nm);
  }
  std::vector<uint16_t> ref(D); CK(hipMemcpy(ref.data(), final_a, D * 2, hipMemcpyDeviceToHost));
  // ---- C: the engine
  EngineArgs ea{};
  if (nl > 8) { printf("at most 8 layers\n"); return 1; }
  for (int l = 0; l < nl; ++l) ea.layers[l] = host[l];
  ea.nlayers = nl;
  for (int p = 0; p < NPHASE; ++p) { const size_t g = (size_t)(p == P_QKV ? 3 * D / 2 : p == P_GU ? FF / 2 : D / 2); CK(hipMalloc(&ea.vec[p], g * 8)); CK(hipMemset(ea.vec[p], 0, g * 8)); }
  CK(hipMalloc(&ea.err, 4)); CK(hipMemset(ea.err, 0, 4));
  CK(hipMalloc(&ea.final_out, D * 2)); CK(hipMemset(ea.final_out, 0, D * 2));
  const size_t lds = (size_t)RING_BYTES + XBUF_BYTES + CTRL_BYTES;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_engine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_engine, 256, lds));
  printf("engine: %zu B LDS per block, occupancy %d block/CU, grid %d\n", lds, occ, cus);
  if (occ < 1) { printf("engine does not fit\n"); return 1; }
  unsigned epoch = 1;
  const int runs[][4] = {{2, 0, 3, 0}, {2, 0, 3, 1}, {2, 0, 1, 0}, {2, 0, 1, 1}, {3, 0, 1, 1}, {2, 1, 3, 0}, {2, 3, 3, 0}};      // (fills in flight, mode, sweeping waves, loader thinning)
  for (const auto& run : runs) {
    const int depth = run[0];
    ea.depth = depth; ea.mode = run[1]; ea.sweepers = run[2]; ea.thin = run[3];
    char nm[128]; snprintf(nm, sizeof nm, "C engine, %d fills, %d sweeper%s, thinning %s%s", depth, run[2], run[2] == 1 ? "" : "s", run[3] ? "on" : "off",
                           run[1] == 0 ? "" : run[1] == 1 ? " [no edges]" : " [no edges, no dots: loader alone]");
    auto go = [&] {
      ea.epoch0 = epoch;
      hipLaunchKernelGGL(k_publish_input, dim3(D / 2 / 256), dim3(256), 0, s, ea.vec[P_DOWN], x0, epoch);
      hipLaunchKernelGGL(k_engine, dim3(cus), dim3(256), lds, s, ea);
      epoch += (unsigned)(nl * NPHASE + 2);
    };
    time_it(go, nm);
    unsigned err = 0; CK(hipMemcpy(&err, ea.err, 4, hipMemcpyDeviceToHost));
    std::vector<uint16_t> got(D); CK(hipMemcpy(got.data(), ea.final_out, D * 2, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < D; ++i) bad += got[i] != ref[i];
    printf("   engine err code %u; final vector vs variant A: %d of %d elements differ (first values %04x %04x %04x vs %04x %04x %04x)\n",
           err, bad, D, got[0], got[1], got[2], ref[0], ref[1], ref[2]);
    if (err) { CK(hipMemset(ea.err, 0, 4)); }
  }
  // ---- where an edge's time goes: one traced launch (consumer wave 0 of every CU stamps "input ready" and "my last item published")
  {
    const int nph = nl * NPHASE;
    long long* dtrace; CK(hipMalloc(&dtrace, (size_t)cus * nph * 2 * 8)); CK(hipMemset(dtrace, 0, (size_t)cus * nph * 2 * 8));
    ea.depth = 2; ea.mode = 0; ea.sweepers = 3; ea.thin = 0; ea.trace = dtrace; ea.epoch0 = epoch;
    hipLaunchKernelGGL(k_publish_input, dim3(D / 2 / 256), dim3(256), 0, s, ea.vec[P_DOWN], x0, epoch);
    hipLaunchKernelGGL(k_engine, dim3(cus), dim3(256), lds, s, ea);
    CK(hipStreamSynchronize(s));
    std::vector<long long> tr((size_t)cus * nph * 2); CK(hipMemcpy(tr.data(), dtrace, tr.size() * 8, hipMemcpyDeviceToHost));
    const char* names[NPHASE] = {"qkv", "attn", "o", "gu", "down"};
    printf("edge anatomy (one launch, wave 0 of each CU; us, 100 MHz wall clock): per phase, over CUs: ready = input vector in registers, done = own last item published\n");
    long long t00 = tr[0]; for (int c = 0; c < cus; ++c) t00 = tr[(size_t)c * nph * 2] < t00 ? tr[(size_t)c * nph * 2] : t00;
    double sum_edge = 0, sum_spread = 0, sum_body = 0; int n_edge = 0;
    for (int ph = 0; ph < nph; ++ph) {
      long long rmin = 1LL << 62, rmax = 0, dmin = 1LL << 62, dmax = 0; double ravg = 0, davg = 0;
      for (int c = 0; c < cus; ++c) {
        const long long r = tr[((size_t)c * nph + ph) * 2], d = tr[((size_t)c * nph + ph) * 2 + 1];
        rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax; dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; ravg += r; davg += d;
      }
      ravg /= cus; davg /= cus;
      if (ph >= NPHASE && ph < 2 * NPHASE)      // print layer 1 in full (layer 0 carries the cold start)
        printf("  layer 1 %-5s ready min/avg/max %7.2f %7.2f %7.2f   done min/avg/max %7.2f %7.2f %7.2f   body(avg) %6.2f  done spread %5.2f\n", names[ph % NPHASE],
               (rmin - t00) / 100.0, (ravg - t00) / 100.0, (rmax - t00) / 100.0, (dmin - t00) / 100.0, (davg - t00) / 100.0, (dmax - t00) / 100.0,
               (davg - ravg) / 100.0, (dmax - dmin) / 100.0);
      if (ph >= NPHASE) { sum_spread += (dmax - dmin) / 100.0; sum_body += (davg - ravg) / 100.0; }
      if (ph + 1 < nph && ph >= NPHASE) {       // edge: slowest producer's publish -> average consumer has the vector
        double nr = 0; for (int c = 0; c < cus; ++c) nr += tr[((size_t)c * nph + ph + 1) * 2]; nr /= cus;
        sum_edge += (nr - dmax) / 100.0; ++n_edge;
      }
    }
    printf("  layers 1..%d, per layer: sum of phase bodies (avg CU) %.1f us, sum over phases of the finish spread across CUs %.1f us, sum of edges (slowest publish -> avg ready) %.1f us\n",
           nl - 1, sum_body / (nl - 1), sum_spread / (nl - 1), sum_edge / n_edge * NPHASE);
  }
  return 0;
}
