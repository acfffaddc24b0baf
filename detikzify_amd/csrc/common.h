// common.h — device helpers shared by the gfx950 kernels of libdtk_hip.so.
// wave = 64 lanes everywhere (CDNA4); bf16 values travel as raw uint16 bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DTK_WAVE 64

// Raising a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE property of the function, so the
// "already done" mark of a launcher is a bit per device, not a process-wide bool: a second dtk_ctx on another GPU of the same process
// sets it again.  A refused attribute is reported once on stderr and remembered (dtk_lds_attr_failed: the step launchers fail on it).
static inline bool dtk_lds_attr_todo(unsigned long long& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;       // unknown device: set it every time
  const unsigned long long bit = 1ull << dev;
  if (done & bit) return false;
  done |= bit;
  return true;
}
int& dtk_lds_attr_error(int device);                                                // dtk_api.hip: the first hipError_t a launcher's attribute call returned ON THAT DEVICE (0 = none):
                                                                                    // a refusal on one GPU does not fail the contexts of another
static inline void dtk_lds_attr(hipError_t e, const char* file, int line) {
  if (e == hipSuccess) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!dtk_lds_attr_error(dev)) {
    fprintf(stderr, "libdtk_hip: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d at %s:%d: %s\n", dev, file, line, hipGetErrorString(e));
    dtk_lds_attr_error(dev) = (int)e;
  }
}
#define DTK_LDS_ATTR(call) dtk_lds_attr((call), __FILE__, __LINE__)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even (v_cvt_pk_bf16_f32), same as torch .to(bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
// round an fp32 value through bf16 (every place HF materialises a bf16 tensor)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float pk_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float pk_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
// c + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs, fp32 accumulate (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                         __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ float dot8(const u32x4& a, const u32x4& b, float c) {
  c = dot2(a[0], b[0], c);
  c = dot2(a[1], b[1], c);
  c = dot2(a[2], b[2], c);
  c = dot2(a[3], b[3], c);
  return c;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// streamed-once weights: non-temporal 16-byte load
__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

// Decode-step state that lives in device memory so a captured hipGraph can be
// replayed unchanged: kernels read the position / token from here.
struct DecState {
  int32_t pos;       // position index of the token being forwarded this step
  int32_t next_pos;  // number of tokens that have KV after this step
  int32_t token;     // token sampled this step (input of the forward)
  uint32_t draw;     // number of tokens sampled since dtk_set_sampling
  int32_t force_plus1;   // != 0: this step forwards token force_plus1 - 1 instead of sampling one (dtk_resume_slot); cleared by the sampler
  int32_t pad[3];
};

// Batched decode (kernels_batch_decode.hip): which of the (up to 64) slots take part in this step.
#ifndef DTK_MAX_BATCH
#define DTK_MAX_BATCH 64   // == include/dtk.h
#endif
#define DTK_PFX_GRID 16    // group rows of k_attn_prefix_g's grid: row y takes groups y, y + 16, ...
#define DTK_PFX_GROUPS 64  // prefix groups a step can hand to k_attn_prefix_g: the worst case of 64 slots that share nothing (64 singleton groups), so that
                           // whether a slot's prefix goes through the matrix cores never depends on how many OTHER prefixes the step holds (was 16)
// One shared prefix scored on the matrix cores: the active slots that read the first `len` rows of their cache from slot `src`
// (forks of one image: dtk_kv_fork), at most 16 of them = the columns of one MFMA tile.  A source with more forks has several groups.
struct PfxGroup {
  int32_t src;       // the slot whose cache holds the rows
  int32_t len;       // keys [0, len) are the group's shared prefix
  int32_t n;         // members (1..16)
  int32_t pad;
  int32_t slot[16];  // member slots (entries >= n repeat slot[0])
};
struct BatchState {
  int32_t active[DTK_MAX_BATCH];
  int32_t step;      // global step counter (token ring index)
  // Shared prefixes on the matrix cores (k_attn_prefix_g): the host groups the step's active slots by (share_src, share_len) — a
  // property of the slot alone, so whether and how a slot's prefix is scored never depends on which other slots decode with it —
  // into groups of <= 16 slots (at most DTK_PFX_GROUPS = 64: every slot always finds a group); group_plus1[slot] = the slot's group + 1 (0: none, the slot's whole context
  // is walked by k_attn_tail_b).  n_groups == 0: no prefix kernel work this step.
  int32_t n_groups;
  int32_t pad[14];
  // forked slots (dtk_kv_fork) hold a bit-identical copy of their source's first share_len keys: attention reads those
  // rows from the SOURCE slot's cache instead (-1 = none), so the 32 rollouts of one image stream the 243-key image
  // prefix from HBM once per layer and hit the XCD's L2 afterwards
  int32_t share_src[DTK_MAX_BATCH];
  int32_t share_len[DTK_MAX_BATCH];
  int32_t group_plus1[DTK_MAX_BATCH];
  int32_t pfx_len_of[DTK_MAX_BATCH];   // = groups[group_plus1[slot] - 1].len, or 0: what k_attn_tail_b needs, without the dependent lookup
  PfxGroup groups[DTK_PFX_GROUPS];
};

// Batched decode keeps the GEMV INPUT vectors (normalised x, attention output, SwiGLU activation) of the slots in
// the MFMA B-operand fragment order: tile (slot/16, k/32) is 1 KiB, lane = ((k%32)/8)*16 + slot%16 holds 8
// consecutive k.  A wave's x load is then 1 KiB contiguous (8 full cache lines) instead of 16 rows x 64 B
// (16 half-used lines): measured gate/up at 32 slots 48.3 -> 35.2 us.
__device__ __forceinline__ size_t xtile_off(int slot, int k, int nsteps) {
  return ((size_t)((slot >> 4) * nsteps + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (slot & 15)) * 8 + (k & 7);
}

struct SamplingDev {
  int32_t do_sample;
  float temperature;
  float top_p;
  int32_t top_k;
  uint64_t seed;
  int32_t n_bad;
  int32_t bad_ids[8];
  int32_t n_begin;
  int32_t begin_ids[8];
  int32_t n_always;
  int32_t always_ids[8];
};

// scratch of the multi-block sampler (kernels_sample_mb.hip), one per sequence / slot
#define DTK_SAMPLE_MB_MAX_SLICES 32      // 32 x 8192 = vocabularies up to 262 144
struct SampleMB {
  float bmax[DTK_SAMPLE_MB_MAX_SLICES];
  int32_t barg[DTK_SAMPLE_MB_MAX_SLICES];
  unsigned long long bkept[DTK_SAMPLE_MB_MAX_SLICES];
  unsigned long long hmass[4][256];      // [radix level][bin]: integer probability mass
  unsigned int hcnt[4][256];
  unsigned long long above[4];           // mass strictly above the chosen bin, per level
  unsigned int bin[4];
  unsigned long long total;
  unsigned int thr;
  // DecState.draw / .force_plus1 as they were BEFORE this step's draw kernel (written by block 0 of k_smb_max): k_smb_draw's
  // blocks read only these — its last block rewrites DecState while blocks of the same slot may not have started yet
  unsigned int draw_snap;
  int forced_snap, pad;
};

// Top-p radix level: among the 256 (mass, count) bins of one level pick the LOWEST non-empty bin whose strictly-above
// mass (above_in + mass of all higher bins) is still < pq; if even the highest non-empty bin fails, pick that one.  This
// is exactly the serial scan `for b = 255..0` of the samplers (whose dependent LDS reads cost ~10 us per level), done by
// threads 0..255 with a wave suffix scan + ballots.  EVERY thread of the block must call it (it contains barriers);
// s_w: 8 x u64 of LDS scratch, s_i: 8 ints.  Results (bin, strictly-above mass of that bin) are returned to all threads.
__device__ __forceinline__ void bin_select_mass(const unsigned long long* s_m, const unsigned int* s_c, unsigned long long pq,
                                                unsigned long long above_in, unsigned long long* s_w, int* s_i,
                                                unsigned int& bin_out, unsigned long long& above_out) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long m = 0, incl = 0; unsigned int c = 0;
  if (tid < 256) {
    m = s_m[tid]; c = s_c[tid];
    incl = m;                                   // inclusive suffix sum over the wave's 64 bins (lanes >= lane)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long v = __shfl_down(incl, off, 64);
      if (lane + off < 64) incl += v;
    }
    if (lane == 0) s_w[w] = incl;
  }
  __syncthreads();
  unsigned long long above = 0;
  bool ok = false;
  if (tid < 256) {
    unsigned long long hi = 0;
    for (int k = w + 1; k < 4; ++k) hi += s_w[k];
    above = above_in + hi + (incl - m);          // mass strictly above bin tid
    ok = c > 0 && above < pq;
    const unsigned long long mk = __ballot(ok), ne = __ballot(c > 0);
    if (lane == 0) {
      s_i[w] = mk ? (__ffsll((long long)mk) - 1) + 64 * w : -1;             // lowest qualifying bin of this wave
      s_i[4 + w] = ne ? (63 - __clzll((long long)ne)) + 64 * w : -1;        // highest non-empty bin of this wave
    }
  }
  __syncthreads();
  int sel = -1;
  for (int k = 0; k < 4 && sel < 0; ++k) sel = s_i[k];
  if (sel < 0) for (int k = 3; k >= 0 && sel < 0; --k) sel = s_i[4 + k];
  if (sel < 0) sel = 0;
  __syncthreads();
  if (tid == sel) s_w[4] = above;
  __syncthreads();
  bin_out = (unsigned)sel; above_out = s_w[4];
  __syncthreads();
}

// splitmix64: the counter-based RNG shared with oracle/sampling.py
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
