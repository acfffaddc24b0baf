// kernels_sample_mb.hip — multi-block sampler for large vocabularies (V > 32768: the v2 models, 128 256 tokens).
//
// The single-block samplers (kernels_decode.hip) walk the whole vocabulary with ONE CU: 126 logits per thread and
// heavily contended LDS histograms cost 570 us per token on LLaMA-3.1's vocabulary (16 % of the v2 decode step).  Here
// the vocabulary is cut into slices of 8192 (1024 threads x 8 CONSECUTIVE logits, register resident) and the sampler
// becomes a chain of 7 tiny kernels; a kernel boundary inside the captured graph (1.6 us, profiles/r01_launch_probe.txt)
// is the cheapest device-wide hand-off on this part (a software grid barrier costs 6 us, r01_engine_probe.txt):
//   P0  per-slice masked max / first arg-max                                     -> bmax, barg   (+ zero the histograms)
//   P1..P4  radix level 3..0 of the top-p threshold key: per-wave private LDS histograms of (integer mass, count) ->
//           block histogram -> global integer atomics (order independent: exact); every block re-derives the
//           previous level's bin from the global histogram, block 0 publishes it
//   P5  threshold = level-0 bin; kept mass per slice                              -> bkept
//   P6  target = floor(kept * r / 2^32); the slice / thread / element that owns it in index order -> token; advance
//       DecState; gather the token's embedding (first op of the next forward)
// Integer semantics identical to k_sample / k_sample_fast / oracle/sampling.py (q = floor(exp(z - zmax) * 2^31), kept set
// { key >= thr }, inverse-CDF draw in index order with the splitmix64 counter RNG): same tokens, bit for bit.
// Supports greedy, temperature, top-p and the suppression lists; top-k > 0 stays on the single-block kernel (the API
// re-captures the graph when a sampling configuration needs the other kind).
#include "kernels.h"

#define MB_THREADS 1024
#define MB_PER 8
#define MB_SLICE (MB_THREADS * MB_PER)

__device__ __forceinline__ uint32_t mb_fkey(float f) {   // order-preserving float -> uint key (as kernels_decode.hip fkey)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct MbElems {
  float z[MB_PER];
  uint32_t key[MB_PER];
  uint32_t q[MB_PER];
};

// per-(sequence) view of the arguments in batched mode
__device__ __forceinline__ bool mb_bind(SampleArgs& a, int slot) {
  if (a.bs) {
    if (!a.bs->active[slot]) return false;
    a.logits += (size_t)slot * a.logits_stride;
    a.sp += slot;
    a.st += slot;
    a.x += (size_t)slot * a.d;
    a.tok_ring += (size_t)((unsigned)a.bs->step % (unsigned)a.ring) * DTK_MAX_BATCH + slot;
    a.ring = 1;
    a.step_override = -1;
    a.mb += slot;
  }
  return true;
}

// this thread's 8 consecutive logits: scaled, suppressed ids -> -inf, keys; out-of-range -> -inf
__device__ __forceinline__ void mb_load(const SampleArgs& a, const SamplingDev* sp, bool first, float invT, int base, MbElems& e) {
  const int V = a.V;
  if (base + MB_PER <= V && ((reinterpret_cast<uintptr_t>(a.logits + base) & 15) == 0)) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(a.logits + base);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.logits + base + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { e.z[i] = v0[i] * invT; e.z[4 + i] = v1[i] * invT; }
  } else {
#pragma unroll
    for (int i = 0; i < MB_PER; ++i) e.z[i] = (base + i < V) ? a.logits[base + i] * invT : -INFINITY;
  }
  // suppression lists: which of the (<= 24) ids fall into this thread's 8 elements
  auto ban = [&](int id) {
    const int o = id - base;
#pragma unroll
    for (int i = 0; i < MB_PER; ++i) if (o == i) e.z[i] = -INFINITY;
  };
  for (int j = 0; j < sp->n_bad; ++j) ban(sp->bad_ids[j]);
  for (int j = 0; j < sp->n_always; ++j) ban(sp->always_ids[j]);
  if (first) for (int j = 0; j < sp->n_begin; ++j) ban(sp->begin_ids[j]);
#pragma unroll
  for (int i = 0; i < MB_PER; ++i) e.key[i] = mb_fkey(e.z[i]);
}

__device__ __forceinline__ void mb_mass(MbElems& e, float zmax) {
#pragma unroll
  for (int i = 0; i < MB_PER; ++i) e.q[i] = (uint32_t)(unsigned long long)((double)expf(e.z[i] - zmax) * 2147483648.0);
}

struct MbCommon {
  SamplingDev sp;
  float zmax;
  int argmax;
};

// sampling config -> LDS, global max / first arg-max from the per-slice results of P0
__device__ __forceinline__ void mb_common(const SampleArgs& a, int nblk, MbCommon* cm) {
  const int tid = threadIdx.x;
  if (tid < (int)(sizeof(SamplingDev) / 4)) reinterpret_cast<uint32_t*>(&cm->sp)[tid] = reinterpret_cast<const uint32_t*>(a.sp)[tid];
  if (tid < 64) {   // <= 32 slices: one wave reduces (max, first arg-max)
    float b = tid < nblk ? a.mb->bmax[tid] : -INFINITY; int bi = tid < nblk ? a.mb->barg[tid] : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(b, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ob > b || (ob == b && oi < bi)) { b = ob; bi = oi; }
    }
    if (tid == 0) { cm->zmax = b; cm->argmax = bi; }
  }
  __syncthreads();
}

// ---- P0
__global__ __launch_bounds__(MB_THREADS) void k_smb_max(SampleArgs a) {
  if (!mb_bind(a, blockIdx.y)) return;
  __shared__ SamplingDev s_sp;
  __shared__ float s_f[16];
  __shared__ int s_i[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  if (tid < (int)(sizeof(SamplingDev) / 4)) reinterpret_cast<uint32_t*>(&s_sp)[tid] = reinterpret_cast<const uint32_t*>(a.sp)[tid];
  __syncthreads();
  const uint32_t draw = (a.step_override >= 0) ? (uint32_t)a.step_override : a.st->draw;
  const float invT = s_sp.do_sample ? 1.f / s_sp.temperature : 1.f;
  MbElems e;
  const int base = blk * MB_SLICE + tid * MB_PER;
  mb_load(a, &s_sp, draw == 0, invT, base, e);
  float best = -INFINITY; int besti = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < MB_PER; ++i)
    if (base + i < a.V && (e.z[i] > best || (e.z[i] == best && base + i < besti))) { best = e.z[i]; besti = base + i; }
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
    a.mb->bmax[blk] = b; a.mb->barg[blk] = bi;
    if (blk == 0) { a.mb->draw_snap = draw; a.mb->forced_snap = a.advance ? a.st->force_plus1 : 0; }
  }
  // the histograms of this step start from zero (P1 is a later kernel)
  if (blk == 0) {
    unsigned long long* hm = &a.mb->hmass[0][0];
    unsigned int* hc = &a.mb->hcnt[0][0];
    for (int i = tid; i < 4 * 256; i += MB_THREADS) { hm[i] = 0ull; hc[i] = 0u; }
  }
}

// bin of radix level `lv` from its global histogram (bin_select_mass, common.h); called by every block, published by block 0
__device__ __forceinline__ void mb_select(const SampleArgs& a, int lv, unsigned long long pq, unsigned long long above_in,
                                          unsigned long long* s_m, unsigned int* s_c, unsigned long long* s_w, int* s_i,
                                          unsigned int& bin_out, unsigned long long& above_out) {
  const int tid = threadIdx.x;
  if (tid < 256) { s_m[tid] = a.mb->hmass[lv][tid]; s_c[tid] = a.mb->hcnt[lv][tid]; }
  __syncthreads();
  bin_select_mass(s_m, s_c, pq, above_in, s_w, s_i, bin_out, above_out);
}

// ---- P1..P4: LV = 3, 2, 1, 0
template <int LV>
__global__ __launch_bounds__(MB_THREADS) void k_smb_hist(SampleArgs a, int nblk) {
  if (!mb_bind(a, blockIdx.y)) return;
  __shared__ MbCommon cm;
  __shared__ unsigned long long w_m[16][256];   // per-wave private histograms: conflicts only inside a wave
  __shared__ unsigned int w_c[16][256];
  __shared__ unsigned long long s_m[256];
  __shared__ unsigned int s_c[256];
  __shared__ unsigned long long s_w[8];
  __shared__ int s_i[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  mb_common(a, nblk, &cm);
  if (!cm.sp.do_sample || !(cm.sp.top_p < 1.0f)) return;   // greedy / no nucleus: nothing to select (block-uniform)
  const uint32_t draw = (a.step_override >= 0) ? (uint32_t)a.step_override : a.st->draw;
  // selection of the previous level (LV + 1), then this level's prefix
  uint32_t prefix = 0;
  if (LV < 3) {
    // total mass = sum of the level-3 histogram
    unsigned long long total = 0, above = 0;
    if (LV + 1 == 3) {     // total mass = sum of the level-3 bins: the same suffix machinery with pq = 0 (nothing qualifies)
      unsigned long long t = 0;
      if (tid < 256) {
        t = a.mb->hmass[3][tid];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) s_w[wave] = t;
      }
      __syncthreads();
      total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
      __syncthreads();
      if (blk == 0 && tid == 0) a.mb->total = total;
    } else {
      total = a.mb->total; above = a.mb->above[LV + 2];
      for (int l = 3; l > LV + 1; --l) prefix |= a.mb->bin[l] << (l * 8);
    }
    const unsigned long long pq = (unsigned long long)((double)cm.sp.top_p * (double)total);
    unsigned int bin; unsigned long long ab;
    mb_select(a, LV + 1, pq, above, s_m, s_c, s_w, s_i, bin, ab);
    prefix |= bin << ((LV + 1) * 8);
    if (blk == 0 && tid == 0) { a.mb->bin[LV + 1] = bin; a.mb->above[LV + 1] = ab; }
  }
  for (int i = tid; i < 16 * 256; i += MB_THREADS) { (&w_m[0][0])[i] = 0ull; (&w_c[0][0])[i] = 0u; }
  __syncthreads();
  const float invT = 1.f / cm.sp.temperature;
  MbElems e;
  const int base = blk * MB_SLICE + tid * MB_PER;
  mb_load(a, &cm.sp, draw == 0, invT, base, e);
  mb_mass(e, cm.zmax);
  constexpr int shift = LV * 8;
  // run-length aggregation over the thread's 8 consecutive elements (neighbours mostly share the digit), then atomics
  // on the wave's private histogram
  unsigned long long run_q = 0; unsigned int run_c = 0; int run_d = -1;
#pragma unroll
  for (int i = 0; i < MB_PER; ++i) {
    const bool in = base + i < a.V;
    const bool match = in && ((LV == 3) || ((e.key[i] >> (shift + 8)) == (prefix >> (shift + 8))));
    const int d = match ? (int)((e.key[i] >> shift) & 255) : -1;
    if (d != run_d) {
      if (run_d >= 0) { atomicAdd(&w_m[wave][run_d], run_q); atomicAdd(&w_c[wave][run_d], run_c); }
      run_d = d; run_q = 0; run_c = 0;
    }
    if (d >= 0) { run_q += e.q[i]; run_c += 1; }
  }
  if (run_d >= 0) { atomicAdd(&w_m[wave][run_d], run_q); atomicAdd(&w_c[wave][run_d], run_c); }
  __syncthreads();
  if (tid < 256) {
    unsigned long long m = 0; unsigned int c = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { m += w_m[w][tid]; c += w_c[w][tid]; }
    if (c) { atomicAdd(&a.mb->hmass[LV][tid], m); atomicAdd(&a.mb->hcnt[LV][tid], c); }
  }
  (void)lane;
}

// ---- P5: threshold + kept mass per slice
__global__ __launch_bounds__(MB_THREADS) void k_smb_kept(SampleArgs a, int nblk) {
  if (!mb_bind(a, blockIdx.y)) return;
  __shared__ MbCommon cm;
  __shared__ unsigned long long s_m[256];
  __shared__ unsigned int s_c[256];
  __shared__ unsigned long long s_w[8];
  __shared__ int s_i[8];
  __shared__ unsigned long long s_q[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  mb_common(a, nblk, &cm);
  if (!cm.sp.do_sample) return;
  const uint32_t draw = (a.step_override >= 0) ? (uint32_t)a.step_override : a.st->draw;
  uint32_t thr = 0;
  if (cm.sp.top_p < 1.0f) {
    const unsigned long long total = a.mb->total;
    const unsigned long long pq = (unsigned long long)((double)cm.sp.top_p * (double)total);
    uint32_t prefix = 0;
    for (int l = 3; l > 0; --l) prefix |= a.mb->bin[l] << (l * 8);
    unsigned int bin; unsigned long long ab;
    mb_select(a, 0, pq, a.mb->above[1], s_m, s_c, s_w, s_i, bin, ab);
    thr = prefix | bin;
  }
  if (blk == 0 && tid == 0) a.mb->thr = thr;
  const float invT = 1.f / cm.sp.temperature;
  MbElems e;
  const int base = blk * MB_SLICE + tid * MB_PER;
  mb_load(a, &cm.sp, draw == 0, invT, base, e);
  mb_mass(e, cm.zmax);
  unsigned long long mine = 0;
#pragma unroll
  for (int i = 0; i < MB_PER; ++i) if (base + i < a.V && e.key[i] >= thr) mine += e.q[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if (lane == 0) s_q[wave] = mine;
  __syncthreads();
  if (tid == 0) { unsigned long long t = 0; for (int w = 0; w < 16; ++w) t += s_q[w]; a.mb->bkept[blk] = t; }
}

// ---- P6: draw (or arg-max), token ring, DecState, embedding gather
__global__ __launch_bounds__(MB_THREADS) void k_smb_draw(SampleArgs a, int nblk) {
  if (!mb_bind(a, blockIdx.y)) return;
  __shared__ MbCommon cm;
  __shared__ unsigned long long scan[MB_THREADS];
  __shared__ int s_token;
  const int tid = threadIdx.x, blk = blockIdx.x;
  mb_common(a, nblk, &cm);
  // the snapshot of k_smb_max, NOT DecState: the block that owns the token advances DecState (draw, force_plus1) at its end,
  // and with more blocks than the chip holds at once (v2-8b: 16 slices x 64 slots) a block of the same slot can start after
  // that — it would see the next step's draw index / a cleared force flag, pick an owner of its own and advance a second time
  const uint32_t draw = a.mb->draw_snap;
  const int forced = a.mb->forced_snap;
  const bool sampling = cm.sp.do_sample != 0;
  if (tid == 0) s_token = -1;
  __syncthreads();
  unsigned long long kept = 0;
  bool owner = false;
  if (!sampling) {
    owner = blk == 0;
    if (owner && tid == 0) s_token = cm.argmax;
    if (a.probs_out) {
      const int base = blk * MB_SLICE + tid * MB_PER;
      for (int i = 0; i < MB_PER; ++i) if (base + i < a.V) a.probs_out[base + i] = (base + i == cm.argmax) ? 1.f : 0.f;
    }
  } else {
    unsigned long long before = 0;
    for (int k = 0; k < nblk; ++k) { const unsigned long long v = a.mb->bkept[k]; if (k < blk) before += v; kept += v; }
    const uint64_t r = splitmix64(cm.sp.seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(draw + 1))) >> 32;
    const unsigned long long target = __umul64hi(kept, r << 32);  // floor(kept * r / 2^32)
    const unsigned long long mine_blk = a.mb->bkept[blk];
    owner = mine_blk > 0 && target >= before && target < before + mine_blk;     // exactly one slice
    if (owner || a.probs_out) {
      const uint32_t thr = a.mb->thr;
      const float invT = 1.f / cm.sp.temperature;
      MbElems e;
      const int base = blk * MB_SLICE + tid * MB_PER;
      mb_load(a, &cm.sp, draw == 0, invT, base, e);
      mb_mass(e, cm.zmax);
      if (a.probs_out)
        for (int i = 0; i < MB_PER; ++i)
          if (base + i < a.V) a.probs_out[base + i] = (e.key[i] >= thr) ? (float)((double)e.q[i] / (double)kept) : 0.f;
      if (owner) {
        unsigned long long mine = 0;
#pragma unroll
        for (int i = 0; i < MB_PER; ++i) if (base + i < a.V && e.key[i] >= thr) mine += e.q[i];
        scan[tid] = mine;
        __syncthreads();
        for (int off = 1; off < MB_THREADS; off <<= 1) {   // block-uniform branch (owner): barriers are safe
          unsigned long long v = 0;
          if (tid >= off) v = scan[tid - off];
          __syncthreads();
          scan[tid] += v;
          __syncthreads();
        }
        const unsigned long long excl = before + scan[tid] - mine;
        if (mine > 0 && target >= excl && target < excl + mine) {
          unsigned long long run = excl;
#pragma unroll
          for (int i = 0; i < MB_PER; ++i) {
            if (base + i < a.V && e.key[i] >= thr) {
              if (target < run + e.q[i] && s_token < 0) { s_token = base + i; }
              run += e.q[i];
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (forced > 0) owner = blk == 0;     // nothing was asked of the (stale) logits: exactly one block writes the token
  if (!owner) return;
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
    for (int c = tid; c < (a.d >> 3); c += MB_THREADS) dst[c] = src[c];
  }
}

void launch_sample_mb(const SampleArgs& a, hipStream_t s) {
  const int nblk = (a.V + MB_SLICE - 1) / MB_SLICE;   // <= DTK_SAMPLE_MB_MAX_SLICES
  const dim3 g(nblk, a.bs ? a.nslots : 1), b(MB_THREADS);
  hipLaunchKernelGGL(k_smb_max, g, b, 0, s, a);
  hipLaunchKernelGGL((k_smb_hist<3>), g, b, 0, s, a, nblk);
  hipLaunchKernelGGL((k_smb_hist<2>), g, b, 0, s, a, nblk);
  hipLaunchKernelGGL((k_smb_hist<1>), g, b, 0, s, a, nblk);
  hipLaunchKernelGGL((k_smb_hist<0>), g, b, 0, s, a, nblk);
  hipLaunchKernelGGL(k_smb_kept, g, b, 0, s, a, nblk);
  hipLaunchKernelGGL(k_smb_draw, g, b, 0, s, a, nblk);
}
