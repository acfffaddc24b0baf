// batch_epi.h — what the batched decode GEMV kernels share: the row maps of the paired row tiles (RoPE pairs, gate / up pairs),
// the epilogues with the reference's rounding points, the LDS-DMA piece.  Included by kernels_batch_gemm.hip and kernels_batch_mx.hip.
#pragma once
#include "kernels.h"

// rows of tile t of row group g (a group = the T tiles one wave owns)
template <int EPI, int T>
__device__ __forceinline__ int gg_tile_row0(const GemvBArgs& a, int g, int t) {
  if (EPI == EPI_QKV) return (g >> 2) * 128 + (g & 3) * 16 + t * 64;   // 16 RoPE pairs (i, i + 64) of one head block
  if (EPI == EPI_SWIGLU) return g * 16 + t * a.ff;                     // gate rows, up rows
  return (g * T + t) * 16;
}
template <int EPI, int T>
__host__ __device__ __forceinline__ int gg_groups(int N, int ff, int H, int KVH) {
  if (EPI == EPI_QKV) return (H + 2 * KVH) * 4;
  if (EPI == EPI_SWIGLU) return (ff + 15) / 16;
  return (N + 16 * T - 1) / (16 * T);
}

// one output element set: slot n, row m of the group's tiles, v[t] = reduced fp32 sums
template <int EPI, int T>
__device__ __forceinline__ void gg_epilogue(const GemvBArgs& a, int g, int n, int m, const float (&v)[T]) {
  if (EPI == EPI_RESID) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = gg_tile_row0<EPI, T>(a, g, t) + m;
      if (row < a.N) {
        bf16_t* y = a.Y + (size_t)n * a.ldy + row;
        *y = f2bf(bf2f(*y) + rbf(v[t]));
      }
    }
  } else if (EPI == EPI_LOGITS) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = gg_tile_row0<EPI, T>(a, g, t) + m;
      if (row < a.N) a.logits[(size_t)n * a.N + row] = rbf(v[t]);
    }
  } else if (EPI == EPI_SWIGLU) {
    const int i = g * 16 + m;
    if (i < a.ff) {
      const float gte = rbf(v[0]), up = rbf(v[T - 1]);
      const float sl = rbf(gte / (1.f + expf(-gte)));
      a.Y[xtile_off(n, i, (a.ff + 31) >> 5)] = f2bf(sl * up);   // input of the down projection: fragment-major
    }
  } else if (EPI == EPI_QKV) {
    const int hb = g >> 2, i = (g & 3) * 16 + m;
    const int sec = hb < a.H ? 0 : (hb < a.H + a.KVH ? 1 : 2);
    const int head = sec == 0 ? hb : (sec == 1 ? hb - a.H : hb - a.H - a.KVH);
    const int pos = a.st[n].pos;
    const float x1 = rbf(v[0]), x2 = rbf(v[T - 1]);
    const size_t slot_kv = (size_t)n * a.kv_slot_stride;
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

// The epilogue of one unit (T = 2 paired row tiles x 64 slots) from the accumulators of a wave of k_gemv_bx / k_gemv_bl / k_gemv_br:
// lane holds rows (lane >> 4) * 4 + r, column (slot) nt * 16 + (lane & 15) of each tile.  Everything the epilogue READS — the
// slots' active flags and positions, the fp8 row scales, the RoPE table entries of 16 (slot, row) pairs — is issued up front:
// written as a loop of `if (!active) continue; pos = ...; cos = table[pos]...` it was up to a dozen dependent L2 round trips at
// the end of every wave (round 4: the same mistake that made the GEMM epilogues half of a ViT launch).  Same arithmetic, same
// rounding points as gg_epilogue.
template <int EPI, int T, bool F8, int NT = 4>      // NT = 16-slot column tiles the wave holds (k_gemv_mxu: 1, 2 or 4), the first of them being tile nt0 (k_gemv_bc: a wave per column tile)
__device__ __forceinline__ void gg_finish_unit(const GemvBArgs& a, int g, const f32x4 (&tot)[T][NT], int lane, int nt0 = 0) {
  static_assert(T == 2, "paired row tiles");
  const int m0 = (lane >> 4) * 4;
  int act[NT], pos[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (nt0 + nt) * 16 + (lane & 15);
    act[nt] = a.bs->active[n];
    pos[nt] = (EPI == EPI_QKV) ? a.st[n].pos : 0;
  }
  float sc[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[t][r] = 1.f;
      if (F8) {
        int row = gg_tile_row0<EPI, T>(a, g, t) + m0 + r;
        if (row >= a.N) row = a.N - 1;
        sc[t][r] = a.wscale[row];                                 // power of two: exact
      }
    }
  // A lane holds FOUR consecutive rows (m0 .. m0 + 3) of each of its columns: 8 contiguous bytes of bf16 in every destination (cache
  // rows, q, the fragment-major activation), 16 of fp32 logits — one store per lane, tile and column instead of four 2-byte ones
  // (round 6, profiles/r06d_bc_probe.txt: the epilogue's cost was its stores; the launchers of these kernels admit only ff % 16 == 0
  // and N % 32 == 0, so no tile is ragged — the ragged shapes go to k_gemv_b, which stores element by element through gg_epilogue).
  if (EPI == EPI_QKV) {
    const int hb = g >> 2;
    const int sec = hb < a.H ? 0 : (hb < a.H + a.KVH ? 1 : 2);
    const int head = sec == 0 ? hb : (sec == 1 ? hb - a.H : hb - a.H - a.KVH);
    float cs[NT][4], sn[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cs[nt][r] = 1.f; sn[nt][r] = 0.f;
        if (sec != 2) {      // (an inactive slot's stale position is clamped: its table entry is read and dropped)
          const int p = min(max(pos[nt], 0), a.T_max - 1), i = (g & 3) * 16 + m0 + r;
          cs[nt][r] = bf2f(a.rope_cos[(size_t)p * 64 + i]);
          sn[nt][r] = bf2f(a.rope_sin[(size_t)p * 64 + i]);
        }
      }
    const int i0 = (g & 3) * 16 + m0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (nt0 + nt) * 16 + (lane & 15);
      if (!act[nt]) continue;
      const size_t slot_kv = (size_t)n * a.kv_slot_stride;
      float lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = rbf(tot[0][nt][r] * sc[0][r]), x2 = rbf(tot[1][nt][r] * sc[1][r]);
        if (sec == 2) { lo[r] = x1; hi[r] = x2; }
        else {
          const float c = cs[nt][r], sv = sn[nt][r];
          lo[r] = rbf(rbf(x1 * c) + rbf(-x2 * sv));
          hi[r] = rbf(rbf(x2 * c) + rbf(x1 * sv));
        }
      }
      bf16_t* dst = (sec == 0) ? (a.q_out + (size_t)n * a.d + head * 128)
                               : ((sec == 1 ? a.kcache : a.vcache) + slot_kv + ((size_t)head * a.T_max + pos[nt]) * 128);
      *reinterpret_cast<u32x2*>(dst + i0) = (u32x2){pack2(lo[0], lo[1]), pack2(lo[2], lo[3])};
      *reinterpret_cast<u32x2*>(dst + i0 + 64) = (u32x2){pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (nt0 + nt) * 16 + (lane & 15);
    if (!act[nt]) continue;
    if (EPI == EPI_SWIGLU) {
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gte = rbf(tot[0][nt][r] * sc[0][r]), up = rbf(tot[1][nt][r] * sc[1][r]);
        const float sl = rbf(gte / (1.f + expf(-gte)));
        y[r] = sl * up;
      }
      *reinterpret_cast<u32x2*>(a.Y + xtile_off(n, g * 16 + m0, (a.ff + 31) >> 5)) = (u32x2){pack2(y[0], y[1]), pack2(y[2], y[3])};
    } else if (EPI == EPI_LOGITS) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int row = gg_tile_row0<EPI, T>(a, g, t) + m0;
        if (!(a.N & 3) && row + 3 < a.N) {
          *reinterpret_cast<f32x4*>(a.logits + (size_t)n * a.N + row) =
              (f32x4){rbf(tot[t][nt][0] * sc[t][0]), rbf(tot[t][nt][1] * sc[t][1]), rbf(tot[t][nt][2] * sc[t][2]), rbf(tot[t][nt][3] * sc[t][3])};
        } else {                          // the last tile of a vocabulary that is no multiple of 32 (k_gemv_mxu: cl-7b's 32 024)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row + r < a.N) a.logits[(size_t)n * a.N + row + r] = rbf(tot[t][nt][r] * sc[t][r]);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v[T] = {tot[0][nt][r] * sc[0][r], tot[1][nt][r] * sc[1][r]};
        gg_epilogue<EPI, T>(a, g, n, m0 + r, v);
      }
    }
  }
}

// gg_finish_unit for ONE column tile, in two halves (k_gemv_bc): everything the epilogue READS — the slot's active flag and position,
// the fp8 row scales, the RoPE table entries — does not depend on the sums, so a kernel issues `gg_pre_load` before its k loop and
// finds the values in registers at the end (leaving parts of k_gemv_bc out, profiles/r06d_bc_probe.txt: the epilogue cost 4.8 us per
// launch, two or three dependent L2 round trips behind the last MFMA).  Same arithmetic and rounding points as gg_finish_unit.
template <int EPI, int T, bool F8>
struct gg_pre {
  int act, pos;
  float sc[T][4];
  float cs[4], sn[4];
};
template <int EPI, int T, bool F8>
__device__ __forceinline__ void gg_pre_load(const GemvBArgs& a, int g, int lane, int nt0, gg_pre<EPI, T, F8>& o) {
  static_assert(T == 2, "paired row tiles");
  const int m0 = (lane >> 4) * 4, n = nt0 * 16 + (lane & 15);
  o.act = a.bs->active[n];
  o.pos = (EPI == EPI_QKV) ? a.st[n].pos : 0;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o.sc[t][r] = 1.f;
      if (F8) {
        int row = gg_tile_row0<EPI, T>(a, g, t) + m0 + r;
        if (row >= a.N) row = a.N - 1;
        o.sc[t][r] = a.wscale[row];
      }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    o.cs[r] = 1.f; o.sn[r] = 0.f;
    if (EPI == EPI_QKV) {
      const int hb = g >> 2;
      if (hb < a.H + a.KVH) {      // q / k sections (an inactive slot's stale position is clamped: its table entry is read and dropped)
        const int p = min(max(o.pos, 0), a.T_max - 1), i = (g & 3) * 16 + m0 + r;
        o.cs[r] = bf2f(a.rope_cos[(size_t)p * 64 + i]);
        o.sn[r] = bf2f(a.rope_sin[(size_t)p * 64 + i]);
      }
    }
  }
}
template <int EPI, int T, bool F8>
__device__ __forceinline__ void gg_pre_store(const GemvBArgs& a, int g, const f32x4 (&tot)[T], int lane, int nt0, const gg_pre<EPI, T, F8>& o) {
  // A lane holds FOUR consecutive rows (m0 .. m0 + 3) of its column: their bf16 results are 8 contiguous bytes in every destination
  // (cache rows, q, the fragment-major activation: xtile_off keeps k & 7 innermost), fp32 logits 16 — one store per lane and tile
  // instead of four 2-byte ones (the launcher admits only ff % 16 == 0 and N % 32 == 0, so no row of a tile is ragged).
  if (!o.act) return;
  const int m0 = (lane >> 4) * 4, n = nt0 * 16 + (lane & 15);
  if (EPI == EPI_QKV) {
    const int hb = g >> 2;
    const int sec = hb < a.H ? 0 : (hb < a.H + a.KVH ? 1 : 2);
    const int head = sec == 0 ? hb : (sec == 1 ? hb - a.H : hb - a.H - a.KVH);
    const size_t slot_kv = (size_t)n * a.kv_slot_stride;
    const int i0 = (g & 3) * 16 + m0;
    float lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x1 = rbf(tot[0][r] * o.sc[0][r]), x2 = rbf(tot[1][r] * o.sc[1][r]);
      if (sec == 2) { lo[r] = x1; hi[r] = x2; }
      else {
        const float c = o.cs[r], sv = o.sn[r];
        lo[r] = rbf(rbf(x1 * c) + rbf(-x2 * sv));
        hi[r] = rbf(rbf(x2 * c) + rbf(x1 * sv));
      }
    }
    bf16_t* dst = (sec == 0) ? (a.q_out + (size_t)n * a.d + head * 128)
                             : ((sec == 1 ? a.kcache : a.vcache) + slot_kv + ((size_t)head * a.T_max + o.pos) * 128);
    *reinterpret_cast<u32x2*>(dst + i0) = (u32x2){pack2(lo[0], lo[1]), pack2(lo[2], lo[3])};
    *reinterpret_cast<u32x2*>(dst + i0 + 64) = (u32x2){pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
    return;
  }
  if (EPI == EPI_SWIGLU) {
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gte = rbf(tot[0][r] * o.sc[0][r]), up = rbf(tot[1][r] * o.sc[1][r]);
      const float sl = rbf(gte / (1.f + expf(-gte)));
      y[r] = sl * up;
    }
    *reinterpret_cast<u32x2*>(a.Y + xtile_off(n, g * 16 + m0, (a.ff + 31) >> 5)) = (u32x2){pack2(y[0], y[1]), pack2(y[2], y[3])};
    return;
  }
  if (EPI == EPI_LOGITS) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = gg_tile_row0<EPI, T>(a, g, t) + m0;
      *reinterpret_cast<f32x4*>(a.logits + (size_t)n * a.N + row) =
          (f32x4){rbf(tot[t][0] * o.sc[t][0]), rbf(tot[t][1] * o.sc[t][1]), rbf(tot[t][2] * o.sc[t][2]), rbf(tot[t][3] * o.sc[t][3])};
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v[T] = {tot[0][r] * o.sc[0][r], tot[1][r] * o.sc[1][r]};
    gg_epilogue<EPI, T>(a, g, n, m0 + r, v);
  }
}

// One 1 KiB fragment, global -> LDS, no VGPR: lane l's 16 bytes land at lds_byte + 16 l (guides/cdna_hip_programming.md §5.7:
// M0 carries the wave-uniform LDS address and is restored; the load is invisible to the compiler's vmcnt bookkeeping).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
