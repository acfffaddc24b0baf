// kernels.h — host-callable launchers of the gfx950 kernels (one stream argument each).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <string.h>

// ---------------------------------------------------------------- decode-step GEMV family
enum { PRO_COPY = 0, PRO_RMSNORM = 1, PRO_ATTN = 2 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_SWIGLU = 3, EPI_LOGITS = 4 };

struct GemvArgs {
  const bf16_t* W;       // [N][K] row-major, K contiguous
  const uint8_t* W8;     // fp8 (e4m3) copy of W, [N][K] bytes, or null; then wscale[N] = per-row 2^e scales
  const float* wscale;
  int N;                 // weight rows
  int K;                 // input dim (multiple of 8)
  // prologue inputs
  const bf16_t* x;       // PRO_COPY / PRO_RMSNORM: input vector [K]
  const bf16_t* norm_w;  // PRO_RMSNORM: weight [K]
  float eps;
  const float* pm;       // PRO_ATTN: split maxima   [H][S]
  const float* pl;       //           split sums     [H][S]
  const float* po;       //           split outputs  [H][S][128]
  int S;
  // epilogue outputs
  bf16_t* y;             // EPI_STORE: out[N]; EPI_RESID: residual stream (in place); EPI_SWIGLU: act[ff]
  float* logits;         // EPI_LOGITS
  // EPI_QKV
  bf16_t* q_out;         // [d]
  bf16_t* kcache;        // this layer's K cache [H][T_max][128]
  bf16_t* vcache;        // this layer's V cache
  const bf16_t* rope_cos;  // [max_pos][64]
  const bf16_t* rope_sin;
  const DecState* st;
  int T_max;
  int d;                 // hidden (== H*128)
  int ff;                // EPI_SWIGLU: N == 2*ff
  int H;                 // EPI_QKV: query heads; rows = [H q heads | KVH k heads | KVH v heads] x 128, N == (H + 2*KVH)*128
  int KVH;               //          key/value heads (GQA: H % KVH == 0; MHA: KVH == H)
};

void launch_gemv(int pro, int epi, const GemvArgs& a, hipStream_t s);
void launch_gemv_variant(int pro, int epi, int variant, const GemvArgs& a, hipStream_t s);
void set_gemv_default_variant(int epi, int variant);

struct AttnDecArgs {
  const bf16_t* q;       // [H*128] (RoPE applied)
  const bf16_t* kcache;  // [KVH][T_max][128]; query head h reads kv head h / G
  const bf16_t* vcache;
  const DecState* st;    // keys 0..st->pos
  float* pm; float* pl; float* po;   // [H][S], [H][S], [H][S][130] (130 = 128 o + m + l when combine)
  int H; int S; int T_max; float scale;
  int combine;           // 1: the last-arriving split of a head writes the bf16 head output
  bf16_t* out;           // [H*128] attention output (combine)
  unsigned* counters;    // [H] arrival tickets, zero between launches
  int G;                 // query heads per kv head (1 = MHA)
  int threads;           // 0: contiguous key ranges per split (k_attn_decode); 256 | 512 | 1024: k_attn_decode_t, tiles dealt round-robin to the splits
};
void launch_attn_decode(const AttnDecArgs& a, hipStream_t s);

struct SampleArgs {
  const float* logits; int V;
  const SamplingDev* sp;
  DecState* st;
  const bf16_t* embed;   // [V][d]
  bf16_t* x;             // residual stream [d] (embedding of the sampled token is written here)
  int d;
  int64_t* tok_ring;     // pinned/mapped or device ring [ring]
  int ring;
  float* probs_out;      // optional (tests): filtered, renormalised distribution [V]
  int advance;           // 1: product path (advance DecState, gather embedding)
  int step_override;     // >=0: use as draw index (tests)
  const BatchState* bs;  // batched mode: block = slot, inactive slots return; pointers are slot 0's
  int logits_stride;     // elements between slots' logits (batched mode)
  int nslots;            // batched mode: grid = 16, 32 or 64 slots
  SampleMB* mb;          // scratch of the multi-block sampler (slot 0's in batched mode), or null
};
void launch_sample_b(const SampleArgs& a, hipStream_t s);
void launch_sample(const SampleArgs& a, hipStream_t s);
// 7-kernel chain for V > 32768 (greedy / temperature / top-p; not top-k); single sequence or (a.bs) every active slot
void launch_sample_mb(const SampleArgs& a, hipStream_t s);
static inline bool sample_mb_supported(int V) { return V > 32768 && V <= DTK_SAMPLE_MB_MAX_SLICES * 8192; }
// on the v1 vocabularies (<= 32768) the register-resident single-block kernel is as fast (ds-7b sampling decode 366.2 tok/s
// vs 366.6 with the chain): one launch, so it stays the default there; DTK_SAMPLER=mb forces the chain for V >= 16384 (A/B)
static inline bool sample_mb_preferred(int V, bool do_sample) {
  static int force = -1;
  if (force < 0) { const char* e = getenv("DTK_SAMPLER"); force = (e && !strcmp(e, "mb")) ? 1 : 0; }
  return sample_mb_supported(V) || (force && do_sample && V >= 16384 && V <= 32768);
}

// ---------------------------------------------------------------- batched decode (up to 64 slots share W)
struct GemvBArgs {
  const bf16_t* W; int N; int K;       // weight [N][K] in the fragment-major (tiled) copy
  const bf16_t* X; int ldx;            // inputs [16][ldx] (slot-major)
  bf16_t* Y; int ldy;                  // RESID: residual streams [16][ldy]; SWIGLU: act [16][ldy]; STORE
  float* logits;                       // LOGITS: [16][N]
  const BatchState* bs;
  const DecState* st;                  // [16]
  bf16_t* q_out;                       // QKV: [16][d]
  bf16_t* kcache; bf16_t* vcache;      // this layer's caches of slot 0; slot s at + s*kv_slot_stride
  size_t kv_slot_stride;
  const bf16_t* rope_cos; const bf16_t* rope_sin;
  int T_max; int d; int ff;
  int H; int KVH;                      // QKV: head counts (rows = [H | KVH | KVH] x 128)
  const uint8_t* W8;                   // fp8 (e4m3) pair-tiled copy of the weights, or null; then wscale[N] = per-row 2^e scales
  const float* wscale;
  int nt;                              // 16-slot column tiles: 1 (<= 16 slots), 2 (<= 32) or 4 (<= 64)
  float* kpart; unsigned* kctr;        // k_gemv_bk: K-split partials [8][N / 16][4][256] fp32, one arrival counter per row tile (zero between launches)
  unsigned* err;                       // sticky count of expired in-kernel hand-off waits (the LDS-ring kernels), or null
  // fp8 matrix-core path (kernels_batch_mx.hip): MX weight tiles, the slots' MXFP8 input vectors + their E8M0 scales; SWIGLU writes
  // the down projection's MXFP8 input (groups of 16) to Y8 / YS
  const uint8_t* Wm; const uint8_t* X8; const uint8_t* XS;
  uint8_t* Y8; uint8_t* YS;
};
void launch_gemv_b(int epi, const GemvBArgs& a, hipStream_t s);
bool launch_gemv_bx(int epi, int variant, const GemvBArgs& a, hipStream_t s);    // kernels_batch_gemm.hip: x once per CU through LDS phases (64 slots; bit-identical to k_gemv_b); false = not covered
bool launch_gemv_bl(int epi, const GemvBArgs& a, hipStream_t s);    // kernels_batch_gemm.hip: k_gemv_bx with both operands streamed into LDS rings by a loader wave (LDS-DMA); false = not covered / off
bool launch_gemv_bc(int epi, const GemvBArgs& a, hipStream_t s);    // kernels_batch_gemm.hip: a compute wave per COLUMN tile, x from L2 into registers, the weights through an LDS ring (bit-identical to k_gemv_b); false = not covered / off
void set_gemv_bc(int v);       // 0 off; 128 = measured default per role; bit 0 qkv, bit 1 gate/up, bit 2 lm_head; bits 4..6 force the units per block
void set_gemv_bl(int v);       // bit 0: gate/up + lm_head, bit 1: qkv by pair units, bit 2: fp8 weights too, bit 3: qkv as pair unit + V row tile per block where that fills the chip, bit 4: for any MHA model
void set_gemv_br_wd(int v);    // k_gemv_br with fp8 weights: phases of the register ring, 4 (default) | 8 (measured slower)
void set_gemv_loaders(int v);  // loader waves of the Q3 qkv kernel: 1 (ring of 3 phases) or 2 (alternate phases, ring of 5)
void set_gemv_xw(int v);       // k_gemv_bl / k_gemv_bkl: 1 = x fragments by an extra wave's ordinary loads + ds_write_b128 instead of LDS-DMA pieces
bool launch_gemv_bus(int epi, const GemvBArgs& a, hipStream_t s);   // kernels_batch_ks.hip: qkv / gate-up at 64 slots, a block per CU whose 8 waves are the 8 K slices, every operand straight into the wave's registers (bit-identical to k_gemv_b); false = not covered / off
void set_gemv_bus(int v);      // 0 off; 128 = measured default per role and weight format; else bit 0 qkv, bit 1 gate/up
void set_gemv_bkl(int v);      // 1: the resid_kparts weight kernel with LDS-DMA operand rings (k_gemv_bkl) instead of k_gemv_bkp
// N = d roles at 33..64 slots as two launches (kernels_batch_gemm.hip): k_gemv_bkp = K split over the CUs of a row group, plain stores
// of the fp32 partials; k_resid_norm_b = reduce + residual + the RMSNorm that follows the role anyway (replaces k_rmsnorm_b there)
bool resid_kparts_covers(const GemvBArgs& a);
void launch_gemv_bkp(const GemvBArgs& a, hipStream_t s);
void launch_resid_norm_b(const float* part, bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int D, float eps, const BatchState* bs,
                         int nslots, hipStream_t s, uint8_t* Y8 = nullptr, uint8_t* YS = nullptr);
void set_resid_split(int v);   // batched N = d roles at 64 slots: 0 = one row tile x 64 slots per block, 1 = two row tiles x 32 slots
void set_gemv_bx(int v);       // 0: off, 1: on (units per block from the CU count), 2..4: on with that many units per block
void set_gemv_b_wide(int v);   // row tiles per block of the batched kernels: 0 round-1 shapes, 1 twice as many, 2 auto (wide from 32 slots)
void launch_retile(const bf16_t* src, bf16_t* dst, int N, int K, hipStream_t s);
static inline size_t tiled_elems(int N, int K) { return (size_t)((N + 15) >> 4) * ((K + 31) >> 5) * 512; }
// fp8: a 1 KiB tile covers 16 rows x 64 k (two MFMA k-steps): lane l holds row l&15, bytes 0..7 = k0 + (l>>4)*8 + 0..7,
// bytes 8..15 = the same columns of the next k-step (k0 + 32 + ...)
void launch_retile_f8(const uint8_t* src, uint8_t* dst, int N, int K, hipStream_t s);
static inline size_t tiled_bytes_f8(int N, int K) { return (size_t)((N + 15) >> 4) * ((K + 63) >> 6) * 1024; }
// Y8 / YS non-null: the normalised rows go out as MXFP8 (groups of 32, mx_quant.h) instead of bf16 fragments
void launch_rmsnorm_b(const bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int ldy, int D, float eps,
                      const BatchState* bs, int nslots, hipStream_t s, uint8_t* Y8 = nullptr, uint8_t* YS = nullptr);
// ---- fp8 matrix-core path of the batched step (kernels_batch_mx.hip)
void launch_retile_mx(const uint8_t* src, uint8_t* dst, int N, int K, int G, hipStream_t s);       // row-major fp8 -> MX weight tiles (G = 32 | 16)
void launch_quant_mx_rows(const bf16_t* X, int K, uint8_t* X8, uint8_t* XS, int G, int nslots, hipStream_t s);   // op-level tests
bool mx_unit_covers(int K);                                   // q/k/v, gate/up, lm_head: K a multiple of 512
bool mx_kparts_covers(int N, int K, int G);                   // o_proj (G = 32), down (G = 16)
void launch_gemv_mxu(int epi, const GemvBArgs& a, hipStream_t s);
bool launch_gemv_mxk(const GemvBArgs& a, int G, hipStream_t s);
void set_mx_nc(int role, int nc);                             // compute waves per block of the unit kernel: role 0 qkv, 1 gate/up, 2 lm_head; 0 = from the CU count
struct AttnDecBArgs {
  const bf16_t* q;                     // [16][d]
  const bf16_t* kcache; const bf16_t* vcache; size_t kv_slot_stride;
  const DecState* st; const BatchState* bs;
  bf16_t* out;                         // [16][d]
  int H; int T_max; int d; float scale;
  int G;                               // query heads per kv head (1 = MHA)
  int nslots;                          // grid y of k_attn_tail_b: 1 / 2 / 4 (multi-vector step), 16, 32 or 64
  int use_prefix;                      // score BatchState's prefix groups once per group on the matrix cores (k_attn_prefix_g)
  int pfx_splits;                      // key splits of the prefix kernel (its grid y)
  int tail_threads;                    // k_attn_tail_b block: 512 (default) | 256
  int gqa_fused;             // k_attn_tail_b: 1 = the query heads of a GQA group share one block (default), 0 = a block per query head
  int nt_private = 0;        // k_attn_tail_b: non-temporal loads for key / value tiles beyond the shared prefix
  float* pfx_m; float* pfx_l; float* pfx_o;   // [slots][H][pfx_splits], ..., [slots][H][pfx_splits][128]: un-normalised prefix states
  uint8_t* out8 = nullptr; uint8_t* outs = nullptr;   // k_attn_tail_b: the head outputs as MXFP8 (groups of 32) instead of bf16 fragments
};
void launch_attn_decode_b(const AttnDecBArgs& a, hipStream_t s);

// ---------------------------------------------------------------- batched decode with <= 4 slots (kernels_decode_mv.hip)
// k_gemv_mv = k_gemv carrying NB <= 4 input vectors (slots 0..NB-1): row-major weights read once, per slot the arithmetic of
// the single-sequence kernel.
struct GemvMvArgs {
  const bf16_t* W;       // [N][K] row-major (the single-sequence copy)
  const uint8_t* W8;     // fp8 rows + wscale[N], or null
  const float* wscale;
  int N; int K;
  const bf16_t* X;       // PRO_RMSNORM: residual streams [slot][ldx]; PRO_COPY: fragment order (xtile_off) unless x_rowmajor
  int ldx;
  int x_rowmajor;        // PRO_COPY input is [slot][ldx] (op-level tests)
  const bf16_t* norm_w; float eps;
  bf16_t* Y; int ldy;    // RESID: residual streams [slot][ldy] in place; SWIGLU: activation, fragment order; STORE: [slot][ldy]
  float* logits;         // LOGITS: [slot][N]
  bf16_t* q_out;         // QKV: [slot][d]
  bf16_t* kcache; bf16_t* vcache; size_t kv_slot_stride;   // this layer's caches of slot 0
  const bf16_t* rope_cos; const bf16_t* rope_sin;
  const DecState* st;    // [slots]
  const BatchState* bs;  // active flags, or null = every slot
  int T_max; int d; int ff; int H; int KVH;
};
void launch_gemv_mv(int pro, int epi, int nb, const GemvMvArgs& a, hipStream_t s);   // nb = 1, 2 or 4 vectors
void set_gemv_mv_shape(int role, int shape);   // role = epilogue id (5 = o_proj alone); shape 0..3, -1 = measured default

// ---------------------------------------------------------------- batched (prefill / ViT) kernels
#define GEMM_BIAS 1
#define GEMM_GELU_ERF 2
#define GEMM_GELU_TANH 4
#define GEMM_RESIDUAL 8
#define GEMM_SWIGLU 16384           // k_gemm_g3 (W stage from a PAIR-INTERLEAVED fragment-major copy of [gate rows | up rows]: launch_retile_pairs): the epilogue is
                                    // k_silu_mul's arithmetic on a lane's (gate, up) accumulator pair, C = [M][N / 2] activations — no [M][N] buffer, no second pass
#define GEMM_EPI_DIRECT 32768       // k_gemm_g3: the lanes store their accumulators directly (the epilogue before round 6's LDS-transposed one; A/B and bit-identity tests)
#define GEMM_PROBE_NOFILL 65536    // k_gemm_g3 timing experiments (DTK_G3_PROBE): leave parts of the kernel out
#define GEMM_PROBE_NOMFMA 131072
struct GemmArgs {
  const bf16_t* A; int lda;      // [M][K]
  const bf16_t* W; int ldw;      // [N][K]
  const bf16_t* bias;            // [N] or null
  const bf16_t* residual; int ldr; // [M][N] or null
  bf16_t* C; int ldc;            // [M][N]
  int M, N, K;                   // K multiple of 8
  int flags;
  // sliced-K roles (launch_gemm_sk / the naive twin): K's 64-wide k-tiles are cut into `kslices` contiguous runs of
  // sk_tiles_per_slice(K, kslices) tiles; every run is an accumulation chain from zero and the runs' sums are added in run order
  // in fp32 — the canonical order of such a role at ANY M and tile shape.  part = fp32 [kslices][part_rows][N] (launch_gemm_sk).
  int kslices = 1;
  float* part = nullptr; long part_stride = 0;
  // W again as fragment-major tiles (launch_retile's image: 1 KiB per (16 rows, 32-wide k-step)), or null.  k_gemm_g3 fills its W stage from
  // it when present: a fill instruction then reads 1 KiB CONTIGUOUS instead of 8 rows x 128 B that lie K x 2 bytes apart (the row-major
  // stream of the prefill GEMMs reached 2.4 TB/s of HBM), and the piece lands in LDS as the MFMA operand.  Same values, same k order.
  const bf16_t* Wt = nullptr;
};
__host__ __device__ inline int sk_tiles_per_slice(int K, int S) { const int T = K / 64; return (T + S - 1) / S; }
// slices of a decoder-prefill role, from its WEIGHT shape alone: the largest power of two <= 8 that keeps 256 x 128 tiles x slices within
// the 256 CUs, every slice at least 4 k-tiles long; 1 = the role stays a one-chain GEMM (it fills the chip, or the sliced kernel does not take it)
#define SK_CHUNK_ROWS 512
#define SK_SL_MIN_ROWS 768      // measured (ds-7b, profiles/r06ab_long_prompts.txt): 600 rows 16.0 ms in chunks / 16.7 in one launch, 1100 rows 28.3 / 24.3, 1900 rows 46.4 / 34.8
inline int sk_role_slices(int N, int K) {
  if ((N % 4) || (K % 8) || K < 128) return 1;
  const int tiles = (N + 127) / 128;
  int S = 1;
  while (S < 8 && tiles * S * 2 <= 256) S *= 2;
  while (S > 1 && (K / 64) / S < 4) S /= 2;
  return S;
}
void launch_gemm_mfma(const GemmArgs& a, hipStream_t s);
// sliced-K GEMM (the decoder prefill's N = d and q/k/v roles): k_gemm_g3's tile and pipeline with a block per (tile, K slice) writing
// fp32 partials, then k_sk_reduce: partials summed in slice order + the GEMM epilogue (+ the RMSNorm that follows the role, fused:
// norm_w / Y / ldy / eps; norm_w = null: none).  false = the shape is not one the kernel takes (nothing launched).
bool gemm_sk_supported(const GemmArgs& a);
void set_gemm_wt(int v);        // 1 (default): k_gemm_g3 fills its W stage from GemmArgs::Wt when given; 0: always from the row-major weights (bit-identical)
void set_gemm_epi_direct(int v);   // 1: k_gemm_g3 stores from the accumulator layout (no LDS transpose); bit-identical
void set_gemm_sk_tile(int v);   // 0 = 256 x 128, 1 = 128 x 256, 2 = by M (default); bit-identical
bool launch_gemm_sk(const GemmArgs& a, const bf16_t* norm_w, bf16_t* Y, int ldy, float eps, hipStream_t s);
void launch_rmsnorm_rows_sk(const bf16_t* X, int ldx, const bf16_t* norm_w, bf16_t* Y, int ldy, int M, int N, float eps, hipStream_t s);   // k_sk_reduce's RMSNorm alone
// gate/up + SiLU*mul in one launch (GEMM_SWIGLU): a.Wt = the pair-interleaved copy, a.C / a.ldc = the activations [M][N / 2]; false = the shape does not
// take k_gemm_g3 (fewer than its minimum of blocks, misaligned operands, gemm_wt off): nothing launched, the caller runs Linear + k_silu_mul
bool launch_gemm_g3_swiglu(const GemmArgs& a, hipStream_t s);
void launch_retile_pairs(const bf16_t* src, bf16_t* dst, int ff, int K, hipStream_t s);    // [2 ff][K] row-major (gate rows, then up rows) -> fragment-major tiles in the order gate tile 0, up tile 0, gate tile 1, ...
bool launch_gemm_g3_sliced(const GemmArgs& a, hipStream_t s);        // the same role in ONE launch (second accumulator set, slices folded in registers): for large M; bit-identical
bool launch_gemm_sk_partials(const GemmArgs& a, hipStream_t s);     // the GEMM alone: a.part holds the slices' sums afterwards
// the q/k/v role's reduction fused with k_rope_scatter (same rounding points: bf16 of the summed slices, then RoPE)
void launch_sk_rope_scatter(const float* part, long part_stride, int kslices, bf16_t* Qh, bf16_t* kcache, bf16_t* vcache,
                            const bf16_t* cos_t, const bf16_t* sin_t, int T, int start_pos, int H, int KVH, int T_max, hipStream_t s);
void launch_sk_reduce_swiglu(const GemmArgs& a, bf16_t* ACT, int ldact, hipStream_t s);    // a sliced gate/up role's partials (a.part, a.N = 2 ff) -> SiLU(gate) * up, [M][ff]
void launch_sk_reduce(const GemmArgs& a, const bf16_t* norm_w, bf16_t* Y, int ldy, float eps, hipStream_t s);
void set_gemm_bk(int v);     // k-tile of the 64x64 GEMM: 64 | 128
void set_gemm_stages(int v); // register prefetch depth of the 64x64 tile: 1..4
void set_gemm_tile(int v);   // 0 auto, 1 = 64x64, 2 = 128x64, 3 = 128x128, 4 = 64x32, 5 = 32x32
void set_gemm_impl(int v);   // 0 = k_gemm_mfma (operands staged through registers), 1 = k_gemm_dma (LDS-DMA ring, fragment-major LDS), 2 = k_gemm_glds, 3 = auto, 4 = k_gemm_g3 (8 waves, 3 LDS stages)
void set_gemm_ring(int v);   // LDS stages of k_gemm_dma: 2..4
void set_gemm_g3_min_blocks(int v);     // gemm_impl 4: 256 x 128 blocks a shape needs to take k_gemm_g3 (default 128)
void set_gemm_glds_min_tiles(int v);   // gemm_impl 2: 128 x 128 tiles a shape needs to take k_gemm_glds
void launch_gemm_naive(const GemmArgs& a, hipStream_t s);

void launch_layernorm_rows(const bf16_t* X, int ldx, const bf16_t* w, const bf16_t* b,
                           bf16_t* Y, int ldy, int M, int D, float eps, hipStream_t s);
void launch_rmsnorm_rows(const bf16_t* X, int ldx, const bf16_t* w, bf16_t* Y, int ldy,
                         int M, int D, float eps, hipStream_t s);
void launch_silu_mul(const bf16_t* GU, int ff, bf16_t* ACT, int M, hipStream_t s);
void launch_embed_gather(const int32_t* ids, const bf16_t* embed, bf16_t* X, int T, int d,
                         hipStream_t s);
void launch_copy_rows(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int M, int D,
                      hipStream_t s);
void launch_im2col(const float* pixels, bf16_t* patches, int image, int patch, int ldp,
                   hipStream_t s);
// q,k RoPE + scatter of one prefill chunk: QKV [T][3d] -> Qh [H][T][128], caches at start_pos+t
void launch_rope_scatter(const bf16_t* QKV, bf16_t* Qh, bf16_t* kcache, bf16_t* vcache,
                         const bf16_t* cos_t, const bf16_t* sin_t, int T, int start_pos,
                         int H, int KVH, int T_max, hipStream_t s);

struct AttnArgs {
  const bf16_t* Q; long q_sh; long q_st;   // element strides: head, token
  const bf16_t* K; long k_sh; long k_st;
  const bf16_t* V; long v_sh; long v_st;
  bf16_t* O; long o_sh; long o_st;
  int H, Tq, Tk, hd;
  int causal; int q_offset;   // query i sits at absolute position q_offset + i
  float scale;
  int kv_group;  // query heads per K/V head (GQA): head h reads K/V head h / kv_group; 0 or 1 = one K/V head per query head
  int impl;   // 0 auto (MFMA flash kernel when hd is 72/128), 1 VALU kernel, 2 MFMA kernel
  // several independent attention problems of the same shape in ONE launch (grid z): problem b reads / writes at + b * stride
  // (elements).  The ViT batch: 8 images x 16 heads x 12 query blocks = 1536 blocks instead of 8 launches of 192 (the chip has 256 CUs)
  int nbatch = 1;
  long q_sb = 0, k_sb = 0, v_sb = 0, o_sb = 0;
};
void launch_attention(const AttnArgs& a, hipStream_t s);

// per-row fp8 quantisation with a power-of-two scale; W is overwritten with the de-quantised values
void launch_quant_fp8_rows(bf16_t* W, uint8_t* W8, float* scale, int N, int K, hipStream_t s);
void launch_fill_synth(bf16_t* dst, int64_t n, uint64_t seed, uint32_t tag, float scale,
                       float offset, hipStream_t s);
void launch_f32_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t s);
