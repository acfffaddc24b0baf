// dtk_api.hip — the C ABI (include/dtk.h): context, weight registry, ViT / prefill /
// decode orchestration, hipGraph capture of the per-token decode step.
//
// Reference call sites this replaces (all Python in potamides/DeTikZify):
//   DetikzifyVisionModel.forward / get_intermediate_layers  v1/modeling_detikzify.py:63-72
//   DetikzifyModel.get_vision_features + mm_projector        v1/modeling_detikzify.py:132-137,163
//   embedding splice                                          v1/modeling_detikzify.py:158-189
//   LlamaModel.forward / lm_head                              v1/modeling_detikzify.py:191-200,250-257
//   HF GenerationMixin._sample loop body                      via infer/generate.py:218-227
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dtk.h"
#include "kernels.h"
#include "mx_quant.h"

int& dtk_lds_attr_error(int device) {     // common.h: set by a launcher whose hipFuncSetAttribute was refused, per device
  static int e[65] = {};
  return e[(device >= 0 && device < 64) ? device : 64];
}

namespace {

thread_local std::string g_create_error;
// dtk_last_error is per calling thread (ABI 6): the engine's loop thread, reward threads inside dtk_vit_encode and a caller's own
// thread may all fail on one context; each reads the text of ITS failed call
thread_local std::string g_thread_error;
thread_local const void* g_thread_error_ctx = nullptr;

struct TensorEntry {
  std::string name;
  bf16_t* ptr = nullptr;   // destination (view into the arena)
  int64_t rows = 1;        // logical rows
  int64_t cols = 0;        // logical row length (elements)
  int64_t stride = 0;      // destination row stride (elements), >= cols
  float synth_scale = 0.02f;
  float synth_offset = 0.f;
  bool loaded = false;     // written by dtk_load_tensor / dtk_fill_synthetic
  int64_t numel() const { return rows * cols; }
};

struct SeqHost {   // host-side mirror of one sequence's decode state
  std::vector<int64_t> cached_ids;
  bool cached_with_image = false;  // the cached KV was computed with image features spliced in
  uint64_t image_key = 0;          // ... of the image with this caller-supplied key (0 = unknown: never reused across images)
  int host_next_pos = 0;           // tokens with KV after all launched steps
  bool have_logits = false;
  int share_src = -1;              // slot whose first share_len cached tokens are bit-identical to ours (dtk_kv_fork), or -1
  int share_len = 0;
  int last_reuse_start = 0;        // tokens of the previous cache the last prefill kept
};

struct LayerW {
  bf16_t *wqkv, *wo, *wgu, *wdown, *ln1, *ln2;
  bf16_t *t_wqkv = nullptr, *t_wo = nullptr, *t_wgu = nullptr, *t_wdown = nullptr;  // fragment-major copies (batched decode)
  bf16_t *p_wqkv = nullptr, *p_wo = nullptr, *p_wgu = nullptr, *p_wdown = nullptr;  // fragment-major bf16 copies the prefill GEMMs stream (= t_* where those exist)
  bf16_t* p_wgui = nullptr;          // gate/up again with its row tiles in (gate, up) pair order: the W stage of the SwiGLU-fused prefill GEMM (models whose gate/up role is one chain)
  uint8_t *q_wqkv = nullptr, *q_wo = nullptr, *q_wgu = nullptr, *q_wdown = nullptr;  // fp8 e4m3 copies (weight_format 1)
  float *s_wqkv = nullptr, *s_wo = nullptr, *s_wgu = nullptr, *s_wdown = nullptr;    // per-row power-of-two scales
  uint8_t *t8_wqkv = nullptr, *t8_wo = nullptr, *t8_wgu = nullptr, *t8_wdown = nullptr;  // fp8 pair-tiled copies (batched decode, fp8)
  uint8_t *m_wqkv = nullptr, *m_wo = nullptr, *m_wgu = nullptr, *m_wdown = nullptr;      // fp8 MX tiles (fp8 matrix-core step; down in groups of 16)
};
struct VitBlockW {
  bf16_t *n1w, *n1b, *qkvw, *qkvb, *projw, *projb, *n2w, *n2b, *fc1w, *fc1b, *fc2w, *fc2b;
};

}  // namespace

struct dtk_ctx {
  dtk_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::mutex err_mu;
  // dtk_vit_encode (own stream, any thread: the SelfSim rewards) and a prefill (the engine's loop thread) share the ViT activation
  // buffers and `cur_stream`: one at a time
  std::mutex vit_mu;

  // derived sizes
  int d, L, H, ff, V, Tmax, S;           // S: split-K factor of the single-sequence decode attention
  int vD, vDepth, vH, vHd, vMlp, vN, vPatchK, vPatchLd, nImg;

  // weights
  unsigned char* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<TensorEntry> tensors;
  std::unordered_map<std::string, int> tindex;
  std::vector<LayerW> layers;
  bf16_t *embed, *final_norm, *lm_head, *mm_w, *mm_b;
  bf16_t *rope_cos, *rope_sin;
  std::vector<VitBlockW> vblocks;
  bf16_t *pe_w, *pe_b, *pos_embed, *vnorm_w, *vnorm_b;
  bf16_t *ap_latent, *ap_qw, *ap_qb, *ap_kvw, *ap_kvb, *ap_pw, *ap_pb, *ap_nw, *ap_nb, *ap_f1w,
      *ap_f1b, *ap_f2w, *ap_f2b;

  // KV cache [L][2][H][Tmax][128]
  bf16_t* kv = nullptr;

  // activations: decoder prefill
  bf16_t *X, *Xn, *QKV, *Qh, *AO, *GU, *ACT;
  int sk_sl_min_rows = SK_SL_MIN_ROWS;   // above this many rows a sliced role is one launch with the slices folded in registers (dtk_set_option "sk_sl_min_rows"; bit-identical)
  float* skpart = nullptr;           // fp32 partials of the sliced-K prefill GEMMs: [kslices][SK_CHUNK_ROWS][N], one role at a time
  size_t skpart_floats = 0;
  int swiglu_fused = 1;              // gate/up's epilogue is SiLU*mul (dtk_set_option "swiglu_fused"; bit-identical)
  int qkv_rope_fused = 1;            // a sliced q/k/v role reduces inside the RoPE + KV-append kernel (dtk_set_option "qkv_rope_fused"; bit-identical)
  int prefill_sk = 1;                // sliced-K prefill GEMMs for the roles with <= 128 tiles of 256 x 128 (dtk_set_option "prefill_sk": 0 = the one-chain kernels, 2 / 4 / 8 = a cap on the slices)
  int32_t* ids_dev = nullptr;
  // decode step
  bf16_t *x, *q, *act;
  float *logits, *pm, *pl, *po;
  bf16_t* attn_out = nullptr;        // combined attention output [d] (in-kernel combine)
  unsigned* attn_ctr = nullptr;      // [H] arrival tickets
  int attn_combine = 0;              // 0: consumer (o_proj prologue), 1: last-arriver in k_attn_decode, 2: own kernel
  DecState* st = nullptr;
  SamplingDev* sp = nullptr;
  int64_t* tok_ring_dev = nullptr;   // device ring
  int64_t* tok_ring_host = nullptr;  // pinned host mirror
  float* probs_dev = nullptr;        // op_sample output
  // ViT
  float* pixels_dev = nullptr;
  bf16_t *patches, *VX, *VN, *VQKV, *VAO, *VH, *feats, *last_hidden;
  bf16_t *pq, *pkv, *pao, *px, *pn, *ph, *pooled;
  bf16_t* IMG;  // projected image embeddings [nImg][d]
  // op-level scratch
  unsigned char* scratch = nullptr;
  size_t scratch_bytes = 0;

  // host state
  SeqHost seq0;              // the single-sequence API (dtk_prefill / dtk_decode*)
  int wfmt = 0;              // 0 = bf16 decoder weights, 1 = fp8 e4m3 + per-row 2^e scale (dtk_config.reserved[1])
  bool fp8_ready = false;
  uint8_t* q_lm_head = nullptr;
  float* s_lm_head = nullptr;
  uint64_t cached_image_key = 0;
  bool have_image = false;
  // ---- batched decode (dtk_*_slot / dtk_decode_batch_*): up to 64 decoding slots (+1) with their own KV
  int KVH = 0;                       // key/value heads (dtk_config.reserved[2]; 0 -> heads)
  bool proj_bias = true;             // mm_projector has a bias (v1) / bias-free connector (v2)
  int nb = 0;                        // number of batch slots (dtk_config.reserved[0])
  bool share_reads = true;           // forked slots read their shared prefix from the source slot (dtk_set_option "share_prefix_reads")
  int nt = 1;                        // 16-slot column tiles of the batched kernels (2 when nb > 17)
  std::vector<SeqHost> bseq;
  bf16_t* kvb = nullptr;             // [nb][L][2][H][Tmax][128]
  size_t kv_slot_stride = 0;
  bf16_t *xb = nullptr, *xnb = nullptr, *qb = nullptr, *aob = nullptr, *actb = nullptr;  // [16][d|ff]
  float* logits_b = nullptr;
  float* kpart = nullptr; unsigned* kctr = nullptr;   // k_gemv_bk / k_gemv_bkp partial sums + arrival counters
  int attn_nt = 1;             // batched attention: non-temporal loads of private K / V tiles (64 slots x 500 private keys: 7.14 -> 6.74 ms per step) (dtk_set_option "attn_nt")
  bool resid_kparts = true;    // batched N = d roles at 64 slots as two launches (k_gemv_bkp + k_resid_norm_b; measured 20.2 -> 21.2 rollouts/s): dtk_set_option("resid_kparts")
  float *pfx_m = nullptr, *pfx_l = nullptr, *pfx_o = nullptr;   // shared-prefix states [64][H][4] (+ x 128)
  int prefix_mfma = 0;               // shared prefixes (forks of one image) scored once per group of <= 16 slots on the matrix cores (k_attn_prefix_g); dtk_set_option "prefix_mfma".
                                     // Set at dtk_create from the context's size: 1 with 64 decoding slots, 0 below (see there)
  int pfx_splits = 4;                // key splits of that kernel (its grid z)
  int gqa_fused = 1;                 // batched attention: one block per (K/V head, slot) for GQA models
  int tail_threads = 256;            // block of k_attn_tail_b (rows per memory round trip = threads / 4); dtk_create: 128 with 64 decoding slots (the blocks then walk
                                     // private keys only and 2048 blocks of 2 waves are all resident), 256 below
  DecState* st_b = nullptr;          // [16]
  SamplingDev* sp_b = nullptr;       // [16]
  BatchState* bs_dev = nullptr;
  bf16_t* t_lm_head = nullptr;       // fragment-major copy of lm_head
  uint8_t* t8_lm_head = nullptr;     // fp8 pair-tiled copy of lm_head (weight_format fp8)
  // fp8 matrix-core step (kernels_batch_mx.hip; dtk_set_option "act_fp8"): the MFMA-family step of an fp8 model runs its GEMVs as
  // v_mfma_scale_f32_16x16x128_f8f6f4 on MXFP8 activations.  mx_ok = the model's shapes are covered and the buffers exist
  uint8_t* m_lm_head = nullptr;
  uint8_t *xn8 = nullptr, *xns = nullptr, *ao8 = nullptr, *aos = nullptr, *act8 = nullptr, *acts = nullptr;
  bool mx_ok = false;
  // act_fp8: 0 (default since round 5) = bf16 activations (fp8 weights widened in registers, bf16 MFMA: rounds 1-3); 1 = opt-in.  MXFP8
  // activations keep 3 mantissa bits: logits move ~1e-1 rel-L2 from the bf16-activation result (tests/test_gpu_parity_mx.py,
  // test_mxfp8_activations_against_bf16_activations), so the faster step is the caller's choice, not the library's.
  int act_fp8 = 0;
  bool launch_refused = false;       // a launcher of the step being issued had no kernel for its shape (nothing was launched for that role)
  bool tiled_ready = false;          // the fragment-major copies match the row-major weights
  bool ptiled_ready = false;         // ... and the prefill's own (LayerW::p_*)
  BatchState* bs_host = nullptr;     // pinned ring [DTK_MAX_INFLIGHT]
  SamplingDev* sp_stage = nullptr; uint32_t* draw_stage = nullptr;   // pinned [DTK_MAX_SLOTS]: per-slot set_sampling uploads queued on the stream
  DecState* st_stage = nullptr;      // pinned [DTK_MAX_SLOTS]: dtk_resume_slot's state upload (no stream sync: a slot is resumed again only sequences later)
  int64_t* tokb_dev = nullptr;       // [DTK_MAX_INFLIGHT][16]
  int64_t* tokb_host = nullptr;      // pinned mirror
  uint64_t blaunched = 0, bwaited = 0;
  hipEvent_t bstep_done[DTK_MAX_INFLIGHT] = {};
  // one captured step per column-tile count (1, 2, 4 tiles of 16 slots): a step only pays for the tiles up to its highest
  // active slot (32 trees on a 65-slot context run the 2-tile kernels)
  // (+ three more for the multi-vector step of a context with <= 5 slots: 1, 2 or 4 vectors)
  bool bgraph_ready[6] = {false, false, false, false, false, false};
  hipGraph_t bgraph[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t bgraph_exec[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int nt_step = 1;                   // tile count of the step being launched / captured
  // Contexts with at most 5 slots (<= 4 decoding + a prefix slot: BASELINE config 4's 2 / 4 trees per rank at N = 8 / 4) decode
  // with the multi-vector kernels (kernels_decode_mv.hip): the single-sequence GEMVs carrying 1, 2 or 4 x vectors instead of a
  // 16-column MFMA tile.  The kernel family is a property of the CONTEXT (slot count at dtk_create, dtk_set_option "mv_slots"),
  // never of the active set, so a sequence's logits do not depend on which other slots happen to decode with it.
  int mv_slots = 4;                  // contexts with nb <= mv_slots + 1 use the family (0 = never)
  int mv_step = 0;                   // vectors of the step being launched / captured (0 = an MFMA step)
  int mv_tail_threads = 512;         // block of k_attn_tail_b in a multi-vector step (few blocks: more rows per round trip)
  dtk_sampling sampling{};
  SampleMB* smb = nullptr;           // multi-block sampler scratch (single sequence) / per slot
  SampleMB* smb_b = nullptr;
  bool mb_single = false;            // the captured single-sequence graph uses the multi-block sampler
  bool mb_batch = false;             // ... the batched graph
  bool slot_topk[DTK_MAX_SLOTS] = {};   // slots whose sampling needs top-k (single-block sampler only)
  bool slot_samples[DTK_MAX_SLOTS] = {}; // slots that sample (not greedy)
  uint64_t launched = 0, waited = 0;
  hipEvent_t step_done[DTK_MAX_INFLIGHT] = {};
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
  // dtk_vit_encode (the SelfSim reward's ViT passes) runs on its own stream so it overlaps with decode steps of other
  // sequences; the caller serialises it with prefills (they share the ViT activation buffers), see model/modeling.py
  hipStream_t stream_vit = nullptr;
  hipStream_t cur_stream = nullptr;   // stream of the GEMMs being issued (vit_forward sets it)
  hipEvent_t ev_va = nullptr, ev_vb = nullptr;
  hipEvent_t probe_a = nullptr, probe_b = nullptr;
  bool use_graph = true;
  bool graph_ready = false;
  hipGraph_t graph = nullptr, graph_short = nullptr;
  hipGraphExec_t graph_exec = nullptr, graph_short_exec = nullptr;
  int attn_full_max = 0;             // contexts below this use the one-block-per-head attention (measured slower: off)
  int attn_threads = 0;              // decode attention: 0 = k_attn_decode (contiguous key range per split), 256 | 512 | 1024 = k_attn_decode_t
  int attn_impl = 0;                 // prefill / ViT attention kernel: 0 auto, 1 VALU, 2 MFMA flash (dtk_set_option "attn_impl")
  bool gemm_naive = false;
  int probe = 0;
  bool probe_pending = false;
  dtk_stats stats{};
};

namespace {

int fail(dtk_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) {
    { std::lock_guard<std::mutex> g(c->err_mu); c->err = buf; }
    g_thread_error = buf;
    g_thread_error_ctx = c;
  } else {
    g_create_error = buf;
  }
  return code;
}

#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), DTK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

uint16_t host_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float host_bf2f(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
float host_h2f(uint16_t h) {
  const uint32_t sign = (h >> 15) & 1u, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  float v;
  if (exp == 0) v = ldexpf((float)man, -24);
  else if (exp == 31) v = man ? NAN : INFINITY;
  else v = ldexpf((float)(man | 0x400u), (int)exp - 25);
  return sign ? -v : v;
}

// ---- arena planning: first pass sums sizes, second pass hands out pointers
struct Planner {
  size_t off = 0;
  unsigned char* base = nullptr;
  template <typename T>
  T* take(size_t n_elems) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n_elems * sizeof(T);
    return p;
  }
};

void add_tensor(dtk_ctx* c, const std::string& name, bf16_t* ptr, int64_t rows, int64_t cols,
                int64_t stride, float scale, float offset) {
  TensorEntry t;
  t.name = name; t.ptr = ptr; t.rows = rows; t.cols = cols; t.stride = stride;
  t.synth_scale = scale; t.synth_offset = offset;
  c->tindex[name] = (int)c->tensors.size();
  c->tensors.push_back(t);
}

// Lays out every device buffer.  Called twice (size pass with base == nullptr, then for real).
void plan(dtk_ctx* c, Planner& P, bool reg) {
  const int d = c->d, L = c->L, ff = c->ff, V = c->V, T = c->Tmax;
  const int kvd = c->KVH * 128, qkvn = d + 2 * kvd;   // fused QKV rows: [H q heads | KVH k heads | KVH v heads] x 128
  const int D = c->vD, N = c->vN, mlp = c->vMlp;
  const float ws = 0.02f;
  auto R = [&](const std::string& n, bf16_t* p, int64_t r, int64_t cl, int64_t st, float sc,
               float of) { if (reg) add_tensor(c, n, p, r, cl, st, sc, of); };
  // ---- decoder weights
  c->embed = P.take<bf16_t>((size_t)V * d);
  R("model.embed_tokens.weight", c->embed, V, d, d, ws, 0.f);
  if (reg) c->layers.resize(L);
  for (int i = 0; i < L; ++i) {
    LayerW w;
    w.ln1 = P.take<bf16_t>(d);
    w.wqkv = P.take<bf16_t>((size_t)qkvn * d);
    w.wo = P.take<bf16_t>((size_t)d * d);
    w.ln2 = P.take<bf16_t>(d);
    w.wgu = P.take<bf16_t>((size_t)2 * ff * d);
    w.wdown = P.take<bf16_t>((size_t)d * ff);
    if (reg) {
      c->layers[i] = w;
      const std::string p = "model.layers." + std::to_string(i) + ".";
      R(p + "input_layernorm.weight", w.ln1, 1, d, d, 0.1f, 1.f);
      R(p + "self_attn.q_proj.weight", w.wqkv, d, d, d, ws, 0.f);
      R(p + "self_attn.k_proj.weight", w.wqkv + (size_t)d * d, kvd, d, d, ws, 0.f);
      R(p + "self_attn.v_proj.weight", w.wqkv + (size_t)(d + kvd) * d, kvd, d, d, ws, 0.f);
      R(p + "self_attn.o_proj.weight", w.wo, d, d, d, ws, 0.f);
      R(p + "post_attention_layernorm.weight", w.ln2, 1, d, d, 0.1f, 1.f);
      R(p + "mlp.gate_proj.weight", w.wgu, ff, d, d, ws, 0.f);
      R(p + "mlp.up_proj.weight", w.wgu + (size_t)ff * d, ff, d, d, ws, 0.f);
      R(p + "mlp.down_proj.weight", w.wdown, d, ff, ff, ws, 0.f);
    }
  }
  c->final_norm = P.take<bf16_t>(d);
  R("model.norm.weight", c->final_norm, 1, d, d, 0.1f, 1.f);
  c->lm_head = P.take<bf16_t>((size_t)V * d);
  R("lm_head.weight", c->lm_head, V, d, d, ws, 0.f);
  c->mm_w = P.take<bf16_t>((size_t)d * c->cfg.concat_patches * D);
  c->mm_b = P.take<bf16_t>(d);
  R("model.mm_projector.weight", c->mm_w, d, (int64_t)c->cfg.concat_patches * D,
    (int64_t)c->cfg.concat_patches * D, ws, 0.f);
  if (c->proj_bias) R("model.mm_projector.bias", c->mm_b, 1, d, d, 0.01f, 0.f);   // v2 connector is bias-free
  c->rope_cos = P.take<bf16_t>((size_t)T * 64);
  c->rope_sin = P.take<bf16_t>((size_t)T * 64);
  R("rope.cos", c->rope_cos, T, 64, 64, 0.f, 0.f);
  R("rope.sin", c->rope_sin, T, 64, 64, 0.f, 0.f);
  // ---- vision tower (timm VisionTransformer state-dict names)
  const std::string vp = "vision_model.";
  c->pe_w = P.take<bf16_t>((size_t)D * c->vPatchLd);
  c->pe_b = P.take<bf16_t>(D);
  c->pos_embed = P.take<bf16_t>((size_t)N * D);
  R(vp + "patch_embed.proj.weight", c->pe_w, D, c->vPatchK, c->vPatchLd, ws, 0.f);
  R(vp + "patch_embed.proj.bias", c->pe_b, 1, D, D, 0.01f, 0.f);
  R(vp + "pos_embed", c->pos_embed, N, D, D, ws, 0.f);
  if (reg) c->vblocks.resize(c->vDepth);
  for (int i = 0; i < c->vDepth; ++i) {
    VitBlockW w;
    w.n1w = P.take<bf16_t>(D); w.n1b = P.take<bf16_t>(D);
    w.qkvw = P.take<bf16_t>((size_t)3 * D * D); w.qkvb = P.take<bf16_t>(3 * D);
    w.projw = P.take<bf16_t>((size_t)D * D); w.projb = P.take<bf16_t>(D);
    w.n2w = P.take<bf16_t>(D); w.n2b = P.take<bf16_t>(D);
    w.fc1w = P.take<bf16_t>((size_t)mlp * D); w.fc1b = P.take<bf16_t>(mlp);
    w.fc2w = P.take<bf16_t>((size_t)D * mlp); w.fc2b = P.take<bf16_t>(D);
    if (reg) {
      c->vblocks[i] = w;
      const std::string p = vp + "blocks." + std::to_string(i) + ".";
      R(p + "norm1.weight", w.n1w, 1, D, D, 0.1f, 1.f);
      R(p + "norm1.bias", w.n1b, 1, D, D, 0.01f, 0.f);
      R(p + "attn.qkv.weight", w.qkvw, 3 * D, D, D, ws, 0.f);
      R(p + "attn.qkv.bias", w.qkvb, 1, 3 * D, 3 * D, 0.01f, 0.f);
      R(p + "attn.proj.weight", w.projw, D, D, D, ws, 0.f);
      R(p + "attn.proj.bias", w.projb, 1, D, D, 0.01f, 0.f);
      R(p + "norm2.weight", w.n2w, 1, D, D, 0.1f, 1.f);
      R(p + "norm2.bias", w.n2b, 1, D, D, 0.01f, 0.f);
      R(p + "mlp.fc1.weight", w.fc1w, mlp, D, D, ws, 0.f);
      R(p + "mlp.fc1.bias", w.fc1b, 1, mlp, mlp, 0.01f, 0.f);
      R(p + "mlp.fc2.weight", w.fc2w, D, mlp, mlp, ws, 0.f);
      R(p + "mlp.fc2.bias", w.fc2b, 1, D, D, 0.01f, 0.f);
    }
  }
  c->vnorm_w = P.take<bf16_t>(D); c->vnorm_b = P.take<bf16_t>(D);
  R(vp + "norm.weight", c->vnorm_w, 1, D, D, 0.1f, 1.f);
  R(vp + "norm.bias", c->vnorm_b, 1, D, D, 0.01f, 0.f);
  c->ap_latent = P.take<bf16_t>(D);
  c->ap_qw = P.take<bf16_t>((size_t)D * D); c->ap_qb = P.take<bf16_t>(D);
  c->ap_kvw = P.take<bf16_t>((size_t)2 * D * D); c->ap_kvb = P.take<bf16_t>(2 * D);
  c->ap_pw = P.take<bf16_t>((size_t)D * D); c->ap_pb = P.take<bf16_t>(D);
  c->ap_nw = P.take<bf16_t>(D); c->ap_nb = P.take<bf16_t>(D);
  c->ap_f1w = P.take<bf16_t>((size_t)mlp * D); c->ap_f1b = P.take<bf16_t>(mlp);
  c->ap_f2w = P.take<bf16_t>((size_t)D * mlp); c->ap_f2b = P.take<bf16_t>(D);
  R(vp + "attn_pool.latent", c->ap_latent, 1, D, D, ws, 0.f);
  R(vp + "attn_pool.q.weight", c->ap_qw, D, D, D, ws, 0.f);
  R(vp + "attn_pool.q.bias", c->ap_qb, 1, D, D, 0.01f, 0.f);
  R(vp + "attn_pool.kv.weight", c->ap_kvw, 2 * D, D, D, ws, 0.f);
  R(vp + "attn_pool.kv.bias", c->ap_kvb, 1, 2 * D, 2 * D, 0.01f, 0.f);
  R(vp + "attn_pool.proj.weight", c->ap_pw, D, D, D, ws, 0.f);
  R(vp + "attn_pool.proj.bias", c->ap_pb, 1, D, D, 0.01f, 0.f);
  R(vp + "attn_pool.norm.weight", c->ap_nw, 1, D, D, 0.1f, 1.f);
  R(vp + "attn_pool.norm.bias", c->ap_nb, 1, D, D, 0.01f, 0.f);
  R(vp + "attn_pool.mlp.fc1.weight", c->ap_f1w, mlp, D, D, ws, 0.f);
  R(vp + "attn_pool.mlp.fc1.bias", c->ap_f1b, 1, mlp, mlp, 0.01f, 0.f);
  R(vp + "attn_pool.mlp.fc2.weight", c->ap_f2w, D, mlp, mlp, ws, 0.f);
  R(vp + "attn_pool.mlp.fc2.bias", c->ap_f2b, 1, D, D, 0.01f, 0.f);

  // ---- KV cache + activations
  c->kv = P.take<bf16_t>((size_t)L * 2 * c->KVH * T * 128);
  c->X = P.take<bf16_t>((size_t)T * d);
  c->Xn = P.take<bf16_t>((size_t)T * d);
  c->QKV = P.take<bf16_t>((size_t)T * qkvn);
  c->Qh = P.take<bf16_t>((size_t)T * d);
  c->AO = P.take<bf16_t>((size_t)T * d);
  c->GU = P.take<bf16_t>((size_t)T * 2 * ff);
  c->ACT = P.take<bf16_t>((size_t)T * ff);
  {
    size_t need = 0;
    const int roles[4][2] = {{qkvn, d}, {d, d}, {2 * ff, d}, {d, ff}};
    for (auto& r : roles) need = std::max(need, (size_t)sk_role_slices(r[0], r[1]) * SK_CHUNK_ROWS * (size_t)r[0]);
    c->skpart_floats = need;
    c->skpart = P.take<float>(need);
  }
  c->ids_dev = P.take<int32_t>(T);
  c->x = P.take<bf16_t>(d);
  c->q = P.take<bf16_t>(d);
  c->act = P.take<bf16_t>(ff);
  c->logits = P.take<float>(V);
  c->pm = P.take<float>((size_t)c->H * 16);          // sized for the largest split factor (dtk_set_option "attn_splits")
  c->pl = P.take<float>((size_t)c->H * 16);
  c->po = P.take<float>((size_t)c->H * 16 * 130);
  c->attn_out = P.take<bf16_t>(d);
  c->attn_ctr = P.take<unsigned>(c->H);
  c->st = P.take<DecState>(1);
  c->sp = P.take<SamplingDev>(1);
  c->smb = P.take<SampleMB>(1);
  c->tok_ring_dev = P.take<int64_t>(DTK_MAX_INFLIGHT);
  c->probs_dev = P.take<float>(V);
  // the tower's activations hold DTK_VIT_BATCH images: dtk_vit_encode(batch) runs them as ONE pass (GEMM rows = images x
  // patches: the M = 729 GEMMs of a single image leave the matrix cores 93 % idle), the prefill uses the first image's share
  const size_t VB = DTK_VIT_BATCH;
  c->pixels_dev = P.take<float>(VB * 3 * c->cfg.vit_image * c->cfg.vit_image);
  c->patches = P.take<bf16_t>(VB * N * c->vPatchLd);
  c->VX = P.take<bf16_t>(VB * N * D);
  c->VN = P.take<bf16_t>(VB * N * D);
  c->VQKV = P.take<bf16_t>(VB * N * 3 * D);
  c->VAO = P.take<bf16_t>(VB * N * D);
  c->VH = P.take<bf16_t>(VB * N * mlp);
  c->feats = P.take<bf16_t>(VB * N * D);
  c->last_hidden = P.take<bf16_t>(VB * N * D);
  c->pq = P.take<bf16_t>(D);
  c->pkv = P.take<bf16_t>(VB * N * 2 * D);
  c->pao = P.take<bf16_t>(VB * D);
  c->px = P.take<bf16_t>(VB * D);
  c->pn = P.take<bf16_t>(VB * D);
  c->ph = P.take<bf16_t>(VB * mlp);
  c->pooled = P.take<bf16_t>(VB * D);
  c->IMG = P.take<bf16_t>((size_t)c->nImg * d);
  if (c->wfmt == 1) {
    for (int i = 0; i < L; ++i) {
      uint8_t* q1 = P.take<uint8_t>((size_t)qkvn * d); float* s1 = P.take<float>(qkvn);
      uint8_t* q2 = P.take<uint8_t>((size_t)d * d);     float* s2 = P.take<float>(d);
      uint8_t* q3 = P.take<uint8_t>((size_t)2 * ff * d); float* s3 = P.take<float>(2 * ff);
      uint8_t* q4 = P.take<uint8_t>((size_t)d * ff);    float* s4 = P.take<float>(d);
      if (reg) {
        LayerW& w = c->layers[i];
        w.q_wqkv = q1; w.s_wqkv = s1; w.q_wo = q2; w.s_wo = s2; w.q_wgu = q3; w.s_wgu = s3; w.q_wdown = q4; w.s_wdown = s4;
      }
    }
    c->q_lm_head = P.take<uint8_t>((size_t)V * d);
    c->s_lm_head = P.take<float>(V);
  }
  if (c->nb > 0) {
    c->kv_slot_stride = (size_t)L * 2 * c->KVH * T * 128;
    c->kvb = P.take<bf16_t>((size_t)c->nb * c->kv_slot_stride);
    c->xb = P.take<bf16_t>((size_t)DTK_MAX_BATCH * d);
    // xnb / aob / actb are fragment-major GEMV inputs (common.h xtile_off): K rounded up to whole 32-wide k-steps
    c->xnb = P.take<bf16_t>((size_t)DTK_MAX_BATCH * align_up((size_t)(d > ff ? d : ff), 32));
    c->qb = P.take<bf16_t>((size_t)DTK_MAX_BATCH * d);
    c->aob = P.take<bf16_t>((size_t)DTK_MAX_BATCH * align_up((size_t)d, 32));
    c->actb = P.take<bf16_t>((size_t)DTK_MAX_BATCH * align_up((size_t)ff, 32));
    c->logits_b = P.take<float>((size_t)c->nb * V);
    c->pfx_m = P.take<float>((size_t)DTK_MAX_BATCH * c->H * 4);
    c->pfx_l = P.take<float>((size_t)DTK_MAX_BATCH * c->H * 4);
    c->pfx_o = P.take<float>((size_t)DTK_MAX_BATCH * c->H * 4 * 128);
    c->kpart = P.take<float>((size_t)8 * ((size_t)(d + 15) / 16) * 4 * 256);   // k_gemv_bk: 8 K-slice partials of every row tile x 64 slots
    c->kctr = P.take<unsigned>((size_t)(d + 15) / 16);                              // arrival counters (the arena is zeroed once; the last arrival resets)
    c->st_b = P.take<DecState>(DTK_MAX_SLOTS);
    c->sp_b = P.take<SamplingDev>(DTK_MAX_SLOTS);
    c->smb_b = P.take<SampleMB>(DTK_MAX_SLOTS);
    c->bs_dev = P.take<BatchState>(1);
    if (c->wfmt == 1) {             // fp8: pair-tiled fp8 copies (+6.6 GB for cl-7b), no bf16 tiles
      for (int i = 0; i < L; ++i) {
        uint8_t* a1 = P.take<uint8_t>(tiled_bytes_f8(qkvn, d));
        uint8_t* a2 = P.take<uint8_t>(tiled_bytes_f8(d, d));
        uint8_t* a3 = P.take<uint8_t>(tiled_bytes_f8(2 * ff, d));
        uint8_t* a4 = P.take<uint8_t>(tiled_bytes_f8(d, ff));
        if (reg) { c->layers[i].t8_wqkv = a1; c->layers[i].t8_wo = a2; c->layers[i].t8_wgu = a3; c->layers[i].t8_wdown = a4; }
      }
      c->t8_lm_head = P.take<uint8_t>(tiled_bytes_f8(V, d));
      // (MHA models only: the MXFP8 epilogue of the GQA-fused attention instantiations has not run on hardware yet — a v2 model with fp8
      // weights keeps the bf16-activation kernels)
      c->mx_ok = c->KVH == c->H && mx_unit_covers(d) && mx_kparts_covers(d, d, 32) && mx_kparts_covers(d, ff, 16) && (ff % 64) == 0 && (qkvn % 32) == 0;
      if (c->mx_ok) {               // + one more fp8 copy in MX tile order, the slots' MXFP8 vectors and their scales
        for (int i = 0; i < L; ++i) {
          uint8_t* a1 = P.take<uint8_t>(mx_w_bytes(qkvn, d, 32));
          uint8_t* a2 = P.take<uint8_t>(mx_w_bytes(d, d, 32));
          uint8_t* a3 = P.take<uint8_t>(mx_w_bytes(2 * ff, d, 32));
          uint8_t* a4 = P.take<uint8_t>(mx_w_bytes(d, ff, 16));
          if (reg) { c->layers[i].m_wqkv = a1; c->layers[i].m_wo = a2; c->layers[i].m_wgu = a3; c->layers[i].m_wdown = a4; }
        }
        c->m_lm_head = P.take<uint8_t>(mx_w_bytes(V, d, 32));
        c->xn8 = P.take<uint8_t>(mx_x_bytes(d, 32)); c->xns = P.take<uint8_t>(mx_s_bytes(d, 32));
        c->ao8 = P.take<uint8_t>(mx_x_bytes(d, 32)); c->aos = P.take<uint8_t>(mx_s_bytes(d, 32));
        c->act8 = P.take<uint8_t>(mx_x_bytes(ff, 16)); c->acts = P.take<uint8_t>(mx_s_bytes(ff, 16));
      }
    } else {
      for (int i = 0; i < L; ++i) {   // fragment-major copies of the decoder weights (288 GB HBM: +13 GB for ds-7b)
        bf16_t* a1 = P.take<bf16_t>(tiled_elems(qkvn, d));
        bf16_t* a2 = P.take<bf16_t>(tiled_elems(d, d));
        bf16_t* a3 = P.take<bf16_t>(tiled_elems(2 * ff, d));
        bf16_t* a4 = P.take<bf16_t>(tiled_elems(d, ff));
        if (reg) { c->layers[i].t_wqkv = a1; c->layers[i].t_wo = a2; c->layers[i].t_wgu = a3; c->layers[i].t_wdown = a4; }
      }
      c->t_lm_head = P.take<bf16_t>(tiled_elems(V, d));
    }
    c->tokb_dev = P.take<int64_t>((size_t)DTK_MAX_INFLIGHT * DTK_MAX_BATCH + 1);   // + the sticky device error word (TOKB_ERR)
  }
  for (int i = 0; i < L; ++i) {     // the prefill GEMMs' fragment-major weights: the batched step's copies where a bf16 context has them
    if (sk_role_slices(2 * ff, d) == 1) { bf16_t* ai = P.take<bf16_t>(tiled_elems(2 * ff, d)); if (reg) c->layers[i].p_wgui = ai; }
    if (c->nb > 0 && c->wfmt != 1) { if (reg) { LayerW& w = c->layers[i]; w.p_wqkv = w.t_wqkv; w.p_wo = w.t_wo; w.p_wgu = w.t_wgu; w.p_wdown = w.t_wdown; } continue; }
    bf16_t* a1 = P.take<bf16_t>(tiled_elems(qkvn, d));
    bf16_t* a2 = P.take<bf16_t>(tiled_elems(d, d));
    bf16_t* a3 = P.take<bf16_t>(tiled_elems(2 * ff, d));
    bf16_t* a4 = P.take<bf16_t>(tiled_elems(d, ff));
    if (reg) { LayerW& w = c->layers[i]; w.p_wqkv = a1; w.p_wo = a2; w.p_wgu = a3; w.p_wdown = a4; }
  }
  c->scratch_bytes = (size_t)64 << 20;
  c->scratch = P.take<unsigned char>(c->scratch_bytes);
}

void gemm(dtk_ctx* c, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
          const bf16_t* res, int ldr, bf16_t* C, int ldc, int M, int N, int K, int flags) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.residual = res; g.ldr = ldr;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.flags = flags;
  hipStream_t s = c->cur_stream ? c->cur_stream : c->stream;
  if (c->gemm_naive) launch_gemm_naive(g, s);
  else launch_gemm_mfma(g, s);
}

// A sliced-K role (g.kslices > 1) at g.M rows: up to SK_SL_MIN_ROWS rows as (tile, slice) blocks + the reduction, in chunks of SK_CHUNK_ROWS
// rows whose partials stay inside the Infinity Cache; above, where the tiles alone fill the chip, as one launch that folds the slices in
// registers.  Same arithmetic per row either way (tested), so a row never depends on how many rows travel with it.
bool gemm_sliced(dtk_ctx* c, const GemmArgs& g, float* part, size_t part_floats, const bf16_t* norm_w, bf16_t* Y, int ldy, hipStream_t s) {
  if (!gemm_sk_supported(g) || (size_t)g.kslices * SK_CHUNK_ROWS * (size_t)g.N > part_floats) return false;
  if (g.M > c->sk_sl_min_rows && launch_gemm_g3_sliced(g, s)) {
    if (norm_w) launch_rmsnorm_rows_sk(g.C, g.ldc, norm_w, Y, ldy, g.M, g.N, c->cfg.rms_eps, s);
    return true;
  }
  for (int m0 = 0; m0 < g.M; m0 += SK_CHUNK_ROWS) {
    GemmArgs h = g;
    h.M = std::min(SK_CHUNK_ROWS, g.M - m0);
    h.A = g.A + (size_t)m0 * g.lda; h.C = g.C + (size_t)m0 * g.ldc; h.residual = g.residual ? g.residual + (size_t)m0 * g.ldr : nullptr;
    h.part = part; h.part_stride = (long)SK_CHUNK_ROWS * g.N;
    if (!launch_gemm_sk(h, norm_w, norm_w ? Y + (size_t)m0 * ldy : nullptr, ldy, c->cfg.rms_eps, s)) return false;
  }
  return true;
}

// One decoder-prefill Linear (+ residual) and, when norm_w is given, the RMSNorm that follows it (-> Y).  Roles whose weight shape gives the
// 256 x 128 tile fewer than 128 blocks run as sliced-K GEMMs (kernels_batched.hip: launch_gemm_sk) — the slice count is a function of the
// WEIGHT shape alone, so a row's arithmetic does not depend on how many rows are prefilled with it (tail prefill == full prefill).
void gemm_role(dtk_ctx* c, const bf16_t* A, int lda, const bf16_t* W, const bf16_t* Wt, int ldw, const bf16_t* res, int ldr, bf16_t* C, int ldc,
               int M, int N, int K, int flags, const bf16_t* norm_w, bf16_t* Y, int ldy) {
  hipStream_t s = c->cur_stream ? c->cur_stream : c->stream;
  const int S = c->prefill_sk ? std::min(sk_role_slices(N, K), c->prefill_sk == 1 ? 8 : c->prefill_sk) : 1;
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = nullptr; g.residual = res; g.ldr = ldr;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.flags = flags; g.kslices = S; g.Wt = Wt;
  bool done = false;
  if (S > 1 && c->gemm_naive) { launch_gemm_naive(g, s); done = true; }
  else if (S > 1) {
    if (!gemm_sliced(c, g, c->skpart, c->skpart_floats, norm_w, Y, ldy, s)) c->launch_refused = true;
    return;
  }    // a sliced role has one canonical order: no silent change of kernel
  if (!done) { g.kslices = 1; if (c->gemm_naive) launch_gemm_naive(g, s); else launch_gemm_mfma(g, s); }
  if (norm_w) launch_rmsnorm_rows(C, ldc, norm_w, Y, ldy, M, N, c->cfg.rms_eps, s);
}

// q/k/v of n <= SK_CHUNK_ROWS prefill rows as a sliced-K GEMM whose reduction is fused with RoPE + the KV append (no [n][qkvn] buffer);
// false = the role is not sliced here (the caller runs Linear + k_rope_scatter).  Same values as that pair, bit for bit.
bool qkv_rope_fused(dtk_ctx* c, const LayerW& w, int n, int start, bf16_t* kc, bf16_t* vc, hipStream_t s) {
  const int d = c->d, qkvn = d + 2 * c->KVH * 128;
  const int S = c->prefill_sk ? std::min(sk_role_slices(qkvn, d), c->prefill_sk == 1 ? 8 : c->prefill_sk) : 1;
  if (S <= 1 || c->gemm_naive || !c->qkv_rope_fused || n > SK_CHUNK_ROWS || n > c->sk_sl_min_rows || (size_t)S * SK_CHUNK_ROWS * (size_t)qkvn > c->skpart_floats) return false;
  GemmArgs g;
  g.A = c->Xn; g.lda = d; g.W = w.wqkv; g.Wt = w.p_wqkv; g.ldw = d; g.bias = nullptr; g.residual = nullptr; g.ldr = 0;
  g.C = c->QKV; g.ldc = qkvn; g.M = n; g.N = qkvn; g.K = d; g.flags = 0; g.kslices = S;
  g.part = c->skpart; g.part_stride = (long)SK_CHUNK_ROWS * qkvn;
  if (!gemm_sk_supported(g)) return false;
  if (!launch_gemm_sk_partials(g, s)) { c->launch_refused = true; return true; }
  launch_sk_rope_scatter(g.part, g.part_stride, S, c->Qh, kc, vc, c->rope_cos, c->rope_sin, n, start, c->H, c->KVH, c->Tmax, s);
  return true;
}

int gelu_flag(const dtk_ctx* c) { return c->cfg.vit_gelu_tanh ? GEMM_GELU_TANH : GEMM_GELU_ERF; }

bf16_t* kcache(dtk_ctx* c, int layer) { return c->kv + (size_t)layer * 2 * c->KVH * c->Tmax * 128; }
bf16_t* vcache(dtk_ctx* c, int layer) { return kcache(c, layer) + (size_t)c->KVH * c->Tmax * 128; }

// ViT trunk + (optionally) MAP head for the image already in pixels_dev.
void vit_forward(dtk_ctx* c, bool want_pooled, hipStream_t s, int B = 1) {
  // B images (<= DTK_VIT_BATCH) in one pass: every row-wise op (LayerNorm, the Linear layers) sees B x N rows, attention and the
  // position-embedding add run per image.  Per row the arithmetic is the single-image arithmetic (a GEMM row does not depend on
  // the other rows of its tile), so image b's features are bit-identical to encoding it alone (tested).
  const int D = c->vD, N = c->vN, mlp = c->vMlp, Hh = c->vH, hd = c->vHd;
  const int R = B * N;
  struct StreamScope { dtk_ctx* c; hipStream_t prev; ~StreamScope() { c->cur_stream = prev; } } scope{c, c->cur_stream};
  c->cur_stream = s;
  const size_t img = (size_t)3 * c->cfg.vit_image * c->cfg.vit_image;
  for (int b = 0; b < B; ++b) {
    launch_im2col(c->pixels_dev + (size_t)b * img, c->patches + (size_t)b * N * c->vPatchLd, c->cfg.vit_image, c->cfg.vit_patch, c->vPatchLd, s);
    // conv(patch)+bias -> bf16, then + pos_embed -> bf16 (timm PatchEmbed, _pos_embed): the residual operand is per patch
    gemm(c, c->patches + (size_t)b * N * c->vPatchLd, c->vPatchLd, c->pe_w, c->vPatchLd, c->pe_b, c->pos_embed, D,
         c->VX + (size_t)b * N * D, D, N, D, c->vPatchLd, GEMM_BIAS | GEMM_RESIDUAL);
  }
  const float scale = 1.0f / sqrtf((float)hd);
  const int fl = c->cfg.vit_feature_layer;
  const int last = want_pooled ? c->vDepth - 1 : fl;
  for (int i = 0; i <= last; ++i) {
    const VitBlockW& w = c->vblocks[i];
    launch_layernorm_rows(c->VX, D, w.n1w, w.n1b, c->VN, D, R, D, c->cfg.vit_ln_eps, s);
    gemm(c, c->VN, D, w.qkvw, D, w.qkvb, nullptr, 0, c->VQKV, 3 * D, R, 3 * D, D, GEMM_BIAS);
    {   // the B images' attention problems in ONE launch (grid z = image): 8 launches of 192 blocks each left the chip a quarter empty
      const bf16_t* qkv = c->VQKV;
      AttnArgs a;
      a.Q = qkv; a.q_sh = hd; a.q_st = 3 * D;
      a.K = qkv + D; a.k_sh = hd; a.k_st = 3 * D;
      a.V = qkv + 2 * D; a.v_sh = hd; a.v_st = 3 * D;
      a.O = c->VAO; a.o_sh = hd; a.o_st = D;
      a.H = Hh; a.Tq = N; a.Tk = N; a.hd = hd; a.causal = 0; a.q_offset = 0; a.scale = scale; a.impl = c->attn_impl; a.kv_group = 1;
      a.nbatch = B; a.q_sb = a.k_sb = a.v_sb = (long)N * 3 * D; a.o_sb = (long)N * D;
      launch_attention(a, s);
    }
    gemm(c, c->VAO, D, w.projw, D, w.projb, c->VX, D, c->VX, D, R, D, D, GEMM_BIAS | GEMM_RESIDUAL);
    launch_layernorm_rows(c->VX, D, w.n2w, w.n2b, c->VN, D, R, D, c->cfg.vit_ln_eps, s);
    gemm(c, c->VN, D, w.fc1w, D, w.fc1b, nullptr, 0, c->VH, mlp, R, mlp, D, GEMM_BIAS | gelu_flag(c));
    gemm(c, c->VH, mlp, w.fc2w, mlp, w.fc2b, c->VX, D, c->VX, D, R, D, mlp, GEMM_BIAS | GEMM_RESIDUAL);
    if (i == fl)  // get_intermediate_layers(n=[layer], norm=True)
      launch_layernorm_rows(c->VX, D, c->vnorm_w, c->vnorm_b, c->feats, D, R, D, c->cfg.vit_ln_eps, s);
  }
  if (!want_pooled) return;
  // forward_features -> final norm; forward_head -> AttentionPoolLatent ('map')
  const bf16_t* lh = c->feats;
  if (fl != c->vDepth - 1) {
    launch_layernorm_rows(c->VX, D, c->vnorm_w, c->vnorm_b, c->last_hidden, D, R, D, c->cfg.vit_ln_eps, s);
    lh = c->last_hidden;
  }
  gemm(c, c->ap_latent, D, c->ap_qw, D, c->ap_qb, nullptr, 0, c->pq, D, 1, D, D, GEMM_BIAS);
  gemm(c, lh, D, c->ap_kvw, D, c->ap_kvb, nullptr, 0, c->pkv, 2 * D, R, 2 * D, D, GEMM_BIAS);
  {
    AttnArgs a;
    a.Q = c->pq; a.q_sh = hd; a.q_st = D;
    a.K = c->pkv; a.k_sh = hd; a.k_st = 2 * D;
    a.V = c->pkv + D; a.v_sh = hd; a.v_st = 2 * D;
    a.O = c->pao; a.o_sh = hd; a.o_st = D;
    a.H = Hh; a.Tq = 1; a.Tk = N; a.hd = hd; a.causal = 0; a.q_offset = 0; a.scale = scale; a.impl = c->attn_impl; a.kv_group = 1;
    a.nbatch = B; a.q_sb = 0; a.k_sb = a.v_sb = (long)N * 2 * D; a.o_sb = D;      // the latent query is the same for every image
    launch_attention(a, s);
  }
  gemm(c, c->pao, D, c->ap_pw, D, c->ap_pb, nullptr, 0, c->px, D, B, D, D, GEMM_BIAS);
  launch_layernorm_rows(c->px, D, c->ap_nw, c->ap_nb, c->pn, D, B, D, c->cfg.vit_ln_eps, s);
  gemm(c, c->pn, D, c->ap_f1w, D, c->ap_f1b, nullptr, 0, c->ph, mlp, B, mlp, D, GEMM_BIAS | gelu_flag(c));
  gemm(c, c->ph, mlp, c->ap_f2w, mlp, c->ap_f2b, c->px, D, c->pooled, D, B, D, mlp, GEMM_BIAS | GEMM_RESIDUAL);
}

void project_image(dtk_ctx* c) {
  // feats [N][D] viewed as [N/concat][concat*D] (3 consecutive patch tokens), Linear with bias
  const int K = c->cfg.concat_patches * c->vD;
  gemm(c, c->feats, K, c->mm_w, K, c->proj_bias ? c->mm_b : nullptr, nullptr, 0, c->IMG, c->d, c->nImg, c->d, K,
       c->proj_bias ? GEMM_BIAS : 0);
}

// launches of one decoded token (captured into the graph, or issued directly)
// tokb_dev / tokb_host: the token ring of the batched step + one word the LDS-ring kernels count expired hand-off waits in (it
// travels to the host with every step's tokens; dtk_decode_batch_wait fails once it is non-zero)
#define TOKB_ERR ((size_t)DTK_MAX_INFLIGHT * DTK_MAX_BATCH)
#define TOKB_WORDS (TOKB_ERR + 1)
static inline unsigned* batch_err_word(const dtk_ctx* c) { return reinterpret_cast<unsigned*>(c->tokb_dev + TOKB_ERR); }

void decode_step_launches(dtk_ctx* c, bool with_probe, bool short_ctx = false) {
  hipStream_t s = c->stream;
  SampleArgs sa;
  sa.logits = c->logits; sa.V = c->V; sa.sp = c->sp; sa.st = c->st; sa.embed = c->embed;
  sa.x = c->x; sa.d = c->d; sa.tok_ring = c->tok_ring_dev; sa.ring = DTK_MAX_INFLIGHT;
  sa.probs_out = nullptr; sa.advance = 1; sa.step_override = -1; sa.bs = nullptr; sa.logits_stride = 0; sa.nslots = 1; sa.mb = c->smb;
  if (c->mb_single) launch_sample_mb(sa, s); else launch_sample(sa, s);
  const float scale = 1.0f / sqrtf(128.f);
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    GemvArgs g{};
    g.eps = c->cfg.rms_eps; g.st = c->st; g.T_max = c->Tmax; g.d = c->d; g.ff = c->ff; g.H = c->H; g.KVH = c->KVH;
    g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin;
    g.pm = c->pm; g.pl = c->pl; g.po = c->po; g.S = c->S;
    // 1. input_layernorm + q/k/v projections + RoPE + KV append
    g.W = w.wqkv; g.W8 = w.q_wqkv; g.wscale = w.s_wqkv; g.N = c->d + 2 * c->KVH * 128; g.K = c->d; g.x = c->x; g.norm_w = w.ln1;
    g.q_out = c->q; g.kcache = kcache(c, l); g.vcache = vcache(c, l);
    launch_gemv(PRO_RMSNORM, EPI_QKV, g, s);
    // 2. split-K attention over the cache
    AttnDecArgs ad;
    ad.q = c->q; ad.kcache = kcache(c, l); ad.vcache = vcache(c, l); ad.st = c->st;
    ad.pm = c->pm; ad.pl = c->pl; ad.po = c->po; ad.H = c->H; ad.S = c->S; ad.T_max = c->Tmax; ad.G = c->H / c->KVH;
    ad.scale = scale;
    ad.threads = c->attn_threads;
    ad.combine = (!ad.threads && short_ctx && c->attn_combine == 2) ? 3 : c->attn_combine; ad.out = c->attn_out; ad.counters = c->attn_ctr;
    if (ad.threads && ad.combine == 1) ad.combine = 2;      // the tile kernel has no in-kernel combine
    launch_attn_decode(ad, s);
    // 3. (combine +) o_proj + residual
    g.W = w.wo; g.W8 = w.q_wo; g.wscale = w.s_wo; g.N = c->d; g.K = c->d; g.y = c->x;
    const bool partials = ad.combine == 0 && !(ad.threads && ad.S == 1);   // o_proj's prologue reduces the split partials
    if (!partials) { g.x = c->attn_out; launch_gemv(PRO_COPY, EPI_RESID, g, s); }
    else launch_gemv(PRO_ATTN, EPI_RESID, g, s);
    // 4. post_attention_layernorm + gate/up + SiLU*mul
    g.W = w.wgu; g.W8 = w.q_wgu; g.wscale = w.s_wgu; g.N = 2 * c->ff; g.K = c->d; g.x = c->x; g.norm_w = w.ln2; g.y = c->act;
    const bool probe_here = with_probe && (l == c->L / 2);
    if (probe_here) (void)hipEventRecord(c->probe_a, s);
    launch_gemv(PRO_RMSNORM, EPI_SWIGLU, g, s);
    if (probe_here) (void)hipEventRecord(c->probe_b, s);
    // 5. down + residual
    g.W = w.wdown; g.W8 = w.q_wdown; g.wscale = w.s_wdown; g.N = c->d; g.K = c->ff; g.x = c->act; g.y = c->x;
    launch_gemv(PRO_COPY, EPI_RESID, g, s);
  }
  GemvArgs g{};
  g.W = c->lm_head; g.W8 = c->q_lm_head; g.wscale = c->s_lm_head; g.N = c->V; g.K = c->d; g.x = c->x; g.norm_w = c->final_norm;
  g.eps = c->cfg.rms_eps; g.logits = c->logits;
  launch_gemv(PRO_RMSNORM, EPI_LOGITS, g, s);
}

static inline bool mx_step(const dtk_ctx* c) { return c->wfmt == 1 && c->mx_ok && c->act_fp8 != 0; }

// The MFMA-family step of an fp8 model on the fp8 matrix cores (kernels_batch_mx.hip): 7 launches per layer as the 64-slot bf16
// step — q/k/v, attention, o_proj partials, reduce + residual + RMSNorm, gate/up, down partials, reduce + residual + RMSNorm —
// at every tile count; the vectors between them travel as MXFP8.
void batch_step_launches_mx(dtk_ctx* c) {
  hipStream_t s = c->stream;
  const int d = c->d, ff = c->ff, nslots = 16 * c->nt_step;
  SampleArgs sa;
  sa.logits = c->logits_b; sa.V = c->V; sa.sp = c->sp_b; sa.st = c->st_b; sa.embed = c->embed;
  sa.x = c->xb; sa.d = d; sa.tok_ring = c->tokb_dev; sa.ring = DTK_MAX_INFLIGHT;
  sa.probs_out = nullptr; sa.advance = 1; sa.step_override = -1; sa.bs = c->bs_dev; sa.logits_stride = c->V; sa.nslots = nslots; sa.mb = c->smb_b;
  if (c->mb_batch) launch_sample_mb(sa, s); else launch_sample_b(sa, s);
  const float scale = 1.0f / sqrtf(128.f);
  const size_t kv_layer = (size_t)2 * c->KVH * c->Tmax * 128;
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    bf16_t* kc = c->kvb + (size_t)l * kv_layer;
    bf16_t* vc = kc + (size_t)c->KVH * c->Tmax * 128;
    GemvBArgs g{};
    g.bs = c->bs_dev; g.st = c->st_b; g.T_max = c->Tmax; g.d = d; g.ff = ff; g.H = c->H; g.KVH = c->KVH; g.nt = c->nt_step;
    g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.kv_slot_stride = c->kv_slot_stride; g.kpart = c->kpart; g.err = batch_err_word(c);
    if (l == 0) launch_rmsnorm_b(c->xb, d, w.ln1, c->xnb, d, d, c->cfg.rms_eps, c->bs_dev, nslots, s, c->xn8, c->xns);
    g.Wm = w.m_wqkv; g.wscale = w.s_wqkv; g.N = d + 2 * c->KVH * 128; g.K = d; g.X8 = c->xn8; g.XS = c->xns; g.q_out = c->qb; g.kcache = kc; g.vcache = vc;
    launch_gemv_mxu(EPI_QKV, g, s);
    AttnDecBArgs ad;
    ad.q = c->qb; ad.kcache = kc; ad.vcache = vc; ad.kv_slot_stride = c->kv_slot_stride; ad.st = c->st_b; ad.bs = c->bs_dev;
    ad.out = c->aob; ad.H = c->H; ad.T_max = c->Tmax; ad.d = d; ad.G = c->H / c->KVH; ad.nslots = nslots;
    ad.scale = scale;
    ad.use_prefix = c->prefix_mfma; ad.pfx_splits = c->pfx_splits; ad.tail_threads = c->tail_threads; ad.gqa_fused = c->gqa_fused; ad.nt_private = c->attn_nt;
    ad.pfx_m = c->pfx_m; ad.pfx_l = c->pfx_l; ad.pfx_o = c->pfx_o;
    ad.out8 = c->ao8; ad.outs = c->aos;
    launch_attn_decode_b(ad, s);
    g.Wm = w.m_wo; g.wscale = w.s_wo; g.N = d; g.K = d; g.X8 = c->ao8; g.XS = c->aos;
    if (!launch_gemv_mxk(g, 32, s)) c->launch_refused = true;
    launch_resid_norm_b(c->kpart, c->xb, d, w.ln2, c->xnb, d, c->cfg.rms_eps, c->bs_dev, nslots, s, c->xn8, c->xns);
    g.Wm = w.m_wgu; g.wscale = w.s_wgu; g.N = 2 * ff; g.K = d; g.X8 = c->xn8; g.XS = c->xns; g.Y8 = c->act8; g.YS = c->acts;
    launch_gemv_mxu(EPI_SWIGLU, g, s);
    g.Wm = w.m_wdown; g.wscale = w.s_wdown; g.N = d; g.K = ff; g.X8 = c->act8; g.XS = c->acts;
    if (!launch_gemv_mxk(g, 16, s)) c->launch_refused = true;
    launch_resid_norm_b(c->kpart, c->xb, d, l + 1 < c->L ? c->layers[l + 1].ln1 : c->final_norm, c->xnb, d, c->cfg.rms_eps, c->bs_dev, nslots, s, c->xn8, c->xns);
  }
  GemvBArgs g{};
  g.bs = c->bs_dev; g.st = c->st_b; g.Wm = c->m_lm_head; g.wscale = c->s_lm_head; g.N = c->V; g.K = d; g.X8 = c->xn8; g.XS = c->xns; g.logits = c->logits_b;
  g.d = d; g.ff = ff; g.nt = c->nt_step; g.H = c->H; g.KVH = c->KVH; g.err = batch_err_word(c);
  launch_gemv_mxu(EPI_LOGITS, g, s);
}

// launches of one batched decode step (all 16 slot columns; inactive slots are skipped in-kernel)
void batch_step_launches(dtk_ctx* c) {
  if (mx_step(c)) { batch_step_launches_mx(c); return; }
  hipStream_t s = c->stream;
  const int d = c->d, ff = c->ff;
  SampleArgs sa;
  sa.logits = c->logits_b; sa.V = c->V; sa.sp = c->sp_b; sa.st = c->st_b; sa.embed = c->embed;
  sa.x = c->xb; sa.d = d; sa.tok_ring = c->tokb_dev; sa.ring = DTK_MAX_INFLIGHT;
  sa.probs_out = nullptr; sa.advance = 1; sa.step_override = -1; sa.bs = c->bs_dev; sa.logits_stride = c->V; sa.nslots = 16 * c->nt_step; sa.mb = c->smb_b;
  if (c->mb_batch) launch_sample_mb(sa, s); else launch_sample_b(sa, s);
  const float scale = 1.0f / sqrtf(128.f);
  const size_t kv_layer = (size_t)2 * c->KVH * c->Tmax * 128;
  bool kparts = false;
  if (c->resid_kparts) {      // both N = d roles must be covered: the norm placement follows from it for the whole step
    GemvBArgs t{};
    t.nt = c->nt_step; t.kpart = c->kpart; t.W8 = c->layers[0].t8_wo; t.N = d; t.K = d;
    kparts = resid_kparts_covers(t);
    t.K = ff;
    kparts = kparts && resid_kparts_covers(t);
  }
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    bf16_t* kc = c->kvb + (size_t)l * kv_layer;
    bf16_t* vc = kc + (size_t)c->KVH * c->Tmax * 128;
    GemvBArgs g{};
    g.bs = c->bs_dev; g.st = c->st_b; g.T_max = c->Tmax; g.d = d; g.ff = ff; g.H = c->H; g.KVH = c->KVH; g.nt = c->nt_step;
    g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.kv_slot_stride = c->kv_slot_stride; g.kpart = c->kpart; g.kctr = c->kctr; g.err = batch_err_word(c);
    // resid_kparts (64 slots): an N = d role is k_gemv_bkp (K split over CUs, fp32 partials stored) and the RMSNorm that follows it
    // is k_resid_norm_b, which first adds the partials + the residual — so the norm of layer l > 0 has already been produced by
    // layer l - 1's down projection, and the final norm by the last layer's
    if (l == 0 || !kparts) launch_rmsnorm_b(c->xb, d, w.ln1, c->xnb, d, d, c->cfg.rms_eps, c->bs_dev, 16 * c->nt_step, s);
    g.W = w.t_wqkv; g.W8 = w.t8_wqkv; g.wscale = w.s_wqkv; g.N = d + 2 * c->KVH * 128; g.K = d; g.X = c->xnb; g.ldx = d; g.q_out = c->qb; g.kcache = kc; g.vcache = vc;
    launch_gemv_b(EPI_QKV, g, s);
    AttnDecBArgs ad;
    ad.q = c->qb; ad.kcache = kc; ad.vcache = vc; ad.kv_slot_stride = c->kv_slot_stride; ad.st = c->st_b; ad.bs = c->bs_dev;
    ad.out = c->aob; ad.H = c->H; ad.T_max = c->Tmax; ad.d = d; ad.G = c->H / c->KVH; ad.nslots = 16 * c->nt_step;
    ad.scale = scale;
    ad.use_prefix = c->prefix_mfma; ad.pfx_splits = c->pfx_splits; ad.tail_threads = c->tail_threads; ad.gqa_fused = c->gqa_fused; ad.nt_private = c->attn_nt;
    ad.pfx_m = c->pfx_m; ad.pfx_l = c->pfx_l; ad.pfx_o = c->pfx_o;
    launch_attn_decode_b(ad, s);
    g.W = w.t_wo; g.W8 = w.t8_wo; g.wscale = w.s_wo; g.N = d; g.K = d; g.X = c->aob; g.ldx = d; g.Y = c->xb; g.ldy = d;
    if (kparts) {
      launch_gemv_bkp(g, s);
      launch_resid_norm_b(c->kpart, c->xb, d, w.ln2, c->xnb, d, c->cfg.rms_eps, c->bs_dev, 16 * c->nt_step, s);
    } else {
      launch_gemv_b(EPI_RESID, g, s);
      launch_rmsnorm_b(c->xb, d, w.ln2, c->xnb, d, d, c->cfg.rms_eps, c->bs_dev, 16 * c->nt_step, s);
    }
    g.W = w.t_wgu; g.W8 = w.t8_wgu; g.wscale = w.s_wgu; g.N = 2 * ff; g.K = d; g.X = c->xnb; g.ldx = d; g.Y = c->actb; g.ldy = ff;
    launch_gemv_b(EPI_SWIGLU, g, s);
    g.W = w.t_wdown; g.W8 = w.t8_wdown; g.wscale = w.s_wdown; g.N = d; g.K = ff; g.X = c->actb; g.ldx = ff; g.Y = c->xb; g.ldy = d;
    if (kparts) {
      launch_gemv_bkp(g, s);
      launch_resid_norm_b(c->kpart, c->xb, d, l + 1 < c->L ? c->layers[l + 1].ln1 : c->final_norm, c->xnb, d, c->cfg.rms_eps, c->bs_dev, 16 * c->nt_step, s);
    } else {
      launch_gemv_b(EPI_RESID, g, s);
    }
  }
  if (!kparts) launch_rmsnorm_b(c->xb, d, c->final_norm, c->xnb, d, d, c->cfg.rms_eps, c->bs_dev, 16 * c->nt_step, s);
  GemvBArgs g{};
  g.bs = c->bs_dev; g.st = c->st_b; g.W = c->t_lm_head; g.W8 = c->t8_lm_head; g.wscale = c->s_lm_head; g.N = c->V; g.K = d; g.X = c->xnb; g.ldx = d; g.logits = c->logits_b;
  g.d = d; g.ff = ff; g.nt = c->nt_step; g.err = batch_err_word(c);
  launch_gemv_b(EPI_LOGITS, g, s);
}

static inline bool mv_family(const dtk_ctx* c) { return c->mv_slots > 0 && c->nb > 0 && c->nb <= c->mv_slots + 1 && c->nb <= 5; }
// slots that may take part in a decode step: 0..3 of a multi-vector context, else the context's column tiles
static inline int max_decode_slots(const dtk_ctx* c) {
  const int lim = mv_family(c) ? 4 : 16 * c->nt;
  return c->nb < lim ? c->nb : lim;
}

// launches of one multi-vector decode step (c->mv_step = 1, 2 or 4 vectors = slots 0..mv_step-1): five kernels per layer
void batch_step_launches_mv(dtk_ctx* c) {
  hipStream_t s = c->stream;
  const int d = c->d, ff = c->ff, NB = c->mv_step;
  SampleArgs sa;
  sa.logits = c->logits_b; sa.V = c->V; sa.sp = c->sp_b; sa.st = c->st_b; sa.embed = c->embed;
  sa.x = c->xb; sa.d = d; sa.tok_ring = c->tokb_dev; sa.ring = DTK_MAX_INFLIGHT;
  sa.probs_out = nullptr; sa.advance = 1; sa.step_override = -1; sa.bs = c->bs_dev; sa.logits_stride = c->V; sa.nslots = NB; sa.mb = c->smb_b;
  if (c->mb_batch) launch_sample_mb(sa, s); else launch_sample_b(sa, s);
  const float scale = 1.0f / sqrtf(128.f);
  const size_t kv_layer = (size_t)2 * c->KVH * c->Tmax * 128;
  GemvMvArgs g{};
  g.bs = c->bs_dev; g.st = c->st_b; g.T_max = c->Tmax; g.d = d; g.ff = ff; g.H = c->H; g.KVH = c->KVH; g.eps = c->cfg.rms_eps;
  g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.kv_slot_stride = c->kv_slot_stride;
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    bf16_t* kc = c->kvb + (size_t)l * kv_layer;
    bf16_t* vc = kc + (size_t)c->KVH * c->Tmax * 128;
    // 1. input_layernorm + q/k/v + RoPE + KV append at every slot's own position
    g.W = w.wqkv; g.W8 = w.q_wqkv; g.wscale = w.s_wqkv; g.N = d + 2 * c->KVH * 128; g.K = d; g.X = c->xb; g.ldx = d; g.norm_w = w.ln1;
    g.q_out = c->qb; g.kcache = kc; g.vcache = vc;
    launch_gemv_mv(PRO_RMSNORM, EPI_QKV, NB, g, s);
    // 2. attention: one block per (head, slot), output in fragment order
    AttnDecBArgs ad;
    ad.q = c->qb; ad.kcache = kc; ad.vcache = vc; ad.kv_slot_stride = c->kv_slot_stride; ad.st = c->st_b; ad.bs = c->bs_dev;
    ad.out = c->aob; ad.H = c->H; ad.T_max = c->Tmax; ad.d = d; ad.G = c->H / c->KVH; ad.nslots = NB;
    ad.scale = scale;
    ad.use_prefix = 0; ad.pfx_splits = c->pfx_splits; ad.tail_threads = c->mv_tail_threads; ad.gqa_fused = c->gqa_fused; ad.nt_private = c->attn_nt;
    ad.pfx_m = c->pfx_m; ad.pfx_l = c->pfx_l; ad.pfx_o = c->pfx_o;
    launch_attn_decode_b(ad, s);
    // 3. o_proj + residual
    g.W = w.wo; g.W8 = w.q_wo; g.wscale = w.s_wo; g.N = d; g.K = d; g.X = c->aob; g.Y = c->xb; g.ldy = d;
    launch_gemv_mv(PRO_COPY, EPI_RESID, NB, g, s);
    // 4. post_attention_layernorm + gate/up + SiLU * mul
    g.W = w.wgu; g.W8 = w.q_wgu; g.wscale = w.s_wgu; g.N = 2 * ff; g.K = d; g.X = c->xb; g.ldx = d; g.norm_w = w.ln2; g.Y = c->actb; g.ldy = ff;
    launch_gemv_mv(PRO_RMSNORM, EPI_SWIGLU, NB, g, s);
    // 5. down + residual
    g.W = w.wdown; g.W8 = w.q_wdown; g.wscale = w.s_wdown; g.N = d; g.K = ff; g.X = c->actb; g.Y = c->xb; g.ldy = d;
    launch_gemv_mv(PRO_COPY, EPI_RESID, NB, g, s);
  }
  g.W = c->lm_head; g.W8 = c->q_lm_head; g.wscale = c->s_lm_head; g.N = c->V; g.K = d; g.X = c->xb; g.ldx = d; g.norm_w = c->final_norm;
  g.logits = c->logits_b;
  launch_gemv_mv(PRO_RMSNORM, EPI_LOGITS, NB, g, s);
}

// fp8 mode: quantise the decoder Linear weights (per-row power-of-two scale) and overwrite the bf16 masters
// with the de-quantised values, so every consumer (prefill GEMM, tiled copy, read-back, oracle) sees the
// same effective weights as the fp8 decode kernels
void ensure_fp8_weights(dtk_ctx* c) {
  if (c->wfmt != 1 || c->fp8_ready) return;
  for (int l = 0; l < c->L; ++l) {
    LayerW& w = c->layers[l];
    launch_quant_fp8_rows(w.wqkv, w.q_wqkv, w.s_wqkv, c->d + 2 * c->KVH * 128, c->d, c->stream);
    launch_quant_fp8_rows(w.wo, w.q_wo, w.s_wo, c->d, c->d, c->stream);
    launch_quant_fp8_rows(w.wgu, w.q_wgu, w.s_wgu, 2 * c->ff, c->d, c->stream);
    launch_quant_fp8_rows(w.wdown, w.q_wdown, w.s_wdown, c->d, c->ff, c->stream);
  }
  launch_quant_fp8_rows(c->lm_head, c->q_lm_head, c->s_lm_head, c->V, c->d, c->stream);
  c->fp8_ready = true;
  c->tiled_ready = false;
  c->ptiled_ready = false;
}

void ensure_tiled_weights(dtk_ctx* c);
// fragment-major bf16 copies for the prefill GEMMs (of the de-quantised weights in fp8 mode)
void ensure_prefill_tiles(dtk_ctx* c) {
  ensure_fp8_weights(c);
  if (c->ptiled_ready) return;
  for (int l = 0; l < c->L; ++l)
    if (c->layers[l].p_wgui) launch_retile_pairs(c->layers[l].wgu, c->layers[l].p_wgui, c->ff, c->d, c->stream);
  if (c->nb > 0 && c->wfmt != 1) { ensure_tiled_weights(c); c->ptiled_ready = true; return; }
  for (int l = 0; l < c->L; ++l) {
    LayerW& w = c->layers[l];
    launch_retile(w.wqkv, w.p_wqkv, c->d + 2 * c->KVH * 128, c->d, c->stream);
    launch_retile(w.wo, w.p_wo, c->d, c->d, c->stream);
    launch_retile(w.wgu, w.p_wgu, 2 * c->ff, c->d, c->stream);
    launch_retile(w.wdown, w.p_wdown, c->d, c->ff, c->stream);
  }
  c->ptiled_ready = true;
}

void ensure_tiled_weights(dtk_ctx* c) {
  ensure_fp8_weights(c);
  if (c->tiled_ready || c->nb <= 0) return;
  if (c->wfmt == 1) {
    const int qkvn = c->d + 2 * c->KVH * 128;
    for (int l = 0; l < c->L; ++l) {
      LayerW& w = c->layers[l];
      launch_retile_f8(w.q_wqkv, w.t8_wqkv, qkvn, c->d, c->stream);
      launch_retile_f8(w.q_wo, w.t8_wo, c->d, c->d, c->stream);
      launch_retile_f8(w.q_wgu, w.t8_wgu, 2 * c->ff, c->d, c->stream);
      launch_retile_f8(w.q_wdown, w.t8_wdown, c->d, c->ff, c->stream);
    }
    launch_retile_f8(c->q_lm_head, c->t8_lm_head, c->V, c->d, c->stream);
    if (c->mx_ok) {
      for (int l = 0; l < c->L; ++l) {
        LayerW& w = c->layers[l];
        launch_retile_mx(w.q_wqkv, w.m_wqkv, qkvn, c->d, 32, c->stream);
        launch_retile_mx(w.q_wo, w.m_wo, c->d, c->d, 32, c->stream);
        launch_retile_mx(w.q_wgu, w.m_wgu, 2 * c->ff, c->d, 32, c->stream);
        launch_retile_mx(w.q_wdown, w.m_wdown, c->d, c->ff, 16, c->stream);
      }
      launch_retile_mx(c->q_lm_head, c->m_lm_head, c->V, c->d, 32, c->stream);
    }
    c->tiled_ready = true;
    return;
  }
  for (int l = 0; l < c->L; ++l) {
    LayerW& w = c->layers[l];
    launch_retile(w.wqkv, w.t_wqkv, c->d + 2 * c->KVH * 128, c->d, c->stream);
    launch_retile(w.wo, w.t_wo, c->d, c->d, c->stream);
    launch_retile(w.wgu, w.t_wgu, 2 * c->ff, c->d, c->stream);
    launch_retile(w.wdown, w.t_wdown, c->d, c->ff, c->stream);
  }
  launch_retile(c->lm_head, c->t_lm_head, c->V, c->d, c->stream);
  c->tiled_ready = true;
}

static inline int nt_index(int nt) { return nt >= 3 ? 2 : nt - 1; }
static inline int step_graph_index(const dtk_ctx* c) { return c->mv_step ? 3 + (c->mv_step >= 4 ? 2 : c->mv_step - 1) : nt_index(c->nt_step); }

void drop_batch_graphs(dtk_ctx* c) {
  for (int i = 0; i < 6; ++i) {
    if (c->bgraph_exec[i]) { (void)hipGraphExecDestroy(c->bgraph_exec[i]); c->bgraph_exec[i] = nullptr; }
    if (c->bgraph[i]) { (void)hipGraphDestroy(c->bgraph[i]); c->bgraph[i] = nullptr; }
    c->bgraph_ready[i] = false;
  }
}

void drop_graph(dtk_ctx* c) {          // the single-sequence step's two captures
  if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
  if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
  if (c->graph_short_exec) { (void)hipGraphExecDestroy(c->graph_short_exec); c->graph_short_exec = nullptr; }
  if (c->graph_short) { (void)hipGraphDestroy(c->graph_short); c->graph_short = nullptr; }
  c->graph_ready = false;
}

int ensure_batch_graph(dtk_ctx* c) {   // for c->nt_step / c->mv_step
  const int gi = step_graph_index(c);
  if (c->bgraph_ready[gi]) return DTK_OK;
  HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  c->launch_refused = false;
  if (c->mv_step) batch_step_launches_mv(c); else batch_step_launches(c);
  HIPCHK(c, hipMemcpyAsync(c->tokb_host, c->tokb_dev, sizeof(int64_t) * TOKB_WORDS,
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamEndCapture(c->stream, &c->bgraph[gi]));
  if (c->launch_refused || dtk_lds_attr_error(c->device)) {       // an incomplete step must never be replayed
    (void)hipGraphDestroy(c->bgraph[gi]); c->bgraph[gi] = nullptr;
    return fail(c, DTK_ERR_STATE, c->launch_refused ? "batched step: a projection's shape has no kernel in its family (nothing launched for it)"
                                                    : "batched step: raising a kernel's dynamic-LDS limit failed (hipFuncSetAttribute, see stderr)");
  }
  HIPCHK(c, hipGraphInstantiate(&c->bgraph_exec[gi], c->bgraph[gi], nullptr, nullptr, 0));
  c->bgraph_ready[gi] = true;
  return DTK_OK;
}

int ensure_graph(dtk_ctx* c) {
  if (c->graph_ready) return DTK_OK;
  // two captures of the same step: split-K attention (any context) and the one-block-per-head
  // attention used while the context is short; the host picks per step (it knows the position)
  for (int v = 0; v < 2; ++v) {
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    decode_step_launches(c, false, v == 1);
    HIPCHK(c, hipMemcpyAsync(c->tok_ring_host, c->tok_ring_dev, sizeof(int64_t) * DTK_MAX_INFLIGHT,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamEndCapture(c->stream, v ? &c->graph_short : &c->graph));
    HIPCHK(c, hipGraphInstantiate(v ? &c->graph_short_exec : &c->graph_exec, v ? c->graph_short : c->graph, nullptr, nullptr, 0));
  }
  c->graph_ready = true;
  return DTK_OK;
}

void compute_rope_tables(const dtk_config& cfg, std::vector<uint16_t>& cosv, std::vector<uint16_t>& sinv) {
  // HF LlamaRotaryEmbedding: inv_freq = 1/theta^(2i/hd) (/factor for 'linear'), fp32;
  // freqs = pos * inv_freq in fp32; cos/sin cast to the activation dtype (bf16).
  const int T = cfg.max_positions;
  cosv.resize((size_t)T * 64); sinv.resize((size_t)T * 64);
  for (int i = 0; i < 64; ++i) {
    float inv = (float)(1.0 / pow((double)cfg.rope_theta, (double)(2 * i) / 128.0));
    if (cfg.rope_factor > 0.f && cfg.rope_factor != 1.f) inv = inv / cfg.rope_factor;
    for (int p = 0; p < T; ++p) {
      const float fr = inv * (float)p;
      cosv[(size_t)p * 64 + i] = host_f2bf((float)cos((double)fr));
      sinv[(size_t)p * 64 + i] = host_f2bf((float)sin((double)fr));
    }
  }
}

}  // namespace

// ============================================================================ C ABI
// the library is built with -fvisibility=hidden: the C ABI of include/dtk.h is all it exports
#pragma GCC visibility push(default)
extern "C" {

int dtk_abi_version(void) { return DTK_ABI_VERSION; }

int dtk_abi_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(dtk_config);
    case 1: return (int)sizeof(dtk_sampling);
    case 2: return (int)sizeof(dtk_stats);
    case 3: return (int)offsetof(dtk_sampling, seed);
    case 4: return (int)offsetof(dtk_config, reserved);
    case 5: return (int)offsetof(dtk_stats, probe_event_pair_ms);
    case 6: return (int)sizeof(dtk_join);
    case 7: return (int)sizeof(dtk_engine_stats);
    case 8: return (int)sizeof(dtk_engine_ops);
    case 9: return (int)offsetof(dtk_join, sampling);
    case 10: return (int)offsetof(dtk_join, error_out);
    default: return -1;
  }
}

const char* dtk_last_error(const dtk_ctx* ctx) {
  if (!ctx) return g_create_error.c_str();
  if (g_thread_error_ctx == ctx) return g_thread_error.c_str();     // this thread's own last failure on ctx
  return ctx->err.c_str();
}

int dtk_create(const dtk_config* cfg, int device, dtk_ctx** out) {
  if (!cfg || !out) return fail(nullptr, DTK_ERR_ARG, "dtk_create: null argument");
  *out = nullptr;
  if (cfg->head_dim != 128) return fail(nullptr, DTK_ERR_ARG, "head_dim must be 128 (got %d)", cfg->head_dim);
  if (cfg->hidden != cfg->heads * cfg->head_dim)
    return fail(nullptr, DTK_ERR_ARG, "hidden (%d) != heads*head_dim", cfg->hidden);
  if (cfg->reserved[2] < 0 || (cfg->reserved[2] > 0 && cfg->heads % cfg->reserved[2] != 0))
    return fail(nullptr, DTK_ERR_ARG, "kv_heads (%d) must divide heads (%d)", cfg->reserved[2], cfg->heads);
  if (cfg->hidden % 8 || cfg->ffn % 8 || cfg->vit_dim % 8 || cfg->vit_mlp % 8)
    return fail(nullptr, DTK_ERR_ARG, "dims must be multiples of 8");
  if (cfg->vit_dim % cfg->vit_heads) return fail(nullptr, DTK_ERR_ARG, "vit_dim %% vit_heads != 0");
  const int vhd = cfg->vit_dim / cfg->vit_heads;
  if (vhd != 72 && vhd != 128 && vhd != 64 && vhd != 32)
    return fail(nullptr, DTK_ERR_ARG, "unsupported ViT head dim %d", vhd);
  // timm PatchEmbed = Conv2d(stride p): 384/14 -> 27 patches per side, the 6 trailing pixels are dropped
  const int np = cfg->vit_image / cfg->vit_patch;
  if ((np * np) % cfg->concat_patches) return fail(nullptr, DTK_ERR_ARG, "patches %% concat != 0");
  if (cfg->max_positions < 8 || cfg->vocab < 2) return fail(nullptr, DTK_ERR_ARG, "bad sizes");
  if (cfg->vit_feature_layer < 0 || cfg->vit_feature_layer >= cfg->vit_depth)
    return fail(nullptr, DTK_ERR_ARG, "vit_feature_layer out of range");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= device)
    return fail(nullptr, DTK_ERR_HIP, "no HIP device %d (count %d: %s)", device, ndev, hipGetErrorString(e));
  dtk_ctx* c = new dtk_ctx();
  c->cfg = *cfg;
  c->device = device;
  c->d = cfg->hidden; c->L = cfg->layers; c->H = cfg->heads; c->ff = cfg->ffn; c->V = cfg->vocab;
  c->mb_single = c->mb_batch = sample_mb_supported(cfg->vocab);   // default sampling is greedy: the multi-block chain serves it
  c->KVH = cfg->reserved[2] > 0 ? cfg->reserved[2] : cfg->heads;      // GQA (v2: LLaMA-3.1, 32 / 8)
  c->proj_bias = (cfg->reserved[3] & DTK_ARCH_PROJ_NO_BIAS) == 0;     // v2 connector: Linear(3*D -> d, bias=False)
  c->Tmax = cfg->max_positions;
  // split-K factor of the decode attention: an explicit config value applies to both paths; auto = 16 for one sequence
  // (ds-7b 369.6 -> 373.0, ds-1.3b 1092 -> 1111, v2-8b 343.4 -> 345.1 tok/s over 8) and 8 for the batched step
  // (32 slots already give 8192 blocks: 4.53 ms/step vs 4.69 with 16)
  c->S = cfg->attn_splits > 0 ? cfg->attn_splits : 4;     // tile-interleaved splits: 4 x 128 rows cover 512 keys per memory round trip
  if (const char* es = getenv("DTK_ATTN_SPLITS")) { const int v = atoi(es); if (v >= 1 && v <= 16) c->S = v; }   // tuning aid
  c->wfmt = cfg->reserved[1] == 1 ? 1 : 0;
  // up to 64 decoding slots (one, two or four 16-column MFMA tiles) + up to 8 slots that are only ever prefilled / forked from
  // (prefix cache: one per image in flight, BASELINE config 5 = 8 images)
  c->nb = cfg->reserved[0] < 0 ? 0 : (cfg->reserved[0] > DTK_MAX_SLOTS ? DTK_MAX_SLOTS : cfg->reserved[0]);
  c->nt = c->nb > 33 ? 4 : (c->nb > 17 ? 2 : 1);   // 17 / 33 / 65 = 16 / 32 / 64 decoding slots + the prefix slot
  c->bseq.resize((size_t)c->nb);
  c->vD = cfg->vit_dim; c->vDepth = cfg->vit_depth; c->vH = cfg->vit_heads; c->vHd = vhd;
  c->vMlp = cfg->vit_mlp; c->vN = np * np;
  c->vPatchK = 3 * cfg->vit_patch * cfg->vit_patch;
  c->vPatchLd = (int)align_up((size_t)c->vPatchK, 8);
  c->nImg = c->vN / cfg->concat_patches;
  const char* gm = getenv("DTK_GEMM");
  c->gemm_naive = gm && !strcmp(gm, "naive");
  const char* ac = getenv("DTK_ATTN_COMBINE");
  // measured default (tools/tune_decode.py, profiles/r02_tune_decode_*.log): 512-thread tile kernel, 4 splits, partials reduced in
  // o_proj's prologue — one launch less per layer (ds-7b 384.2 -> 387.0 tok/s, ds-1.3b 1137 -> 1174)
  c->attn_combine = !ac ? 0 : (!strcmp(ac, "consumer") ? 0 : (!strcmp(ac, "inkernel") ? 1 : 2));
  c->attn_threads = 512;
  if (const char* at = getenv("DTK_ATTN_THREADS")) c->attn_threads = atoi(at);
  if (c->S > 16) c->S = 16;
  if (const char* fm = getenv("DTK_ATTN_FULL_MAX")) c->attn_full_max = atoi(fm);
  if (const char* gv = getenv("DTK_GEMV_VARIANTS")) {  // "epi:variant,epi:variant" (tuning aid)
    int e = 0, v = 0;
    const char* p = gv;
    while (sscanf(p, "%d:%d", &e, &v) == 2) {
      set_gemv_default_variant(e, v);
      p = strchr(p, ',');
      if (!p) break;
      ++p;
    }
  }

#define CCHK(call)                                                                         \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fail(nullptr, DTK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));           \
      dtk_destroy(c);                                                                      \
      return DTK_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)
  CCHK(hipSetDevice(device));
  CCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CCHK(hipStreamCreateWithFlags(&c->stream_vit, hipStreamNonBlocking));
  Planner sz;
  plan(c, sz, false);
  c->arena_bytes = align_up(sz.off, 256) + 256;
  CCHK(hipMalloc((void**)&c->arena, c->arena_bytes));
  Planner real;
  real.base = c->arena;
  plan(c, real, true);
  CCHK(hipMemsetAsync(c->arena, 0, c->arena_bytes, c->stream));
  CCHK(hipHostMalloc((void**)&c->tok_ring_host, sizeof(int64_t) * DTK_MAX_INFLIGHT, hipHostMallocDefault));
  for (int i = 0; i < DTK_MAX_INFLIGHT; ++i) CCHK(hipEventCreateWithFlags(&c->step_done[i], hipEventDisableTiming));
  if (c->nb > 0) {
    CCHK(hipHostMalloc((void**)&c->bs_host, sizeof(BatchState) * DTK_MAX_INFLIGHT, hipHostMallocDefault));
    CCHK(hipHostMalloc((void**)&c->tokb_host, sizeof(int64_t) * TOKB_WORDS, hipHostMallocDefault));
    CCHK(hipHostMalloc((void**)&c->st_stage, sizeof(DecState) * (DTK_MAX_SLOTS), hipHostMallocDefault));
    CCHK(hipHostMalloc((void**)&c->sp_stage, sizeof(SamplingDev) * (DTK_MAX_SLOTS), hipHostMallocDefault));
    CCHK(hipHostMalloc((void**)&c->draw_stage, sizeof(uint32_t) * (DTK_MAX_SLOTS), hipHostMallocDefault));
    for (int i = 0; i < DTK_MAX_INFLIGHT; ++i) CCHK(hipEventCreateWithFlags(&c->bstep_done[i], hipEventDisableTiming));
  }
  CCHK(hipEventCreate(&c->ev_a)); CCHK(hipEventCreate(&c->ev_b)); CCHK(hipEventCreate(&c->ev_c));
  CCHK(hipEventCreate(&c->ev_va)); CCHK(hipEventCreate(&c->ev_vb));
  CCHK(hipEventCreate(&c->probe_a)); CCHK(hipEventCreate(&c->probe_b));
  // default RoPE tables (the Python loader overrides them with torch-computed ones)
  std::vector<uint16_t> cosv, sinv;
  compute_rope_tables(*cfg, cosv, sinv);
  CCHK(hipMemcpyAsync(c->rope_cos, cosv.data(), cosv.size() * 2, hipMemcpyHostToDevice, c->stream));
  CCHK(hipMemcpyAsync(c->rope_sin, sinv.data(), sinv.size() * 2, hipMemcpyHostToDevice, c->stream));
  CCHK(hipStreamSynchronize(c->stream));
#undef CCHK
  // accounting: W = decoder layers + final norm + lm_head, K = 2*L*d*2
  const uint64_t kvd = (uint64_t)c->KVH * 128;
  const uint64_t attn_lin = (uint64_t)2 * c->d * c->d + 2 * kvd * c->d;   // q, o: d x d; k, v: kvd x d
  const uint64_t per_layer = attn_lin + (uint64_t)3 * c->d * c->ff + 2 * (uint64_t)c->d;
  c->stats.weight_bytes_per_token = 2 * (per_layer * c->L + (uint64_t)c->d + (uint64_t)c->V * c->d);
  if (c->wfmt == 1) {  // 1 byte per Linear weight + fp32 scale per row; norm vectors stay bf16
    const uint64_t lin = attn_lin + (uint64_t)3 * c->d * c->ff;
    const uint64_t rows = (uint64_t)3 * c->d + 2 * kvd + (uint64_t)2 * c->ff;
    c->stats.weight_bytes_per_token = (lin + 4 * rows + 4 * (uint64_t)c->d) * c->L + 2 * (uint64_t)c->d + (uint64_t)c->V * c->d + 4 * (uint64_t)c->V;
  }
  c->stats.kv_bytes_per_ctx_token = (uint64_t)2 * c->L * kvd * 2;
  c->stats.probe_kernel_bytes = (uint64_t)2 * c->ff * c->d * (c->wfmt == 1 ? 1 : 2);
  *out = c;
  // Batched attention shape, a property of the CONTEXT (like the multi-vector family): 64 decoding slots -> shared prefixes on the
  // matrix cores + 2-wave tail blocks; fewer -> the per-slot walk with 4-wave blocks.  Measured over a rollout's private lengths
  // (profiles/r05g_step_bench.txt, ds-7b, ms per step at 4 / 260 / 480 private keys): 64 slots 3.96 / 5.43 / 6.52 against 4.35 / 5.67 /
  // 6.57 for round 4's shape; 16 slots 3.31 / 3.99 / 4.40 against 3.45 / 3.69 / 3.94 — with 512 blocks the prefix kernel's launch is not
  // paid back and 2-wave blocks keep too few loads in flight.  A slot's arithmetic depends on its context's size, never on the active set.
  if (c->nb > 0) {
    const bool big = max_decode_slots(c) >= 64;
    c->prefix_mfma = big ? 1 : 0;
    c->tail_threads = big ? 128 : 256;
  }
  if (const char* opts = getenv("DTK_OPTIONS")) {   // "name=value,name=value": dtk_set_option at create (profiling runs of unmodified tools)
    std::string all(opts);
    size_t at = 0;
    while (at < all.size()) {
      size_t end = all.find(',', at);
      if (end == std::string::npos) end = all.size();
      const std::string item = all.substr(at, end - at);
      const size_t eq = item.find('=');
      if (eq != std::string::npos && dtk_set_option(c, item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1)) != DTK_OK)
        fprintf(stderr, "DTK_OPTIONS: %s rejected (%s)\n", item.c_str(), c->err.c_str());
      at = end + 1;
    }
  }
  return DTK_OK;
}

void dtk_destroy(dtk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  if (c->graph_short_exec) (void)hipGraphExecDestroy(c->graph_short_exec);
  if (c->graph_short) (void)hipGraphDestroy(c->graph_short);
  drop_batch_graphs(c);
  for (int i = 0; i < DTK_MAX_INFLIGHT; ++i) if (c->bstep_done[i]) (void)hipEventDestroy(c->bstep_done[i]);
  if (c->bs_host) (void)hipHostFree(c->bs_host);
  if (c->tokb_host) (void)hipHostFree(c->tokb_host);
  if (c->st_stage) (void)hipHostFree(c->st_stage);
  if (c->sp_stage) (void)hipHostFree(c->sp_stage);
  if (c->draw_stage) (void)hipHostFree(c->draw_stage);
  for (int i = 0; i < DTK_MAX_INFLIGHT; ++i) if (c->step_done[i]) (void)hipEventDestroy(c->step_done[i]);
  hipEvent_t evs[] = {c->ev_a, c->ev_b, c->ev_c, c->probe_a, c->probe_b};
  for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
  if (c->tok_ring_host) (void)hipHostFree(c->tok_ring_host);
  if (c->arena) (void)hipFree(c->arena);
  if (c->stream_vit) (void)hipStreamDestroy(c->stream_vit);
  if (c->ev_va) (void)hipEventDestroy(c->ev_va);
  if (c->ev_vb) (void)hipEventDestroy(c->ev_vb);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int dtk_num_tensors(const dtk_ctx* c) { return c ? (int)c->tensors.size() : 0; }
const char* dtk_tensor_name(const dtk_ctx* c, int i) {
  if (!c || i < 0 || i >= (int)c->tensors.size()) return nullptr;
  return c->tensors[i].name.c_str();
}
int64_t dtk_tensor_numel(const dtk_ctx* c, const char* name) {
  if (!c || !name) return -1;
  auto it = c->tindex.find(name);
  return it == c->tindex.end() ? -1 : c->tensors[it->second].numel();
}

int dtk_load_tensor(dtk_ctx* c, const char* name, const void* host, int dtype, const int64_t* shape, int ndim) {
  if (!c || !name || !host || !shape || ndim < 1) return fail(c, DTK_ERR_ARG, "dtk_load_tensor: null argument");
  auto it = c->tindex.find(name);
  if (it == c->tindex.end()) return fail(c, DTK_ERR_ARG, "unknown tensor '%s'", name);
  TensorEntry& t = c->tensors[it->second];
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  if (n != t.numel()) return fail(c, DTK_ERR_ARG, "tensor '%s': %lld elements given, %lld expected", name, (long long)n, (long long)t.numel());
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint16_t> tmp;
  const uint16_t* src16 = nullptr;
  if (dtype == DTK_BF16) {
    src16 = static_cast<const uint16_t*>(host);
  } else if (dtype == DTK_F32) {
    tmp.resize((size_t)n);
    const float* f = static_cast<const float*>(host);
    for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = host_f2bf(f[i]);
    src16 = tmp.data();
  } else if (dtype == DTK_F16) {
    tmp.resize((size_t)n);
    const uint16_t* hf = static_cast<const uint16_t*>(host);
    for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = host_f2bf(host_h2f(hf[i]));
    src16 = tmp.data();
  } else {
    return fail(c, DTK_ERR_ARG, "bad dtype %d", dtype);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy2D(t.ptr, (size_t)t.stride * 2, src16, (size_t)t.cols * 2, (size_t)t.cols * 2, (size_t)t.rows, hipMemcpyHostToDevice));
  t.loaded = true;
  c->have_image = false;
  c->seq0.cached_ids.clear();
  for (auto& b : c->bseq) b.cached_ids.clear();
  c->tiled_ready = false;
  c->ptiled_ready = false;
  c->fp8_ready = false;
  return DTK_OK;
}

int dtk_read_tensor(dtk_ctx* c, const char* name, void* host_out, int64_t n_elems) {
  if (!c || !name || !host_out) return fail(c, DTK_ERR_ARG, "dtk_read_tensor: null argument");
  auto it = c->tindex.find(name);
  if (it == c->tindex.end()) return fail(c, DTK_ERR_ARG, "unknown tensor '%s'", name);
  const TensorEntry& t = c->tensors[it->second];
  if (n_elems != t.numel()) return fail(c, DTK_ERR_ARG, "tensor '%s' has %lld elements", name, (long long)t.numel());
  HIPCHK(c, hipSetDevice(c->device));
  ensure_fp8_weights(c);   // fp8 mode: the stored tensor is the de-quantised (effective) weight
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy2D(host_out, (size_t)t.cols * 2, t.ptr, (size_t)t.stride * 2, (size_t)t.cols * 2, (size_t)t.rows, hipMemcpyDeviceToHost));
  return DTK_OK;
}

int dtk_fill_synthetic(dtk_ctx* c, uint64_t seed) {
  if (!c) return DTK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  for (size_t i = 0; i < c->tensors.size(); ++i) {
    TensorEntry& t = c->tensors[i];
    if (t.name == "rope.cos" || t.name == "rope.sin") continue;
    t.loaded = true;
    if (t.stride == t.cols) {
      launch_fill_synth(t.ptr, t.numel(), seed, (uint32_t)i, t.synth_scale, t.synth_offset, c->stream);
    } else {  // padded rows: fill contiguously in scratch, then pitch-copy
      if ((size_t)t.numel() * 2 > c->scratch_bytes) return fail(c, DTK_ERR_ARG, "scratch too small");
      bf16_t* tmp = reinterpret_cast<bf16_t*>(c->scratch);
      launch_fill_synth(tmp, t.numel(), seed, (uint32_t)i, t.synth_scale, t.synth_offset, c->stream);
      HIPCHK(c, hipMemcpy2DAsync(t.ptr, (size_t)t.stride * 2, tmp, (size_t)t.cols * 2, (size_t)t.cols * 2, (size_t)t.rows, hipMemcpyDeviceToDevice, c->stream));
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_image = false;
  c->seq0.cached_ids.clear();
  for (auto& b : c->bseq) b.cached_ids.clear();
  c->tiled_ready = false;
  c->ptiled_ready = false;
  c->fp8_ready = false;
  return DTK_OK;
}

int dtk_vit_encode(dtk_ctx* c, const float* pixels, int batch, void* feats_out, void* pooled_out) {
  if (!c || !pixels || batch < 1) return fail(c, DTK_ERR_ARG, "dtk_vit_encode: bad argument");
  if (pooled_out) {   // forward_head needs the attention-pool weights: a checkpoint without them must not pool with zeros
    for (const TensorEntry& t : c->tensors)
      if (!t.loaded && t.name.compare(0, 23, "vision_model.attn_pool.") == 0)
        return fail(c, DTK_ERR_STATE, "pooler_output requested but '%s' was never loaded (checkpoint without the pooling head)", t.name.c_str());
  }
  std::lock_guard<std::mutex> vit_guard(c->vit_mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t img = (size_t)3 * c->cfg.vit_image * c->cfg.vit_image;
  hipStream_t sv = c->stream_vit;
  // forward() semantics (pooled requested): last_hidden_state = forward_features = ALL blocks + final norm; without the
  // head: get_intermediate_layers(n=[feature_layer], norm=True).  The two differ when feature_layer != depth - 1.
  const bf16_t* hidden = (pooled_out && c->cfg.vit_feature_layer != c->vDepth - 1) ? c->last_hidden : c->feats;
  for (int b0 = 0; b0 < batch; b0 += DTK_VIT_BATCH) {
    const int B = std::min(batch - b0, (int)DTK_VIT_BATCH);
    HIPCHK(c, hipMemcpyAsync(c->pixels_dev, pixels + (size_t)b0 * img, (size_t)B * img * 4, hipMemcpyHostToDevice, sv));
    HIPCHK(c, hipEventRecord(c->ev_va, sv));
    vit_forward(c, pooled_out != nullptr, sv, B);
    HIPCHK(c, hipEventRecord(c->ev_vb, sv));
    if (feats_out)
      HIPCHK(c, hipMemcpyAsync((bf16_t*)feats_out + (size_t)b0 * c->vN * c->vD, hidden, (size_t)B * c->vN * c->vD * 2, hipMemcpyDeviceToHost, sv));
    if (pooled_out)
      HIPCHK(c, hipMemcpyAsync((bf16_t*)pooled_out + (size_t)b0 * c->vD, c->pooled, (size_t)B * c->vD * 2, hipMemcpyDeviceToHost, sv));
    HIPCHK(c, hipStreamSynchronize(sv));
    HIPCHK(c, hipGetLastError());
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_va, c->ev_vb) == hipSuccess) c->stats.last_vit_ms = ms / (float)B;
    c->stats.vit_images += (uint64_t)B;
  }
  return DTK_OK;  // IMG (the projected prefix of the cached prefill image) is left untouched
}

static int prefill_impl(dtk_ctx* c, SeqHost& sh, bf16_t* kvbase, float* logits_dst, DecState* st_dst, bool is_single,
                        const int64_t* ids, int T, const float* pixels, uint64_t image_key, int flags, float* logits_out) {
  const size_t kv_layer = (size_t)2 * c->KVH * c->Tmax * 128;
  auto kc = [&](int l) { return kvbase + (size_t)l * kv_layer; };
  auto vc = [&](int l) { return kvbase + (size_t)l * kv_layer + (size_t)c->KVH * c->Tmax * 128; };
  if (!c || !ids || T < 1) return fail(c, DTK_ERR_ARG, "dtk_prefill: bad argument");
  if (T > c->Tmax) return fail(c, DTK_ERR_RANGE, "prompt of %d tokens exceeds max_positions %d", T, c->Tmax);
  std::lock_guard<std::mutex> vit_guard(c->vit_mu);
  HIPCHK(c, hipSetDevice(c->device));
  // drain pending decode steps (their tokens are dropped)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ensure_prefill_tiles(c);
  if (is_single) c->waited = c->launched = 0;  // the device draw counter restarts with this prefill
  else c->bwaited = c->blaunched;
  // ---- locate the image placeholder run (reference v1/modeling_detikzify.py:179-184)
  int img_start = -1, img_count = 0;
  for (int t = 0; t < T; ++t) {
    if (ids[t] < 0 || ids[t] >= c->V) return fail(c, DTK_ERR_ARG, "token id %lld out of range", (long long)ids[t]);
    if (ids[t] == c->cfg.image_token_id) { if (img_start < 0) img_start = t; img_count++; }
  }
  const bool has_img = img_count > 0;
  const bool use_img = has_img && (pixels != nullptr || ((flags & DTK_PREFILL_REUSE_IMAGE) && c->have_image && c->cached_image_key == image_key));
  if (use_img) {
    if (img_count != c->nImg)
      return fail(c, DTK_ERR_ARG, "The number of image patch tokens should be the same as the number of image patches.");
    for (int t = 0; t < c->nImg; ++t)
      if (ids[img_start + t] != c->cfg.image_token_id)
        return fail(c, DTK_ERR_ARG, "The image patch tokens should be consecutive.");
  }
  HIPCHK(c, hipEventRecord(c->ev_a, c->stream));
  // ---- longest common prefix with the cached sequence (output-identical KV reuse).  The KV of the image positions
  // depends on the image, not on the (all equal) placeholder ids: the cached sequence must have been computed with the
  // same image key, and with / without spliced image features never mixes.
  int start = 0;
  const bool same_image = !has_img || (sh.cached_with_image == use_img && (!use_img || (image_key != 0 && sh.image_key == image_key)));
  if ((flags & DTK_PREFILL_REUSE_PREFIX) && same_image && !sh.cached_ids.empty()) {
    const int lim = (int)std::min<size_t>(sh.cached_ids.size(), (size_t)T - 1);
    while (start < lim && sh.cached_ids[start] == ids[start]) ++start;
  }
  // ---- image features are needed only if an image position has to be recomputed
  if (use_img && start < img_start + c->nImg) {
    const bool reuse = (flags & DTK_PREFILL_REUSE_IMAGE) && c->have_image && c->cached_image_key == image_key;
    if (!reuse) {
      if (!pixels) return fail(c, DTK_ERR_ARG, "pixels required (no cached image for this key)");
      const size_t img = (size_t)3 * c->cfg.vit_image * c->cfg.vit_image;
      HIPCHK(c, hipMemcpyAsync(c->pixels_dev, pixels, img * 4, hipMemcpyHostToDevice, c->stream));
      vit_forward(c, false, c->stream);
      project_image(c);
      c->stats.vit_images++;
      c->have_image = true;
      c->cached_image_key = image_key;
    }
  }
  HIPCHK(c, hipEventRecord(c->ev_b, c->stream));
  sh.last_reuse_start = start;
  const int n = T - start;
  std::vector<int32_t> ids32((size_t)n);
  for (int t = 0; t < n; ++t) ids32[(size_t)t] = (int32_t)ids[start + t];
  HIPCHK(c, hipMemcpyAsync(c->ids_dev, ids32.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  hipStream_t s = c->stream;
  const int d = c->d, ff = c->ff;
  launch_embed_gather(c->ids_dev, c->embed, c->X, n, d, s);
  if (use_img) {  // splice projected image features over the placeholder embeddings
    const int lo = std::max(img_start, start), hi = img_start + c->nImg;
    if (hi > lo) launch_copy_rows(c->IMG + (size_t)(lo - img_start) * d, d, c->X + (size_t)(lo - start) * d, d, hi - lo, d, s);
  }
  const float scale = 1.0f / sqrtf(128.f);
  c->launch_refused = false;      // (a batched step that was refused leaves it set: this prefill judges its own launches)
  launch_rmsnorm_rows(c->X, d, c->layers[0].ln1, c->Xn, d, n, d, c->cfg.rms_eps, s);
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    const int qkvn = d + 2 * c->KVH * 128;
    if (!qkv_rope_fused(c, w, n, start, kc(l), vc(l), s)) {
      gemm_role(c, c->Xn, d, w.wqkv, w.p_wqkv, d, nullptr, 0, c->QKV, qkvn, n, qkvn, d, 0, nullptr, nullptr, 0);
      launch_rope_scatter(c->QKV, c->Qh, kc(l), vc(l), c->rope_cos, c->rope_sin, n, start, c->H, c->KVH, c->Tmax, s);
    }
    AttnArgs a;
    a.Q = c->Qh; a.q_sh = (long)n * 128; a.q_st = 128;
    a.K = kc(l); a.k_sh = (long)c->Tmax * 128; a.k_st = 128;
    a.V = vc(l); a.v_sh = (long)c->Tmax * 128; a.v_st = 128;
    a.O = c->AO; a.o_sh = 128; a.o_st = d;
    a.H = c->H; a.Tq = n; a.Tk = T; a.hd = 128; a.causal = 1; a.q_offset = start; a.scale = scale; a.impl = c->attn_impl; a.kv_group = c->H / c->KVH;
    launch_attention(a, s);
    gemm_role(c, c->AO, d, w.wo, w.p_wo, d, c->X, d, c->X, d, n, d, d, GEMM_RESIDUAL, w.ln2, c->Xn, d);     // + post_attention_layernorm -> Xn
    bool fused = false;      // gate/up + SiLU*mul in one launch where the role is one chain and the shape takes k_gemm_g3 (bit-identical to the pair below)
    if (c->swiglu_fused && w.p_wgui && !c->gemm_naive) {
      GemmArgs g;
      g.A = c->Xn; g.lda = d; g.W = w.wgu; g.Wt = w.p_wgui; g.ldw = d; g.bias = nullptr; g.residual = nullptr; g.ldr = 0;
      g.C = c->ACT; g.ldc = ff; g.M = n; g.N = 2 * ff; g.K = d; g.flags = 0;
      fused = launch_gemm_g3_swiglu(g, s);
    }
    if (!fused && c->swiglu_fused && !c->gemm_naive && c->prefill_sk && n <= SK_CHUNK_ROWS && n <= c->sk_sl_min_rows && (ff % 4) == 0) {
      // a SLICED gate/up role (d = 2048 models): its reduction is SiLU*mul (one chunk of rows; larger prompts take the pair below)
      const int S = std::min(sk_role_slices(2 * ff, d), c->prefill_sk == 1 ? 8 : c->prefill_sk);
      GemmArgs g;
      g.A = c->Xn; g.lda = d; g.W = w.wgu; g.Wt = w.p_wgu; g.ldw = d; g.bias = nullptr; g.residual = nullptr; g.ldr = 0;
      g.C = c->GU; g.ldc = 2 * ff; g.M = n; g.N = 2 * ff; g.K = d; g.flags = 0; g.kslices = S;
      g.part = c->skpart; g.part_stride = (long)SK_CHUNK_ROWS * 2 * ff;
      if (S > 1 && (size_t)S * SK_CHUNK_ROWS * 2 * (size_t)ff <= c->skpart_floats && gemm_sk_supported(g)) {
        if (!launch_gemm_sk_partials(g, s)) c->launch_refused = true;
        else launch_sk_reduce_swiglu(g, c->ACT, ff, s);
        fused = true;
      }
    }
    if (!fused) {
      gemm_role(c, c->Xn, d, w.wgu, w.p_wgu, d, nullptr, 0, c->GU, 2 * ff, n, 2 * ff, d, 0, nullptr, nullptr, 0);
      launch_silu_mul(c->GU, ff, c->ACT, n, s);
    }
    // + the next layer's input_layernorm -> Xn (the final norm runs on the last row only, inside the lm_head GEMV below)
    gemm_role(c, c->ACT, ff, w.wdown, w.p_wdown, ff, c->X, d, c->X, d, n, d, ff, GEMM_RESIDUAL, l + 1 < c->L ? c->layers[l + 1].ln1 : nullptr, c->Xn, d);
  }
  if (c->launch_refused) { c->launch_refused = false; return fail(c, DTK_ERR_STATE, "prefill: a sliced-K projection was refused by its kernel (nothing launched for it)"); }
  // final norm + lm_head on the last position only (the sampler consumes logits[:, -1])
  GemvArgs g{};
  g.W = c->lm_head; g.W8 = c->q_lm_head; g.wscale = c->s_lm_head; g.N = c->V; g.K = d; g.x = c->X + (size_t)(n - 1) * d; g.norm_w = c->final_norm;
  g.eps = c->cfg.rms_eps; g.logits = logits_dst;
  launch_gemv(PRO_RMSNORM, EPI_LOGITS, g, s);
  DecState st0{};
  st0.pos = T - 1; st0.next_pos = T; st0.token = (int32_t)ids[T - 1]; st0.draw = 0;
  HIPCHK(c, hipMemcpyAsync(st_dst, &st0, sizeof st0, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(c->ev_c, s));
  if (logits_out) HIPCHK(c, hipMemcpyAsync(logits_out, logits_dst, (size_t)c->V * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, c->ev_a, c->ev_c) == hipSuccess) c->stats.last_prefill_ms = ms;
  if (hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) c->stats.last_vit_ms = ms;
  c->stats.prefill_tokens += (uint64_t)n;
  sh.cached_ids.assign(ids, ids + T);
  sh.cached_with_image = use_img;
  sh.image_key = use_img ? image_key : 0;
  sh.host_next_pos = T;
  sh.have_logits = true;
  // a new prefill starts a new generation: reset the draw counter of the sampler
  return DTK_OK;
}

int dtk_prefill(dtk_ctx* c, const int64_t* ids, int T, const float* pixels, uint64_t image_key, int flags, float* logits_out) {
  if (!c) return DTK_ERR_ARG;
  return prefill_impl(c, c->seq0, c->kv, c->logits, c->st, true, ids, T, pixels, image_key, flags, logits_out);
}

int dtk_prefill_slot(dtk_ctx* c, int slot, const int64_t* ids, int T, const float* pixels, uint64_t image_key, int flags, float* logits_out) {
  if (!c || slot < 0 || slot >= c->nb) return fail(c, DTK_ERR_ARG, "dtk_prefill_slot: slot %d of %d", slot, c ? c->nb : 0);
  SeqHost& sh = c->bseq[(size_t)slot];
  sh.last_reuse_start = 0;
  const int rc = prefill_impl(c, sh, c->kvb + (size_t)slot * c->kv_slot_stride, c->logits_b + (size_t)slot * c->V,
                              c->st_b + slot, false, ids, T, pixels, image_key, flags, logits_out);
  // positions >= kept of this slot were (or may have been) rewritten: shared-prefix reads stay valid only below that
  const int kept = rc == DTK_OK ? sh.last_reuse_start : 0;
  auto clip = [&](SeqHost& q) { q.share_len = std::min(q.share_len, kept); if (q.share_len <= 0) { q.share_src = -1; q.share_len = 0; } };
  clip(sh);
  for (int j = 0; j < c->nb; ++j)
    if (j != slot && c->bseq[(size_t)j].share_src == slot) clip(c->bseq[(size_t)j]);
  return rc;
}

static int set_sampling_impl(dtk_ctx* c, const dtk_sampling* sp, SamplingDev* sp_dst, DecState* st_dst, bool is_single) {
  if (!c || !sp) return fail(c, DTK_ERR_ARG, "dtk_set_sampling: null argument");
  if (sp->n_bad < 0 || sp->n_bad > 8 || sp->n_begin_suppress < 0 || sp->n_begin_suppress > 8 ||
      sp->n_always_suppress < 0 || sp->n_always_suppress > 8)
    return fail(c, DTK_ERR_ARG, "at most 8 ids per suppression list");
  if (sp->do_sample && !(sp->temperature > 0.f)) return fail(c, DTK_ERR_ARG, "temperature must be > 0");
  if (sp->do_sample && !(sp->top_p > 0.f && sp->top_p <= 1.f)) return fail(c, DTK_ERR_ARG, "top_p must be in (0, 1]");
  HIPCHK(c, hipSetDevice(c->device));
  SamplingDev dv{};
  dv.do_sample = sp->do_sample; dv.temperature = sp->temperature; dv.top_p = sp->top_p; dv.top_k = sp->top_k;
  dv.seed = sp->seed;
  dv.n_bad = sp->n_bad; dv.n_begin = sp->n_begin_suppress; dv.n_always = sp->n_always_suppress;
  for (int i = 0; i < 8; ++i) {
    dv.bad_ids[i] = sp->bad_ids[i]; dv.begin_ids[i] = sp->begin_suppress_ids[i]; dv.always_ids[i] = sp->always_suppress_ids[i];
  }
  if (!is_single && c->sp_stage) {
    // a slot's parameters change between two of ITS sequences, while a step of the OTHER slots may be in flight: the upload is
    // queued behind that step from a pinned per-slot staging record (rewritten only at the slot's next set_sampling, sequences
    // later) instead of draining the stream — a drain per join cost the batch a ~5 ms bubble (128 joins: 0.57 s of a 8 s search)
    const int slot = (int)(sp_dst - c->sp_b);
    c->sp_stage[slot] = dv;
    c->draw_stage[slot] = 0;
    HIPCHK(c, hipMemcpyAsync(sp_dst, &c->sp_stage[slot], sizeof dv, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(&st_dst->draw, &c->draw_stage[slot], sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  } else {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(sp_dst, &dv, sizeof dv, hipMemcpyHostToDevice));
    const uint32_t zero = 0;
    HIPCHK(c, hipMemcpy(&st_dst->draw, &zero, sizeof zero, hipMemcpyHostToDevice));
  }
  if (is_single) { c->sampling = *sp; c->launched = c->waited = 0; }
  // which sampler the captured graphs must contain: the multi-block chain serves large vocabularies unless a
  // configuration needs top-k; a change of kind drops the graph (re-captured by the next launch)
  const bool needs_topk = sp->do_sample && sp->top_k > 0 && sp->top_k < c->V;
  if (is_single) {
    const bool mb = sample_mb_preferred(c->V, sp->do_sample != 0) && !needs_topk;
    if (mb != c->mb_single) {
      c->mb_single = mb;
      if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
      if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
      if (c->graph_short_exec) { (void)hipGraphExecDestroy(c->graph_short_exec); c->graph_short_exec = nullptr; }
      if (c->graph_short) { (void)hipGraphDestroy(c->graph_short); c->graph_short = nullptr; }
      c->graph_ready = false;
    }
  } else {
    const int slot = (int)(sp_dst - c->sp_b);
    c->slot_topk[slot] = needs_topk;
    c->slot_samples[slot] = sp->do_sample != 0;
    bool any = false, any_sampling = false;
    for (int j = 0; j < c->nb; ++j) { any = any || c->slot_topk[j]; any_sampling = any_sampling || c->slot_samples[j]; }
    const bool mb = sample_mb_preferred(c->V, any_sampling) && !any;
    if (mb != c->mb_batch) {
      c->mb_batch = mb;
      drop_batch_graphs(c);
    }
  }
  return DTK_OK;
}

int dtk_set_sampling(dtk_ctx* c, const dtk_sampling* sp) {
  if (!c) return DTK_ERR_ARG;
  return set_sampling_impl(c, sp, c->sp, c->st, true);
}

int dtk_set_sampling_slot(dtk_ctx* c, int slot, const dtk_sampling* sp) {
  if (!c || slot < 0 || slot >= c->nb) return fail(c, DTK_ERR_ARG, "dtk_set_sampling_slot: slot %d of %d", slot, c ? c->nb : 0);
  return set_sampling_impl(c, sp, c->sp_b + slot, c->st_b + slot, false);
}

int dtk_num_slots(const dtk_ctx* c) { return c ? c->nb : 0; }
int dtk_max_positions(const dtk_ctx* c) { return c ? c->Tmax : 0; }
int dtk_max_decode_slots(const dtk_ctx* c) { return (c && c->nb > 0) ? max_decode_slots(c) : 0; }

// One batched decode step for the slots with active[slot] != 0 (every one must have been prefilled).
int dtk_decode_batch_launch(dtk_ctx* c, const int32_t* active) {
  if (!c || !active) return fail(c, DTK_ERR_ARG, "dtk_decode_batch_launch: null argument");
  if (c->nb <= 0) return fail(c, DTK_ERR_STATE, "context was created without batch slots");
  if (c->blaunched - c->bwaited >= DTK_MAX_INFLIGHT) return fail(c, DTK_ERR_STATE, "too many batch steps in flight");
  int n_active = 0;
  for (int j = 0; j < DTK_MAX_BATCH; ++j) {
    if (!active[j]) continue;
    if (j >= max_decode_slots(c)) return fail(c, DTK_ERR_ARG, "slot %d cannot decode: slots 0..%d of this %d-slot context do", j, max_decode_slots(c) - 1, c->nb);
    const SeqHost& sh = c->bseq[(size_t)j];
    if (!sh.have_logits) return fail(c, DTK_ERR_STATE, "slot %d: decode before prefill", j);
    if (sh.host_next_pos >= c->Tmax) return fail(c, DTK_ERR_RANGE, "slot %d: context length %d reached max_positions", j, sh.host_next_pos);
    ++n_active;
  }
  if (!n_active) return fail(c, DTK_ERR_ARG, "no active slot");
  HIPCHK(c, hipSetDevice(c->device));
  ensure_tiled_weights(c);
  BatchState* hb = c->bs_host + (c->blaunched % DTK_MAX_INFLIGHT);
  for (int j = 0; j < DTK_MAX_BATCH; ++j) {
    hb->active[j] = active[j] ? 1 : 0;
    const bool sh_ok = c->share_reads && j < c->nb && c->bseq[(size_t)j].share_src >= 0;
    hb->share_src[j] = sh_ok ? c->bseq[(size_t)j].share_src : -1;
    hb->share_len[j] = sh_ok ? c->bseq[(size_t)j].share_len : 0;
  }
  hb->step = (int32_t)(c->blaunched % DTK_MAX_INFLIGHT);
  // Shared prefixes for k_attn_prefix_g: the active slots grouped by (share_src, share_len) — both properties of the slot alone, so a
  // slot's arithmetic never depends on which other slots decode with it — in chunks of 16 (one MFMA column tile), sources in slot
  // order.  A source slot that decodes itself is not a member of its forks' groups (that WOULD depend on the others); more than
  // DTK_PFX_GROUPS = 64 groups cover the worst case (64 slots that share nothing), so no slot ever falls back because of its company.
  hb->n_groups = 0;
  for (int j = 0; j < DTK_MAX_BATCH; ++j) { hb->group_plus1[j] = 0; hb->pfx_len_of[j] = 0; }
  if (c->prefix_mfma && !mv_family(c)) {
    for (int j = 0; j < DTK_MAX_BATCH && hb->n_groups < DTK_PFX_GROUPS; ++j) {
      if (!active[j] || hb->group_plus1[j] || hb->share_src[j] < 0 || hb->share_len[j] < 4) continue;
      const int src = hb->share_src[j], len = hb->share_len[j];
      PfxGroup* g = nullptr;
      for (int k = j; k < DTK_MAX_BATCH; ++k) {
        if (!active[k] || hb->group_plus1[k] || hb->share_src[k] != src || hb->share_len[k] != len) continue;
        if (!g || g->n == 16) {
          if (hb->n_groups == DTK_PFX_GROUPS) break;
          g = &hb->groups[hb->n_groups++];
          g->src = src; g->len = len; g->n = 0; g->pad = 0;
        }
        g->slot[g->n++] = k;
        hb->group_plus1[k] = hb->n_groups;
        hb->pfx_len_of[k] = len;
      }
    }
    for (int gi = 0; gi < hb->n_groups; ++gi)
      for (int k = hb->groups[gi].n; k < 16; ++k) hb->groups[gi].slot[k] = hb->groups[gi].slot[0];
  }
  int hi = 0;
  for (int j = 0; j < DTK_MAX_BATCH; ++j) if (active[j]) hi = j;
  c->nt_step = hi < 16 ? 1 : (hi < 32 ? 2 : 4);      // column tiles this step needs (per-column results do not depend on it)
  c->mv_step = mv_family(c) ? (hi < 1 ? 1 : (hi < 2 ? 2 : 4)) : 0;   // vectors of a multi-vector step (per-slot results do not depend on it)
  HIPCHK(c, hipMemcpyAsync(c->bs_dev, hb, sizeof(BatchState), hipMemcpyHostToDevice, c->stream));
  if (c->use_graph) {
    int rc = ensure_batch_graph(c);
    if (rc) return rc;
    HIPCHK(c, hipGraphLaunch(c->bgraph_exec[step_graph_index(c)], c->stream));
  } else {
    c->launch_refused = false;
    if (c->mv_step) batch_step_launches_mv(c); else batch_step_launches(c);
    if (c->launch_refused || dtk_lds_attr_error(c->device))
      return fail(c, DTK_ERR_STATE, c->launch_refused ? "batched step: a projection's shape has no kernel in its family (nothing launched for it)"
                                                      : "batched step: raising a kernel's dynamic-LDS limit failed (hipFuncSetAttribute, see stderr)");
    HIPCHK(c, hipMemcpyAsync(c->tokb_host, c->tokb_dev, sizeof(int64_t) * TOKB_WORDS, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipEventRecord(c->bstep_done[c->blaunched % DTK_MAX_INFLIGHT], c->stream));
  c->stats.last_batch_step_slots = (uint32_t)(c->mv_step ? c->mv_step : 16 * c->nt_step);
  c->stats.last_batch_step_fp8_mfma = (!c->mv_step && mx_step(c)) ? 1u : 0u;
  c->blaunched++;
  c->stats.decode_steps++;
  for (int j = 0; j < DTK_MAX_BATCH; ++j)
    if (active[j]) { c->bseq[(size_t)j].host_next_pos++; c->bseq[(size_t)j].cached_ids.push_back(-1); }
  return DTK_OK;
}

// tokens_out[DTK_MAX_BATCH]: the token sampled for every slot that was active in the oldest un-read step (-1 otherwise)
int dtk_decode_batch_wait(dtk_ctx* c, int64_t* tokens_out) {
  if (!c || !tokens_out) return fail(c, DTK_ERR_ARG, "dtk_decode_batch_wait: null argument");
  if (c->bwaited >= c->blaunched) return fail(c, DTK_ERR_STATE, "no batch step in flight");
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t k = c->bwaited;
  const int ring = (int)(k % DTK_MAX_INFLIGHT);
  HIPCHK(c, hipEventSynchronize(c->bstep_done[ring]));
  const uint32_t dev_err = (uint32_t)((volatile int64_t*)c->tokb_host)[TOKB_ERR];
  if (dev_err) {      // sticky: the tokens of this and every later step cannot be trusted
    c->stats.device_errors = dev_err;
    c->bwaited++;
    return fail(c, DTK_ERR_HIP, "%u in-kernel hand-off waits expired (LDS-ring GEMVs): the step's results are invalid; re-create the context", dev_err);
  }
  const BatchState* hb = c->bs_host + ring;
  // how many steps were launched after step k for each slot (their cached ids are still -1)
  for (int j = 0; j < DTK_MAX_BATCH; ++j) {
    tokens_out[j] = -1;
    if (!hb->active[j]) continue;
    const int64_t tok = ((volatile int64_t*)c->tokb_host)[(size_t)ring * DTK_MAX_BATCH + j];
    tokens_out[j] = tok;
    SeqHost& sh = c->bseq[(size_t)j];
    size_t later = 0;
    for (uint64_t q = k + 1; q < c->blaunched; ++q) later += c->bs_host[q % DTK_MAX_INFLIGHT].active[j] ? 1 : 0;
    if (sh.cached_ids.size() > later) sh.cached_ids[sh.cached_ids.size() - 1 - later] = tok;
  }
  c->bwaited++;
  return DTK_OK;
}

// Copy the KV of the first n_tokens positions of slot src into slot dst (SURVEY §8 f1 / the proposed
// dtk_kv_fork): rollouts that share a prefix (always: the 243 image tokens) reuse its KV bit for bit instead
// of re-running ViT + prefill.  dst then needs a dtk_prefill_slot(..., DTK_PREFILL_REUSE_PREFIX) of the full
// prompt, which only processes what lies beyond the common prefix (at least the last token).
int dtk_kv_fork(dtk_ctx* c, int src, int dst, int n_tokens) {
  if (!c || src < 0 || dst < 0 || src >= c->nb || dst >= c->nb || src == dst)
    return fail(c, DTK_ERR_ARG, "dtk_kv_fork: bad slots %d -> %d of %d", src, dst, c ? c->nb : 0);
  SeqHost& a = c->bseq[(size_t)src];
  if (n_tokens < 1 || (size_t)n_tokens > a.cached_ids.size() || n_tokens > c->Tmax)
    return fail(c, DTK_ERR_ARG, "dtk_kv_fork: %d tokens but slot %d holds %zu", n_tokens, src, a.cached_ids.size());
  for (int i = 0; i < n_tokens; ++i)
    if (a.cached_ids[(size_t)i] < 0) return fail(c, DTK_ERR_STATE, "dtk_kv_fork: source has un-read tokens");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t pitch = (size_t)c->Tmax * 128 * 2;           // one (layer, k|v, head) plane
  const size_t rows = (size_t)c->L * 2 * c->KVH;
  HIPCHK(c, hipMemcpy2DAsync(c->kvb + (size_t)dst * c->kv_slot_stride, pitch, c->kvb + (size_t)src * c->kv_slot_stride, pitch,
                             (size_t)n_tokens * 128 * 2, rows, hipMemcpyDeviceToDevice, c->stream));
  SeqHost& b = c->bseq[(size_t)dst];
  for (int j = 0; j < c->nb; ++j)          // whoever read its prefix from dst must stop: dst is being overwritten
    if (c->bseq[(size_t)j].share_src == dst) { c->bseq[(size_t)j].share_src = -1; c->bseq[(size_t)j].share_len = 0; }
  // dst now holds a bit-identical copy of src[0, n): its attention may read those rows from src (or from src's own
  // source when src itself is a fork covering them), so all forks of one prefix stream the same memory
  if (a.share_src >= 0 && a.share_src != dst && n_tokens <= a.share_len) { b.share_src = a.share_src; b.share_len = n_tokens; }
  else { b.share_src = src; b.share_len = n_tokens; }
  b.cached_ids.assign(a.cached_ids.begin(), a.cached_ids.begin() + n_tokens);
  b.cached_with_image = a.cached_with_image;
  b.image_key = a.image_key;
  b.host_next_pos = n_tokens;
  b.have_logits = false;
  // a fork of the WHOLE source sequence also inherits its next-token logits: the destination can decode at
  // once (a rollout from the MCTS root needs no prefill at all).  The source must not have decoded since its
  // prefill (its logits buffer would belong to a later position).
  if ((size_t)n_tokens == a.cached_ids.size() && a.have_logits && a.host_next_pos == n_tokens) {
    HIPCHK(c, hipMemcpyAsync(c->logits_b + (size_t)dst * c->V, c->logits_b + (size_t)src * c->V, (size_t)c->V * 4,
                             hipMemcpyDeviceToDevice, c->stream));
    DecState& st0 = c->st_stage[dst];             // pinned per-slot record: queued behind the step in flight, no drain
    st0 = DecState{};
    st0.pos = n_tokens - 1; st0.next_pos = n_tokens; st0.token = (int32_t)a.cached_ids[(size_t)n_tokens - 1]; st0.draw = 0;
    HIPCHK(c, hipMemcpyAsync(c->st_b + dst, &st0, sizeof st0, hipMemcpyHostToDevice, c->stream));
    b.have_logits = true;
  }
  return DTK_OK;
}

// Longest common prefix of `ids` with what slot's KV cache holds (prefilled AND decoded tokens), under the rules of
// DTK_PREFILL_REUSE_PREFIX: an image prompt only matches a cache computed with the same image key.  The engine uses it to give
// a returning MCTS tree the slot that still holds its previous rollout.
int dtk_slot_lcp(dtk_ctx* c, int slot, const int64_t* ids, int T, uint64_t image_key, int* lcp_out) {
  if (!c || !ids || !lcp_out || T < 1 || slot < 0 || slot >= c->nb) return fail(c, DTK_ERR_ARG, "dtk_slot_lcp: bad argument");
  const SeqHost& sh = c->bseq[(size_t)slot];
  int img_count = 0;
  for (int t = 0; t < T; ++t) img_count += ids[t] == c->cfg.image_token_id;
  const bool has_img = img_count > 0;
  const bool same_image = !has_img ? !sh.cached_with_image : (sh.cached_with_image && image_key != 0 && sh.image_key == image_key);
  int n = 0;
  if (same_image) {
    const int lim = (int)std::min<size_t>(sh.cached_ids.size(), (size_t)T);
    while (n < lim && sh.cached_ids[(size_t)n] == ids[n]) ++n;
  }
  *lcp_out = n;
  return DTK_OK;
}

// Diagnostic: the token ids slot's cache is known to hold (-1 = a decoded token whose step has not been read yet); returns the
// number of ids written (at most n_max).
int dtk_slot_cached_ids(dtk_ctx* c, int slot, int64_t* out, int n_max) {
  if (!c || !out || slot < 0 || slot >= c->nb || n_max < 0) return -1;
  const SeqHost& sh = c->bseq[(size_t)slot];
  const int n = (int)std::min<size_t>(sh.cached_ids.size(), (size_t)n_max);
  for (int i = 0; i < n; ++i) out[i] = sh.cached_ids[(size_t)i];
  return n;
}

// Resume decoding in place: the slot's cache already holds ids[0, T-1) (dtk_slot_lcp >= T - 1) — typically the path to an MCTS
// node inside the slot's own previous rollout.  No prefill: the context is cut back to T - 1 tokens and the NEXT batched step
// forwards ids[T-1] for this slot instead of sampling (DecState.force_plus1), returning ids[T-1] as that step's token; the step
// after it samples the first new token (draw 0: begin-suppress applies there).  The rows [0, T-1) stay as they were written
// (by the prefill GEMMs and / or the decode kernels of the earlier sequence), like any DTK_PREFILL_REUSE_PREFIX hit.
int dtk_resume_slot(dtk_ctx* c, int slot, const int64_t* ids, int T, uint64_t image_key) {
  if (!c || !ids || T < 2 || slot < 0 || slot >= c->nb) return fail(c, DTK_ERR_ARG, "dtk_resume_slot: bad argument");
  if (T > c->Tmax) return fail(c, DTK_ERR_RANGE, "prompt of %d tokens exceeds max_positions %d", T, c->Tmax);
  if (ids[T - 1] == c->cfg.image_token_id) return fail(c, DTK_ERR_ARG, "dtk_resume_slot: the last prompt token is an image position (its input is a patch feature, not an embedding)");
  int lcp = 0;
  const int rc = dtk_slot_lcp(c, slot, ids, T, image_key, &lcp);
  if (rc) return rc;
  if (lcp < T - 1) return fail(c, DTK_ERR_STATE, "dtk_resume_slot: slot %d holds %d of the %d prompt tokens (needs %d)", slot, lcp, T, T - 1);
  for (uint64_t q = c->bwaited; q < c->blaunched; ++q)
    if (c->bs_host[q % DTK_MAX_INFLIGHT].active[slot]) return fail(c, DTK_ERR_STATE, "dtk_resume_slot: slot %d is part of a step in flight", slot);
  HIPCHK(c, hipSetDevice(c->device));
  SeqHost& sh = c->bseq[(size_t)slot];
  for (int j = 0; j < c->nb; ++j) {        // forks that read rows >= T - 1 from this slot would see them overwritten
    SeqHost& o = c->bseq[(size_t)j];
    if (o.share_src == slot && o.share_len > T - 1) { o.share_src = -1; o.share_len = 0; }
  }
  if (sh.share_len > T - 1) sh.share_len = T - 1;
  if (sh.share_len <= 0) { sh.share_src = -1; sh.share_len = 0; }
  sh.cached_ids.resize((size_t)T - 1);
  sh.host_next_pos = T - 1;
  sh.have_logits = true;                   // "may decode": the first step does not read the logits buffer
  sh.last_reuse_start = T - 1;
  DecState& st0 = c->st_stage[slot];
  st0 = DecState{};
  st0.pos = T - 2; st0.next_pos = T - 1; st0.token = (int32_t)ids[T - 2]; st0.draw = 0; st0.force_plus1 = (int32_t)ids[T - 1] + 1;
  HIPCHK(c, hipMemcpyAsync(c->st_b + slot, &st0, sizeof st0, hipMemcpyHostToDevice, c->stream));   // ordered behind the step in flight
  return DTK_OK;
}

int dtk_get_logits_slot(dtk_ctx* c, int slot, float* out) {
  if (!c || !out || slot < 0 || slot >= c->nb) return fail(c, DTK_ERR_ARG, "dtk_get_logits_slot: bad argument");
  if (!c->bseq[(size_t)slot].have_logits) return fail(c, DTK_ERR_STATE, "no logits yet");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->logits_b + (size_t)slot * c->V, (size_t)c->V * 4, hipMemcpyDeviceToHost));
  return DTK_OK;
}

int dtk_context_len_slot(const dtk_ctx* c, int slot) {
  return (c && slot >= 0 && slot < c->nb) ? c->bseq[(size_t)slot].host_next_pos : -1;
}

int dtk_set_graph_mode(dtk_ctx* c, int enabled) {
  if (!c) return DTK_ERR_ARG;
  c->use_graph = enabled == 1;
  c->probe = enabled == 2;   // 2: plain launches with HIP-event probe around the gate/up GEMV
  if (c->probe && c->stats.probe_event_pair_ms == 0.0) {
    // what an event pair costs with NOTHING between the two records (timestamp packets): part of every probe interval
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sum = 0.0; int n = 0;
    for (int i = 0; i < 40; ++i) {
      HIPCHK(c, hipEventRecord(c->probe_a, c->stream));
      HIPCHK(c, hipEventRecord(c->probe_b, c->stream));
      HIPCHK(c, hipEventSynchronize(c->probe_b));
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->probe_a, c->probe_b) == hipSuccess && i >= 8) { sum += ms; ++n; }
    }
    if (n) c->stats.probe_event_pair_ms = sum / n;
  }
  return DTK_OK;
}

int dtk_decode_launch(dtk_ctx* c) {
  if (!c) return DTK_ERR_ARG;
  if (!c->seq0.have_logits) return fail(c, DTK_ERR_STATE, "dtk_decode before dtk_prefill");
  if (c->launched - c->waited >= DTK_MAX_INFLIGHT) return fail(c, DTK_ERR_STATE, "too many decode steps in flight");
  if (c->seq0.host_next_pos >= c->Tmax) return fail(c, DTK_ERR_RANGE, "context length %d reached max_positions", c->seq0.host_next_pos);
  HIPCHK(c, hipSetDevice(c->device));
  ensure_fp8_weights(c);
  if (c->use_graph) {
    int rc = ensure_graph(c);
    if (rc) return rc;
    const bool short_ctx = c->seq0.host_next_pos < c->attn_full_max;
    HIPCHK(c, hipGraphLaunch(short_ctx ? c->graph_short_exec : c->graph_exec, c->stream));
  } else {
    if (c->probe && c->launched > c->waited) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->probe && c->probe_pending) {  // read the previous step's pair before re-recording it
      float ms = 0.f;
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (hipEventElapsedTime(&ms, c->probe_a, c->probe_b) == hipSuccess) {
        c->stats.probe_kernel_ms_sum += ms;
        c->stats.probe_kernel_launches++;
      }
      c->probe_pending = false;
    }
    decode_step_launches(c, c->probe != 0, c->seq0.host_next_pos < c->attn_full_max);
    if (c->probe) c->probe_pending = true;
    HIPCHK(c, hipMemcpyAsync(c->tok_ring_host, c->tok_ring_dev, sizeof(int64_t) * DTK_MAX_INFLIGHT, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipEventRecord(c->step_done[c->launched % DTK_MAX_INFLIGHT], c->stream));
  c->launched++;
  c->seq0.host_next_pos++;
  c->stats.decode_steps++;
  c->seq0.cached_ids.push_back(-1);  // filled in by dtk_decode_wait
  return DTK_OK;
}

int dtk_decode_wait(dtk_ctx* c, int64_t* token_out) {
  if (!c || !token_out) return fail(c, DTK_ERR_ARG, "dtk_decode_wait: null argument");
  if (c->waited >= c->launched) return fail(c, DTK_ERR_STATE, "no decode step in flight");
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t k = c->waited;
  HIPCHK(c, hipEventSynchronize(c->step_done[k % DTK_MAX_INFLIGHT]));
  // the ring slot of draw k is rewritten only by draw k + DTK_MAX_INFLIGHT, which cannot have
  // been launched yet; later copies of the whole ring rewrite it with the same value
  const int64_t tok = ((volatile int64_t*)c->tok_ring_host)[k % DTK_MAX_INFLIGHT];
  *token_out = tok;
  const size_t idx = c->seq0.cached_ids.size() - (size_t)(c->launched - k);
  c->seq0.cached_ids[idx] = tok;
  c->waited++;
  return DTK_OK;
}

int dtk_decode(dtk_ctx* c, int64_t* token_out) {
  int rc = dtk_decode_launch(c);
  if (rc) return rc;
  return dtk_decode_wait(c, token_out);
}

int dtk_get_logits(dtk_ctx* c, float* out) {
  if (!c || !out) return fail(c, DTK_ERR_ARG, "dtk_get_logits: null argument");
  if (!c->seq0.have_logits) return fail(c, DTK_ERR_STATE, "no logits yet");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->logits, (size_t)c->V * 4, hipMemcpyDeviceToHost));
  return DTK_OK;
}

int dtk_context_len(const dtk_ctx* c) { return c ? c->seq0.host_next_pos : -1; }

int dtk_synchronize(dtk_ctx* c) {
  if (!c) return DTK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return DTK_OK;
}

int dtk_get_stats(dtk_ctx* c, dtk_stats* out) {
  if (!c || !out) return DTK_ERR_ARG;
  *out = c->stats;
  return DTK_OK;
}

// In-situ microbenchmark of one decode GEMV role over all layers (distinct weights per launch,
// so nothing is served from the 256 MB Infinity Cache): avg microseconds per launch via HIP events.
// role: 0 qkv, 1 o_proj, 2 gate/up, 3 down, 4 lm_head.  Clobbers the decode state.
int dtk_bench_gemv(dtk_ctx* c, int role, int variant, int reps, float* avg_us) {
  if (!c || !avg_us || reps < 1) return fail(c, DTK_ERR_ARG, "dtk_bench_gemv: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (role < 0 || role > 4) return fail(c, DTK_ERR_ARG, "dtk_bench_gemv: role %d (0 qkv, 1 o_proj, 2 gate/up, 3 down, 4 lm_head)", role);
  const bool same_layer = (variant & 0x100) != 0;  // every launch re-reads layer 0 (Infinity Cache probe)
  const int plain = (variant >> 9) & 3;            // 0x200: same weights through PRO_COPY + EPI_STORE; 0x400: PRO_RMSNORM + EPI_STORE
  variant &= 0xff;
  const bool shipped = variant == 0xff;            // 0xff: whatever launch_gemv picks for this role and model (the kernel a decode step runs)
  auto launch = [&](int pro, int epi, const GemvArgs& g) { if (shipped) launch_gemv(pro, epi, g, s); else launch_gemv_variant(pro, epi, variant, g, s); };
  auto one_pass = [&]() {
    for (int l = 0; l < c->L; ++l) {
      const LayerW& w = c->layers[same_layer ? 0 : l];
      GemvArgs g{};
      if (plain) {   // cost of the fused prologue / epilogue = full role - this
        g.eps = c->cfg.rms_eps; g.y = c->GU;
        if (role == 0) { g.W = w.wqkv; g.N = c->d + 2 * c->KVH * 128; g.K = c->d; g.x = c->x; g.norm_w = w.ln1; }
        else if (role == 1) { g.W = w.wo; g.N = c->d; g.K = c->d; g.x = c->attn_out; g.norm_w = w.ln1; }
        else if (role == 2) { g.W = w.wgu; g.N = 2 * c->ff; g.K = c->d; g.x = c->x; g.norm_w = w.ln2; }
        else if (role == 3) { g.W = w.wdown; g.N = c->d; g.K = c->ff; g.x = c->act; g.norm_w = w.ln2; }
        else { g.W = c->lm_head; g.N = c->V; g.K = c->d; g.x = c->x; g.norm_w = c->final_norm; }
        launch_gemv_variant(plain == 2 ? PRO_RMSNORM : PRO_COPY, EPI_STORE, variant, g, s);
        continue;
      }
      g.eps = c->cfg.rms_eps; g.st = c->st; g.T_max = c->Tmax; g.d = c->d; g.ff = c->ff; g.H = c->H; g.KVH = c->KVH;
      g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.pm = c->pm; g.pl = c->pl; g.po = c->po; g.S = c->S;
      if (role == 0) { g.W = w.wqkv; g.N = c->d + 2 * c->KVH * 128; g.K = c->d; g.x = c->x; g.norm_w = w.ln1; g.q_out = c->q; g.kcache = kcache(c, l); g.vcache = vcache(c, l); launch(PRO_RMSNORM, EPI_QKV, g); }
      else if (role == 1) { g.W = w.wo; g.N = c->d; g.K = c->d; g.x = c->attn_out; g.y = c->q; launch(PRO_COPY, EPI_RESID, g); }
      else if (role == 2) { g.W = w.wgu; g.N = 2 * c->ff; g.K = c->d; g.x = c->x; g.norm_w = w.ln2; g.y = c->act; launch(PRO_RMSNORM, EPI_SWIGLU, g); }
      else if (role == 3) { g.W = w.wdown; g.N = c->d; g.K = c->ff; g.x = c->act; g.y = c->q; launch(PRO_COPY, EPI_RESID, g); }
      else { g.W = c->lm_head; g.N = c->V; g.K = c->d; g.x = c->x; g.norm_w = c->final_norm; g.logits = c->logits; launch(PRO_RMSNORM, EPI_LOGITS, g); }
    }
  };
  one_pass();  // warm-up (code objects, clocks)
  HIPCHK(c, hipEventRecord(c->ev_a, s));
  for (int r = 0; r < reps; ++r) one_pass();
  HIPCHK(c, hipEventRecord(c->ev_b, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
  *avg_us = ms * 1e3f / (float)(reps * c->L);
  c->seq0.have_logits = false;
  return DTK_OK;
}

// Runtime options (tests / tuning): "attn_full_max" = contexts below this use the one-block-per-head
// decode attention (0 = always split-K); "attn_combine" = 0 consumer | 1 in-kernel | 2 own kernel.
int dtk_set_option(dtk_ctx* c, const char* name, int value) {
  if (!c || !name) return DTK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!strcmp(name, "attn_full_max")) c->attn_full_max = value;
  else if (!strcmp(name, "gemm_tile")) {   // MFMA GEMM block tile: 0 auto, 1 = 64x64, 2 = 128x64, 3 = 128x128 (process-wide)
    if (value < 0 || value > 5) return fail(c, DTK_ERR_ARG, "gemm_tile must be 0..5");
    set_gemm_tile(value);
  }
  else if (!strcmp(name, "share_prefix_reads")) c->share_reads = value != 0;
  else if (!strcmp(name, "prefix_mfma") || !strcmp(name, "pfx_splits") || !strcmp(name, "gemv_b_wide") ||
           !strcmp(name, "tail_threads")) {
    if (!strcmp(name, "prefix_mfma")) c->prefix_mfma = value != 0;
    else if (!strcmp(name, "tail_threads")) { if (value != 64 && value != 128 && value != 256 && value != 512) return fail(c, DTK_ERR_ARG, "tail_threads must be 64, 128, 256 or 512"); c->tail_threads = value; }
    else if (!strcmp(name, "pfx_splits")) { if (value < 1 || value > 4) return fail(c, DTK_ERR_ARG, "pfx_splits must be 1..4"); c->pfx_splits = value; }
    else { if (value < 0 || value > 6) return fail(c, DTK_ERR_ARG, "gemv_b_wide must be 0..6"); set_gemv_b_wide(value); }
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "gemm_impl")) {
    if (value < 0 || value > 4 || value == 1) return fail(c, DTK_ERR_ARG, "gemm_impl must be 0 (registers), 2 (k_gemm_glds where the shape has the tiles), 3 (auto) or 4 (k_gemm_g3: 8 waves, 3 LDS stages)");
    set_gemm_impl(value);
  }
  else if (!strcmp(name, "gemm_g3_min_blocks")) {
    if (value < 1) return fail(c, DTK_ERR_ARG, "gemm_g3_min_blocks must be >= 1");
    set_gemm_g3_min_blocks(value);
  }
  else if (!strcmp(name, "gemm_glds_min_tiles")) {
    if (value < 1) return fail(c, DTK_ERR_ARG, "gemm_glds_min_tiles must be >= 1");
    set_gemm_glds_min_tiles(value);
  }
  else if (!strcmp(name, "gqa_fused")) {
    if (value < 0 || value > 2) return fail(c, DTK_ERR_ARG, "gqa_fused must be 0 (a block per query head), 1 (per K/V head) or 2 (per pair of query heads)");
    c->gqa_fused = value;
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "resid_split")) {
    set_resid_split(value != 0);
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "gemv_bkl")) { set_gemv_bkl(value != 0); drop_batch_graphs(c); }
  else if (!strcmp(name, "gemv_bus")) { set_gemv_bus(value); drop_batch_graphs(c); }          // 64-slot qkv / gate-up by k_gemv_bus (kernels_batch_ks.hip): 0 off, 128 default per role, bit 0 qkv, bit 1 gate/up
  else if (!strcmp(name, "act_fp8")) {          // fp8 models: the MFMA-family step on the fp8 matrix cores with MXFP8 activations (kernels_batch_mx.hip)
    if (value && c->wfmt == 1 && c->nb > 0 && !c->mx_ok) return fail(c, DTK_ERR_ARG, "act_fp8: the model's shapes are not covered by the fp8 matrix-core kernels");
    c->act_fp8 = value != 0;
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "mx_nc_qkv") || !strcmp(name, "mx_nc_gu") || !strcmp(name, "mx_nc_lm_head")) {   // compute waves per block of k_gemv_mxu (0 = from the CU count)
    if (value < 0 || value > 4) return fail(c, DTK_ERR_ARG, "%s must be 0..4", name);
    set_mx_nc(!strcmp(name, "mx_nc_qkv") ? 0 : (!strcmp(name, "mx_nc_gu") ? 1 : 2), value);
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "gemv_br_wd")) { if (value != 4 && value != 8) return fail(c, DTK_ERR_ARG, "gemv_br_wd must be 4 or 8"); set_gemv_br_wd(value); drop_batch_graphs(c); }
  else if (!strcmp(name, "gemv_loaders")) { set_gemv_loaders(value >= 2 ? 2 : 1); drop_batch_graphs(c); }
  else if (!strcmp(name, "gemv_xw")) { set_gemv_xw(value < 0 ? 0 : (value > 2 ? 2 : value)); drop_batch_graphs(c); }   // x waves of k_gemv_bl / k_gemv_bkl
  else if (!strcmp(name, "gemv_bc")) {
    if (value < 0 || value > 255) return fail(c, DTK_ERR_ARG, "gemv_bc must be 0..255 (0 off; 128 = the measured default per role and weight format; else bit 0 qkv, bit 1 gate/up, bit 2 lm_head through k_gemv_bc; bits 4..6 = units per block, 0 = one CU's share)");
    set_gemv_bc(value);
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "gemv_bl")) {
    if (value < 0 || value > 127) return fail(c, DTK_ERR_ARG, "gemv_bl must be 0..127 (bit 6: bf16 qkv through k_gemv_br; bit 0: gate/up + lm_head, bit 1: qkv by pair units, bit 2: fp8 weights too, bit 3: qkv as pair + V tile per block where that fills the chip, bit 4: for any MHA model, bit 5: fp8 weights through registers (k_gemv_br, K = 4096))");
    set_gemv_bl(value);
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "attn_nt")) { c->attn_nt = value != 0; drop_batch_graphs(c); }
  else if (!strcmp(name, "mv_slots")) {     // contexts with at most value + 1 slots decode with the multi-vector kernels (0 = never)
    if (value < 0 || value > 4) return fail(c, DTK_ERR_ARG, "mv_slots must be 0..4");
    if (c->blaunched != c->bwaited) return fail(c, DTK_ERR_STATE, "mv_slots: a batch step is in flight");
    c->mv_slots = value;
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "mv_tail_threads")) {
    if (value != 256 && value != 512 && value != 1024) return fail(c, DTK_ERR_ARG, "mv_tail_threads must be 256, 512 or 1024");
    // GQA models run the group-fused blocks of k_attn_tail_b, which exist for fewer block sizes (launch_attn_decode_b): a value without
    // an instantiation used to be accepted and silently ignored
    const int grp = c->KVH > 0 ? c->H / c->KVH : 1;
    if (grp == 4 && value == 1024) return fail(c, DTK_ERR_ARG, "mv_tail_threads: GQA groups of 4 have blocks of 256 or 512 threads");
    if (grp == 2 && value != 256) return fail(c, DTK_ERR_ARG, "mv_tail_threads: GQA groups of 2 have blocks of 256 threads only");
    c->mv_tail_threads = value;
    drop_batch_graphs(c);
  }
  else if (!strncmp(name, "mv_shape_", 9)) {   // block shape of one multi-vector role: qkv | o | gu | down | lm_head; 0..3, -1 = measured default
    const char* r = name + 9;
    const int role = !strcmp(r, "qkv") ? EPI_QKV : !strcmp(r, "o") ? 5 : !strcmp(r, "gu") ? EPI_SWIGLU : !strcmp(r, "down") ? EPI_RESID : !strcmp(r, "lm_head") ? EPI_LOGITS : -1;
    if (role < 0 || value < -1 || value > 3) return fail(c, DTK_ERR_ARG, "mv_shape_{qkv,o,gu,down,lm_head} must be -1..3");
    set_gemv_mv_shape(role, value);
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "resid_kparts")) {     // batched N = d roles as k_gemv_bkp + k_resid_norm_b (64 slots, bf16 weights)
    c->resid_kparts = value != 0;
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "gemv_bx")) {
    if (value < 0 || value > 4) return fail(c, DTK_ERR_ARG, "gemv_bx must be 0..4");
    set_gemv_bx(value);
    drop_batch_graphs(c);
  }
  else if (!strcmp(name, "gemm_ring")) {
    if (value < 2 || value > 4) return fail(c, DTK_ERR_ARG, "gemm_ring must be 2..4");
    set_gemm_ring(value);
  }
  else if (!strcmp(name, "vit_feature_layer")) {   // diagnostic (per-block parity tests): which block's normed output dtk_vit_encode(feats) returns
    if (value < 0 || value >= c->vDepth) return fail(c, DTK_ERR_ARG, "vit_feature_layer must be 0..%d", c->vDepth - 1);
    HIPCHK(c, hipStreamSynchronize(c->stream_vit));
    c->cfg.vit_feature_layer = value;
    c->have_image = false; c->cached_image_key = 0;   // a cached image prefix was projected from the old layer's features
    c->seq0.image_key = 0; c->seq0.cached_ids.clear();
    for (SeqHost& sh : c->bseq) { sh.image_key = 0; sh.cached_ids.clear(); sh.share_src = -1; sh.share_len = 0; }
  }
  else if (!strcmp(name, "qkv_rope_fused")) c->qkv_rope_fused = value != 0;
  else if (!strcmp(name, "swiglu_fused")) c->swiglu_fused = value != 0;
  else if (!strcmp(name, "sk_sl_min_rows")) { if (value < 1) return fail(c, DTK_ERR_ARG, "sk_sl_min_rows must be >= 1"); c->sk_sl_min_rows = value; }
  else if (!strcmp(name, "gemm_epi_direct")) set_gemm_epi_direct(value != 0);   // k_gemm_g3 without the LDS-transposed epilogue (default 0; process-wide; bit-identical)
  else if (!strcmp(name, "gemm_wt")) set_gemm_wt(value != 0);     // k_gemm_g3's W stage from the fragment-major copy (default 1; process-wide; bit-identical)
  else if (!strcmp(name, "gemm_sk_tile")) {   // block tile of the sliced-K GEMM: 0 = 256 x 128, 1 = 128 x 256, 2 = by M (process-wide; bit-identical)
    if (value < 0 || value > 2) return fail(c, DTK_ERR_ARG, "gemm_sk_tile must be 0, 1 or 2");
    set_gemm_sk_tile(value);
  }
  else if (!strcmp(name, "prefill_sk")) {   // sliced-K prefill GEMMs (default 1).  The two settings round differently: cached prefixes are dropped
    if (value < 0 || value > 8 || (value & (value - 1))) return fail(c, DTK_ERR_ARG, "prefill_sk must be 0 (one-chain GEMMs), 1 (sliced by the weight shape: the default) or a cap of 2 / 4 / 8 slices");
    c->prefill_sk = value;
    c->seq0.cached_ids.clear();
    for (SeqHost& sh : c->bseq) { sh.cached_ids.clear(); sh.share_src = -1; sh.share_len = 0; }
  }
  else if (!strcmp(name, "gemm_bk")) {
    if (value != 64 && value != 128) return fail(c, DTK_ERR_ARG, "gemm_bk must be 64 or 128");
    set_gemm_bk(value);
  }
  else if (!strcmp(name, "gemm_stages")) {
    if (value < 1 || value > 4) return fail(c, DTK_ERR_ARG, "gemm_stages must be 1..4");
    set_gemm_stages(value);
  }
  else if (!strcmp(name, "attn_impl")) {   // prefill / ViT attention: 0 auto, 1 VALU kernel, 2 MFMA flash kernel
    if (value < 0 || value > 2) return fail(c, DTK_ERR_ARG, "attn_impl must be 0..2");
    c->attn_impl = value;
  }
  else if (!strcmp(name, "attn_threads") || !strcmp(name, "attn_splits") || !strcmp(name, "attn_combine")) {
    if (!strcmp(name, "attn_threads")) {
      if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(c, DTK_ERR_ARG, "attn_threads must be 0, 256, 512 or 1024");
      c->attn_threads = value;
    } else if (!strcmp(name, "attn_splits")) {
      if (value < 1 || value > 16) return fail(c, DTK_ERR_ARG, "attn_splits must be 1..16");
      c->S = value;
    } else {
      if (value < 0 || value > 2) return fail(c, DTK_ERR_ARG, "attn_combine must be 0..2");
      c->attn_combine = value;
    }
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->graph_short_exec) { (void)hipGraphExecDestroy(c->graph_short_exec); c->graph_short_exec = nullptr; }
    if (c->graph_short) { (void)hipGraphDestroy(c->graph_short); c->graph_short = nullptr; }
    c->graph_ready = false;
  } else return fail(c, DTK_ERR_ARG, "unknown option '%s'", name);
  return DTK_OK;
}

int dtk_set_gemv_variant(dtk_ctx* c, int epi, int variant) {
  if (!c) return DTK_ERR_ARG;
  if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
  if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
  if (c->graph_short_exec) { (void)hipGraphExecDestroy(c->graph_short_exec); c->graph_short_exec = nullptr; }
  if (c->graph_short) { (void)hipGraphDestroy(c->graph_short); c->graph_short = nullptr; }
  c->graph_ready = false;
  set_gemv_default_variant(epi, variant);
  return DTK_OK;
}

// ------------------------------------------------------------------ op-level test entry points
static int op_scratch(dtk_ctx* c, size_t bytes, size_t& off, void** p) {
  off = align_up(off, 256);
  if (off + bytes > c->scratch_bytes) return fail(c, DTK_ERR_ARG, "op scratch exhausted (%zu bytes)", off + bytes);
  *p = c->scratch + off;
  off += bytes;
  return DTK_OK;
}
#define OPBUF(T, var, n)                                                    \
  T* var = nullptr;                                                         \
  { void* p_; int rc_ = op_scratch(c, (size_t)(n) * sizeof(T), off, &p_); if (rc_) return rc_; var = (T*)p_; }

int dtk_op_gemm(dtk_ctx* c, const uint16_t* A, const uint16_t* W, const uint16_t* bias, const uint16_t* residual, int M, int N, int K, int flags, uint16_t* C) {
  if (!c || !A || !W || !C || K % 8) return fail(c, DTK_ERR_ARG, "dtk_op_gemm: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  OPBUF(bf16_t, dA, (size_t)M * K); OPBUF(bf16_t, dW, (size_t)N * K); OPBUF(bf16_t, dB, N);
  OPBUF(bf16_t, dR, (size_t)M * N); OPBUF(bf16_t, dC, (size_t)M * N);
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(dA, A, (size_t)M * K * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dW, W, (size_t)N * K * 2, hipMemcpyHostToDevice, s));
  int gf = 0;
  if ((flags & (DTK_EPI_BIAS | DTK_EPI_GELU)) && bias) { HIPCHK(c, hipMemcpyAsync(dB, bias, (size_t)N * 2, hipMemcpyHostToDevice, s)); gf |= GEMM_BIAS; }
  if (flags & DTK_EPI_GELU) gf |= gelu_flag(c);
  if ((flags & DTK_EPI_RESIDUAL) && residual) { HIPCHK(c, hipMemcpyAsync(dR, residual, (size_t)M * N * 2, hipMemcpyHostToDevice, s)); gf |= GEMM_RESIDUAL; }
  GemmArgs g;
  g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.bias = dB; g.residual = dR; g.ldr = N; g.C = dC; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.flags = gf;
  if (flags & DTK_GEMM_WT) {          // + the fragment-major copy of W (k_gemm_g3's W stage is filled from it)
    OPBUF(bf16_t, dWt, tiled_elems(N, K));
    launch_retile(dW, dWt, N, K, s);
    g.Wt = dWt;
  }
  const int S = (flags >> DTK_GEMM_KSLICES_SHIFT) & 15;       // 0 = a one-chain GEMM; 1..8 = the sliced-K family (1: one slice through the partial + reduce path)
  if (S) {
    if (S > 8) return fail(c, DTK_ERR_ARG, "dtk_op_gemm: at most 8 K slices");
    g.kslices = S;
    if (flags & DTK_GEMM_NAIVE) launch_gemm_naive(g, s);
    else {
      OPBUF(float, dP, (size_t)S * M * N);
      g.part = dP; g.part_stride = (long)M * N;
      if (flags & DTK_GEMM_SL) { if (!launch_gemm_g3_sliced(g, s)) return fail(c, DTK_ERR_ARG, "dtk_op_gemm: the in-register sliced kernel does not take this shape"); }
      else if (!launch_gemm_sk(g, nullptr, nullptr, 0, 0.f, s)) return fail(c, DTK_ERR_ARG, "dtk_op_gemm: the sliced-K kernel does not take this shape");
    }
  }
  else if (flags & DTK_GEMM_NAIVE) launch_gemm_naive(g, s); else launch_gemm_mfma(g, s);
  HIPCHK(c, hipMemcpyAsync(C, dC, (size_t)M * N * 2, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return DTK_OK;
}

int dtk_op_gemv(dtk_ctx* c, const uint16_t* W, const uint16_t* x, const uint16_t* norm_w, int N, int K, int mode, float eps, uint16_t* y) {
  if (!c || !W || !x || !y || K % 8) return fail(c, DTK_ERR_ARG, "dtk_op_gemv: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  OPBUF(bf16_t, dW, (size_t)N * K); OPBUF(bf16_t, dx, K); OPBUF(bf16_t, dn, K); OPBUF(bf16_t, dy, N);
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(dW, W, (size_t)N * K * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dx, x, (size_t)K * 2, hipMemcpyHostToDevice, s));
  if (mode == 1) { if (!norm_w) return fail(c, DTK_ERR_ARG, "norm_w required"); HIPCHK(c, hipMemcpyAsync(dn, norm_w, (size_t)K * 2, hipMemcpyHostToDevice, s)); }
  GemvArgs g{};
  g.W = dW; g.N = N; g.K = K; g.x = dx; g.norm_w = dn; g.eps = eps; g.y = dy;
  launch_gemv(mode == 1 ? PRO_RMSNORM : PRO_COPY, EPI_STORE, g, s);
  HIPCHK(c, hipMemcpyAsync(y, dy, (size_t)N * 2, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return DTK_OK;
}

// nb (1, 2 or 4) input vectors through k_gemv_mv: y[b] = W . x[b] (mode 0) or W . rmsnorm(x[b], norm_w) (mode 1); per vector
// the result must equal dtk_op_gemv's bit for bit (same lane / chunk order, same wave reduction, same block size)
int dtk_op_gemv_mv(dtk_ctx* c, const uint16_t* W, const uint16_t* X, const uint16_t* norm_w, int N, int K, int mode, float eps, int nb, uint16_t* Y) {
  if (!c || !W || !X || !Y || N < 1 || K < 8 || (K & 7) || (nb != 1 && nb != 2 && nb != 4) || (mode == 1 && !norm_w))
    return fail(c, DTK_ERR_ARG, "dtk_op_gemv_mv: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  OPBUF(bf16_t, dW, (size_t)N * K);
  OPBUF(bf16_t, dX, (size_t)nb * K);
  OPBUF(bf16_t, dG, (size_t)K);
  OPBUF(bf16_t, dY, (size_t)nb * N);
  HIPCHK(c, hipMemcpyAsync(dW, W, (size_t)N * K * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dX, X, (size_t)nb * K * 2, hipMemcpyHostToDevice, c->stream));
  if (norm_w) HIPCHK(c, hipMemcpyAsync(dG, norm_w, (size_t)K * 2, hipMemcpyHostToDevice, c->stream));
  GemvMvArgs g{};
  g.W = dW; g.N = N; g.K = K; g.X = dX; g.ldx = K; g.x_rowmajor = 1; g.norm_w = dG; g.eps = eps; g.Y = dY; g.ldy = N; g.d = K; g.ff = N;
  launch_gemv_mv(mode == 1 ? PRO_RMSNORM : PRO_COPY, EPI_STORE, nb, g, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(Y, dY, (size_t)nb * N * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return DTK_OK;
}

// Where value k of slot `slot` and the E8M0 scale of its group live in the MXFP8 buffers (csrc/mx_quant.h): pure host arithmetic (no
// GPU, no context) — tests/test_mx_layout.py checks on the CPU that this fragment order is the inverse of the operand map the
// instruction was measured to have.
int dtk_mx_layout(int G, int slot, int k, int64_t* data_off, int64_t* scale_off) {
  if ((G != 32 && G != 16) || slot < 0 || slot >= DTK_MAX_BATCH || k < 0 || !data_off || !scale_off) return DTK_ERR_ARG;
  *data_off = (int64_t)(G == 32 ? mx32_off(slot, k) : mx16_off(slot, k));
  *scale_off = (int64_t)(G == 32 ? mx32_soff(slot, k) : mx16_soff(slot, k));
  return DTK_OK;
}

// The fp8 matrix-core GEMVs of kernels_batch_mx.hip on host buffers.  W8 [N][K] e4m3 bytes + wscale [N]; X [nslots][K] bf16 rows,
// quantised to MXFP8 (groups of G = 32 | 16) by the step's own quantiser; nslots = 16 | 32 | 64 (1 / 2 / 4 slot tiles).
//   mode 0: unit kernel, logits epilogue (G = 32): Y [nslots][N] = bf16-rounded acc * wscale
//   mode 1: K-slice kernel of the N = d roles: Y [nslots][N] = the 8 slice partials added in order (fp32)
//   mode 2: unit kernel, SwiGLU epilogue (G = 32 in, N = 2 ff): y8_out / ys_out = the activation as MXFP8 groups of 16 (mx_x_bytes(ff, 16) / mx_s_bytes)
// x8_out / xs_out (optional): the quantised input as the kernels read it.
int dtk_op_gemv_mx(dtk_ctx* c, const uint8_t* W8, const float* wscale, const uint16_t* X, int N, int K, int G, int nslots, int mode,
                   float* Y, uint8_t* x8_out, uint8_t* xs_out, uint8_t* y8_out, uint8_t* ys_out) {
  if (!c || !W8 || !wscale || !X || (G != 32 && G != 16) || (nslots != 16 && nslots != 32 && nslots != 64) || mode < 0 || mode > 2)
    return fail(c, DTK_ERR_ARG, "dtk_op_gemv_mx: bad argument");
  if (mode != 1 && (G != 32 || !mx_unit_covers(K) || (N & 31))) return fail(c, DTK_ERR_ARG, "dtk_op_gemv_mx: unit kernel needs G = 32, K %% 512 == 0, N %% 32 == 0");
  if (mode == 1 && !mx_kparts_covers(N, K, G)) return fail(c, DTK_ERR_ARG, "dtk_op_gemv_mx: N = %d, K = %d, G = %d is not covered by the K-slice kernel", N, K, G);
  if ((mode != 2 && !Y) || (mode == 2 && (!y8_out || !ys_out || (N & 127)))) return fail(c, DTK_ERR_ARG, "dtk_op_gemv_mx: missing output");
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  const int ff = N / 2;
  OPBUF(uint8_t, dW, (size_t)N * K); OPBUF(uint8_t, dWm, mx_w_bytes(N, K, G)); OPBUF(float, dS, N);
  OPBUF(bf16_t, dX, (size_t)64 * K); OPBUF(uint8_t, dX8, mx_x_bytes(K, G)); OPBUF(uint8_t, dXS, mx_s_bytes(K, G));
  OPBUF(float, dY, (size_t)(mode == 1 ? 8 : 1) * 64 * N); OPBUF(BatchState, dBS, 1);
  OPBUF(uint8_t, dY8, mx_x_bytes(ff, 16)); OPBUF(uint8_t, dYS, mx_s_bytes(ff, 16));
  hipStream_t s = c->stream;
  BatchState hbs; memset(&hbs, 0, sizeof hbs);
  for (int i = 0; i < nslots; ++i) hbs.active[i] = 1;
  for (int i = 0; i < DTK_MAX_BATCH; ++i) hbs.share_src[i] = -1;
  HIPCHK(c, hipMemcpyAsync(dBS, &hbs, sizeof hbs, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dW, W8, (size_t)N * K, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dS, wscale, (size_t)N * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(dX, 0, (size_t)64 * K * 2, s));
  HIPCHK(c, hipMemcpyAsync(dX, X, (size_t)nslots * K * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(dY, 0, (size_t)(mode == 1 ? 8 : 1) * 64 * N * 4, s));
  HIPCHK(c, hipMemsetAsync(dY8, 0, mx_x_bytes(ff, 16), s));
  HIPCHK(c, hipMemsetAsync(dYS, 0, mx_s_bytes(ff, 16), s));
  launch_retile_mx(dW, dWm, N, K, G, s);
  launch_quant_mx_rows(dX, K, dX8, dXS, G, 64, s);
  GemvBArgs g{};
  g.Wm = dWm; g.wscale = dS; g.N = N; g.K = K; g.X8 = dX8; g.XS = dXS; g.bs = dBS; g.nt = nslots / 16; g.logits = dY; g.kpart = dY;
  g.d = K; g.ff = ff; g.H = 1; g.KVH = 1; g.Y8 = dY8; g.YS = dYS;
  if (mode == 0) launch_gemv_mxu(EPI_LOGITS, g, s);
  else if (mode == 2) launch_gemv_mxu(EPI_SWIGLU, g, s);
  else if (!launch_gemv_mxk(g, G, s)) return fail(c, DTK_ERR_ARG, "dtk_op_gemv_mx: no K-slice kernel for N %d K %d G %d", N, K, G);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(s));
  if (mode == 0) HIPCHK(c, hipMemcpy(Y, dY, (size_t)nslots * N * 4, hipMemcpyDeviceToHost));
  else if (mode == 1) {
    std::vector<float> part((size_t)8 * 64 * N);
    HIPCHK(c, hipMemcpy(part.data(), dY, part.size() * 4, hipMemcpyDeviceToHost));
    for (int sl = 0; sl < nslots; ++sl)
      for (int n = 0; n < N; ++n) {
        float sum = 0.f;
        for (int ks = 0; ks < 8; ++ks) sum += part[((size_t)ks * 64 + sl) * N + n];
        Y[(size_t)sl * N + n] = sum;
      }
  } else {
    HIPCHK(c, hipMemcpy(y8_out, dY8, mx_x_bytes(ff, 16), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ys_out, dYS, mx_s_bytes(ff, 16), hipMemcpyDeviceToHost));
  }
  if (x8_out) HIPCHK(c, hipMemcpy(x8_out, dX8, mx_x_bytes(K, G), hipMemcpyDeviceToHost));
  if (xs_out) HIPCHK(c, hipMemcpy(xs_out, dXS, mx_s_bytes(K, G), hipMemcpyDeviceToHost));
  return DTK_OK;
}

int dtk_op_attention(dtk_ctx* c, const uint16_t* Q, const uint16_t* K, const uint16_t* V, int H, int Tq, int Tk, int hd, int causal, int q_offset, uint16_t* O) {
  if (!c || !Q || !K || !V || !O) return fail(c, DTK_ERR_ARG, "dtk_op_attention: null argument");
  if (hd != 72 && hd != 128 && hd != 64 && hd != 32) return fail(c, DTK_ERR_ARG, "unsupported head dim %d", hd);
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  OPBUF(bf16_t, dQ, (size_t)H * Tq * hd); OPBUF(bf16_t, dK, (size_t)H * Tk * hd);
  OPBUF(bf16_t, dV, (size_t)H * Tk * hd); OPBUF(bf16_t, dO, (size_t)H * Tq * hd);
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(dQ, Q, (size_t)H * Tq * hd * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dK, K, (size_t)H * Tk * hd * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dV, V, (size_t)H * Tk * hd * 2, hipMemcpyHostToDevice, s));
  AttnArgs a;
  a.Q = dQ; a.q_sh = (long)Tq * hd; a.q_st = hd;
  a.K = dK; a.k_sh = (long)Tk * hd; a.k_st = hd;
  a.V = dV; a.v_sh = (long)Tk * hd; a.v_st = hd;
  a.O = dO; a.o_sh = (long)Tq * hd; a.o_st = hd;
  a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.causal = causal; a.q_offset = q_offset;
  a.scale = 1.0f / sqrtf((float)hd); a.impl = c->attn_impl; a.kv_group = 1;
  launch_attention(a, s);
  HIPCHK(c, hipMemcpyAsync(O, dO, (size_t)H * Tq * hd * 2, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return DTK_OK;
}

int dtk_op_layernorm(dtk_ctx* c, const uint16_t* X, const uint16_t* w, const uint16_t* b, int M, int D, float eps, uint16_t* Y) {
  if (!c || !X || !w || !b || !Y || D % 8) return fail(c, DTK_ERR_ARG, "dtk_op_layernorm: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  OPBUF(bf16_t, dX, (size_t)M * D); OPBUF(bf16_t, dw, D); OPBUF(bf16_t, db, D); OPBUF(bf16_t, dY, (size_t)M * D);
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(dX, X, (size_t)M * D * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(dw, w, (size_t)D * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(db, b, (size_t)D * 2, hipMemcpyHostToDevice, s));
  launch_layernorm_rows(dX, D, dw, db, dY, D, M, D, eps, s);
  HIPCHK(c, hipMemcpyAsync(Y, dY, (size_t)M * D * 2, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return DTK_OK;
}

int dtk_op_sample(dtk_ctx* c, const float* logits, int V, int step, int64_t* token_out, float* probs_out) {
  if (!c || !logits || !token_out || V < 1 || V > c->V) return fail(c, DTK_ERR_ARG, "dtk_op_sample: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  size_t off = 0;
  OPBUF(float, dl, V); OPBUF(int64_t, dtok, 1);
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(dl, logits, (size_t)V * 4, hipMemcpyHostToDevice, s));
  SampleArgs sa;
  sa.logits = dl; sa.V = V; sa.sp = c->sp; sa.st = c->st; sa.embed = c->embed; sa.x = c->x; sa.d = c->d;
  sa.tok_ring = dtok; sa.ring = 1; sa.probs_out = probs_out ? c->probs_dev : nullptr; sa.advance = 0;
  sa.step_override = step; sa.bs = nullptr; sa.logits_stride = 0; sa.nslots = 1; sa.mb = c->smb;
  const bool needs_topk = c->sampling.do_sample && c->sampling.top_k > 0 && c->sampling.top_k < V;
  if (sample_mb_preferred(V, c->sampling.do_sample != 0) && !needs_topk && !getenv("DTK_SAMPLER")) launch_sample_mb(sa, s); else launch_sample(sa, s);
  HIPCHK(c, hipMemcpyAsync(token_out, dtok, 8, hipMemcpyDeviceToHost, s));
  if (probs_out) HIPCHK(c, hipMemcpyAsync(probs_out, c->probs_dev, (size_t)V * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return DTK_OK;
}

}  // extern "C"
#pragma GCC visibility pop
