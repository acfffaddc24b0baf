/*
 * dtk.h — C ABI of libdtk_hip.so, the MI355X (gfx950) image->TikZ decoder behind
 * DeTikZify's Python inference API.
 *
 * The reference (potamides/DeTikZify) has NO native/FFI seam: its boundary is the
 * duck-typed (model, processor) pair returned by detikzify.model.load
 * (reference detikzify/model/__init__.py:28-61) and consumed by
 * DetikzifyGenerator.generate (detikzify/infer/generate.py:209-227) and
 * ImageSim.from_detikzify (detikzify/evaluate/imagesim.py:61-89).  This header is
 * the C ABI that sits *underneath* that Python surface: each entry point names the
 * reference call it replaces.  Plain C, opaque handle, int status (0 = ok,
 * negative = error; text via dtk_last_error), caller-owned host buffers,
 * library-owned device buffers, one HIP stream per context.  A context is not
 * thread-safe, but may be driven from any one thread at a time (the reference
 * calls model.generate from a ThreadPool(1) worker, generate.py:248-258);
 * contexts are independent -> one context per GPU / rank.
 */
#ifndef DTK_H
#define DTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTK_ABI_VERSION 6   /* 2: batch arrays of 32 entries (were 16); 3: DTK_MAX_BATCH = 64; 4: dtk_max_decode_slots,
                             * contexts with <= 5 slots decode in slots 0..3 (multi-vector kernels);
                             * 5: dtk_op_gemv_mx, dtk_mx_layout, dtk_stats.last_batch_step_fp8_mfma;
                             * 6: dtk_engine_* (the native run loop of a batch; replaces ABI 4's dtk_decode_batch_run), dtk_max_positions; dtk_last_error is
                             * per calling thread */

typedef struct dtk_ctx dtk_ctx;

enum { DTK_F32 = 0, DTK_BF16 = 1, DTK_F16 = 2 };

enum {
  DTK_OK = 0,
  DTK_ERR_ARG = -1,     /* bad argument / shape / unknown tensor name          */
  DTK_ERR_HIP = -2,     /* a HIP runtime call failed                            */
  DTK_ERR_STATE = -3,   /* call sequence violated (e.g. decode before prefill)  */
  DTK_ERR_RANGE = -4    /* context length would exceed max_positions            */
};

/* Architecture of one checkpoint.  Decoder fields are HF LlamaConfig fields
 * (read from the checkpoint's config.json, never hard-coded; reference
 * detikzify/model/v1/configuration_detikzify.py:3-13).  Vision fields describe the
 * timm ViT created at detikzify/model/v1/modeling_detikzify.py:94. */
typedef struct dtk_config {
  /* LLaMA decoder */
  int32_t hidden;          /* d                                   */
  int32_t layers;          /* L                                   */
  int32_t heads;           /* H query heads (kv heads: reserved[2]) */
  int32_t head_dim;        /* hd (must be 128)                    */
  int32_t ffn;             /* intermediate_size                   */
  int32_t vocab;           /* V                                   */
  int32_t max_positions;   /* KV capacity in tokens (<= 2048 for v1, generate.py:383) */
  float   rms_eps;
  float   rope_theta;
  float   rope_factor;     /* linear RoPE scaling factor (1 = none) */
  /* vision tower (timm vit_so400m_patch14_siglip_384 family) */
  int32_t vit_dim;         /* D (1152)                            */
  int32_t vit_depth;       /* 27                                  */
  int32_t vit_heads;       /* 16 -> head dim 72                   */
  int32_t vit_mlp;         /* 4304                                */
  int32_t vit_patch;       /* 14                                  */
  int32_t vit_image;       /* 384                                 */
  int32_t vit_feature_layer; /* block index whose output feeds the LM (modeling_detikzify.py:104) */
  float   vit_ln_eps;      /* 1e-6                                */
  int32_t vit_gelu_tanh;   /* 0 = erf GELU (timm default), 1 = tanh approximation */
  /* glue */
  int32_t concat_patches;  /* 3  (modeling_detikzify.py:101)       */
  int32_t image_token_id;  /* == BOS for v1 (v1/__init__.py:49)    */
  int32_t attn_splits;     /* split-K factor of decode attention; 0 = default */
  int32_t reserved[7];     /* [0] = batch slots for dtk_decode_batch_* (0 = none)
                            * [1] = 1: fp8 e4m3 decoder weights
                            * [2] = key/value heads (GQA, num_key_value_heads); 0 = heads (MHA, all v1 models)
                            * [3] = architecture flags, DTK_ARCH_*                        */
} dtk_config;
/* v2 connector (reference detikzify/model/modeling_detikzify.py:62-70): Linear(concat*D -> d, bias=False) */
#define DTK_ARCH_PROJ_NO_BIAS 1

/* Per-generation sampling state: the HF logits processors + sampler that
 * DetikzifyGenerator.generate configures (generate.py:218-227; HF
 * generation/utils.py _sample, logits_process.py:300-303,528-540,1395,1860-1866). */
typedef struct dtk_sampling {
  int32_t do_sample;       /* 0 = greedy argmax, 1 = multinomial                     */
  float   temperature;     /* used when do_sample                                    */
  float   top_p;           /* nucleus; 1.0 disables                                  */
  int32_t top_k;           /* 0 disables                                             */
  uint64_t seed;           /* counter-based RNG key; draw n uses counter n           */
  int32_t n_bad;           /* banned single-token ids (bad_words_ids=[[id]])         */
  int32_t bad_ids[8];
  int32_t n_begin_suppress;/* ids suppressed only for the first generated token      */
  int32_t begin_suppress_ids[8];
  int32_t n_always_suppress; /* extra ids banned at every step (fixed-work benches)  */
  int32_t always_suppress_ids[8];
} dtk_sampling;

typedef struct dtk_stats {
  uint64_t weight_bytes_per_token;  /* W: decoder layers + final norm + lm_head       */
  uint64_t kv_bytes_per_ctx_token;  /* K: 2*L*d*sizeof(bf16)                          */
  uint64_t decode_steps;            /* decode-step launches since create              */
  uint64_t prefill_tokens;          /* tokens pushed through the batched prefill      */
  uint64_t vit_images;              /* images encoded                                 */
  double   last_prefill_ms;         /* device time of last dtk_prefill (HIP events)   */
  double   last_vit_ms;             /* ViT part of it                                 */
  double   probe_kernel_ms_sum;     /* in-graph event probe around the gate/up GEMV of the middle layer */
  uint64_t probe_kernel_launches;
  uint64_t probe_kernel_bytes;      /* algorithmic bytes of one such launch           */
  double   probe_event_pair_ms;     /* elapsed time of an EMPTY hipEventRecord pair on the stream (the fixed cost inside every
                                     * probe interval; calibrated when probe mode is switched on)            */
  uint32_t last_batch_step_slots;   /* slots the kernels of the last dtk_decode_batch_launch computed: 1 | 2 | 4 (multi-vector
                                     * kernels) or 16 | 32 | 64 (one, two, four MFMA column tiles)            */
  uint32_t device_errors;           /* sticky: in-kernel protocol timeouts seen so far (a ring hand-off that expired); any
                                     * non-zero value makes dtk_decode_batch_wait fail                          */
  uint32_t last_batch_step_fp8_mfma; /* 1: the last dtk_decode_batch_launch ran the fp8 matrix-core kernels (MXFP8 activations,
                                     * option "act_fp8"), 0: the bf16-activation kernels                        */
  uint32_t reserved0;
} dtk_stats;

int  dtk_abi_version(void);
/* layout check for bindings: sizeof of 0 dtk_config, 1 dtk_sampling, 2 dtk_stats; offsetof of 3 dtk_sampling.seed,
 * 4 dtk_config.reserved, 5 dtk_stats.probe_event_pair_ms; sizeof of 6 dtk_join, 7 dtk_engine_stats, 8 dtk_engine_ops; offsetof of
 * 9 dtk_join.sampling, 10 dtk_join.error_out; -1 for anything else */
int  dtk_abi_struct_size(int which);
/* last error of a context; ctx may be NULL for the error of a failed dtk_create */
const char* dtk_last_error(const dtk_ctx* ctx);

/* Lifetime.  Allocates every weight, activation and KV buffer up front
 * (replaces DetikzifyForCausalLM.from_pretrained + initialize_vision_modules,
 * reference v1/__init__.py:35-54). */
int  dtk_create(const dtk_config* cfg, int device, dtk_ctx** out);
void dtk_destroy(dtk_ctx* ctx);

/* Weights.  Names are the checkpoint's state-dict keys: HF Llama keys
 * ("model.layers.3.self_attn.q_proj.weight", "model.embed_tokens.weight",
 * "model.norm.weight", "lm_head.weight", "model.mm_projector.weight|bias") and
 * timm ViT keys prefixed "vision_model." ("vision_model.blocks.0.attn.qkv.weight",
 * "vision_model.attn_pool.latent", ...).  Host data may be f32/bf16/f16; storage is bf16. */
int  dtk_load_tensor(dtk_ctx* ctx, const char* name, const void* host, int dtype,
                     const int64_t* shape, int ndim);
/* copy a stored tensor (bf16, same logical shape as the checkpoint tensor) back to the host */
int  dtk_read_tensor(dtk_ctx* ctx, const char* name, void* host_out_bf16, int64_t n_elems);
/* deterministic synthetic weights generated on the device (no real checkpoints
 * exist offline); bit-identical to oracle/synth.py for the same seed. */
int  dtk_fill_synthetic(dtk_ctx* ctx, uint64_t seed);
/* number of tensor names / i-th name / element count (for loaders and tests) */
int  dtk_num_tensors(const dtk_ctx* ctx);
const char* dtk_tensor_name(const dtk_ctx* ctx, int i);
int64_t dtk_tensor_numel(const dtk_ctx* ctx, const char* name);

/* Vision tower: replaces DetikzifyVisionModel.forward / get_intermediate_layers
 * (reference v1/modeling_detikzify.py:63-72).  pixels: B x 3 x S x S fp32 NCHW
 * (host).  feats_out: B x N x D bf16 (post final-LayerNorm features of the
 * configured feature layer = get_intermediate_layers(n=[layer], norm=True)), pooled_out: B x D bf16 (MAP head), either
 * may be NULL.  With pooled_out the call is DetikzifyVisionModel.forward and feats_out receives its last_hidden_state
 * (forward_features: all blocks + final norm; the same tensor unless feature_layer != depth - 1).  Requesting
 * pooled_out from a context whose attn_pool tensors were never loaded is DTK_ERR_STATE. */
int  dtk_vit_encode(dtk_ctx* ctx, const float* pixels, int batch,
                    void* feats_out_bf16, void* pooled_out_bf16);

/* Prefill: replaces DetikzifyForCausalLM.forward with input_ids.shape[1] != 1
 * (reference v1/modeling_detikzify.py:144-200,218-257).  ids: T int64 (host);
 * pixels (1x3xSxS fp32 host) may be NULL when the ids contain no image tokens or
 * when image_key matches the previously encoded image.  flags: DTK_PREFILL_*.
 * logits_last_out (V fp32, host) may be NULL.  Afterwards the context holds KV
 * for T positions and the logits of position T-1. */
#define DTK_PREFILL_REUSE_PREFIX 1   /* keep KV of the longest common prefix with the cached ids */
#define DTK_PREFILL_REUSE_IMAGE  2   /* skip the ViT when image_key equals the cached key        */
int  dtk_prefill(dtk_ctx* ctx, const int64_t* ids, int T, const float* pixels,
                 uint64_t image_key, int flags, float* logits_last_out);

/* Sampling configuration for the following decode calls (resets the draw counter). */
int  dtk_set_sampling(dtk_ctx* ctx, const dtk_sampling* s);

/* Decode: one call = one iteration of HF GenerationMixin._sample (logits
 * processors -> argmax|multinomial on the pending logits -> append -> forward of
 * the new token), hipGraph-replayed.  dtk_decode_launch enqueues a step without
 * waiting; dtk_decode_wait returns the oldest un-read token (at most
 * DTK_MAX_INFLIGHT steps may be pending).  dtk_decode = launch + wait. */
#define DTK_MAX_INFLIGHT 4
#define DTK_VIT_BATCH 8      /* images dtk_vit_encode runs as one pass over the tower (larger batches: several passes) */
int  dtk_decode_launch(dtk_ctx* ctx);
int  dtk_decode_wait(dtk_ctx* ctx, int64_t* token_out);
int  dtk_decode(dtk_ctx* ctx, int64_t* token_out);
/* logits (V fp32) the next sampling step will consume, copied to the host */
int  dtk_get_logits(dtk_ctx* ctx, float* logits_out);
/* current context length (tokens with KV) */
int  dtk_context_len(const dtk_ctx* ctx);
/* 1 = replay the decode step as a hipGraph (default), 0 = plain launches */
int  dtk_set_graph_mode(dtk_ctx* ctx, int enabled);
int  dtk_synchronize(dtk_ctx* ctx);
int  dtk_get_stats(dtk_ctx* ctx, dtk_stats* out);

/* Batched decode for independent rollouts of one GPU: dtk_config.reserved[0] = number
 * of slots (<= DTK_MAX_SLOTS), each with its own KV cache, sampling state and logits.  One
 * dtk_decode_batch_* step = one _sample iteration for every active slot with ONE pass over the weights
 * (bytes/step = W + sum_b K*t_b).  Up to 5 slots: slots 0..3 decode with the multi-vector kernels (the single-sequence GEMVs
 * carrying 1, 2 or 4 input vectors: BASELINE config 4 leaves 2 / 4 trees per rank at N = 8 / 4, reference examples/eval.py:80-83);
 * 6..17 slots: slots 0..15 decode (one 16-column MFMA tile); 18..33 slots:
 * slots 0..31 decode (two tiles); 34..72 slots: slots 0..63 decode (four tiles); slots beyond the decoding ones can only be
 * prefilled / forked from (prefix cache: one per image in flight — BASELINE config 5 runs 8 images on one GPU).
 * The `active` / `tokens_out` arrays always have DTK_MAX_BATCH entries.
 * The image embeddings cache is shared (DTK_PREFILL_REUSE_IMAGE). */
#define DTK_MAX_BATCH 64
#define DTK_MAX_SLOTS (DTK_MAX_BATCH + 8)
int  dtk_num_slots(const dtk_ctx* ctx);
/* how many slots (0 .. n-1) may take part in a decode step: 4, 16, 32 or 64, at most dtk_num_slots */
int  dtk_max_decode_slots(const dtk_ctx* ctx);
int  dtk_prefill_slot(dtk_ctx* ctx, int slot, const int64_t* ids, int T, const float* pixels,
                      uint64_t image_key, int flags, float* logits_last_out);
int  dtk_set_sampling_slot(dtk_ctx* ctx, int slot, const dtk_sampling* s);
int  dtk_decode_batch_launch(dtk_ctx* ctx, const int32_t* active /* [DTK_MAX_BATCH] */);
int  dtk_decode_batch_wait(dtk_ctx* ctx, int64_t* tokens_out /* [DTK_MAX_BATCH] */);
int  dtk_kv_fork(dtk_ctx* ctx, int src_slot, int dst_slot, int n_tokens);   /* share a prefix's KV (f1) */
/* f1, same-slot reuse for returning MCTS trees (reference infer/generate.py:305-353 re-prefills the path to the selected node
 * on every rollout): dtk_slot_lcp = how many leading tokens of `ids` the slot's KV cache still holds (prefilled or decoded, same
 * image key); dtk_resume_slot = continue from there WITHOUT a prefill when all but the last prompt token are cached — the next
 * batched step forwards ids[T-1] for this slot instead of sampling and returns it as that step's token (the caller drops it),
 * the step after samples the first new token. */
int  dtk_slot_lcp(dtk_ctx* ctx, int slot, const int64_t* ids, int n_tokens, uint64_t image_key, int* lcp_out);
int  dtk_resume_slot(dtk_ctx* ctx, int slot, const int64_t* ids, int n_tokens, uint64_t image_key);
int  dtk_slot_cached_ids(dtk_ctx* ctx, int slot, int64_t* ids_out, int n_max);   /* diagnostic: what the slot's cache holds; returns the count */
int  dtk_get_logits_slot(dtk_ctx* ctx, int slot, float* logits_out);
int  dtk_context_len_slot(const dtk_ctx* ctx, int slot);

int  dtk_max_positions(const dtk_ctx* ctx);               /* KV capacity of a sequence in tokens (dtk_config.max_positions) */

/* ---- The run loop of a batch of rollouts, native (ABI 6) ------------------------------------------------------------------------
 * Replaces, for every sequence decoded in a slot, the per-token host iteration of HF GenerationMixin._sample
 * (generation/utils.py:2875-2936) that DetikzifyGenerator.generate runs in a worker thread and consumes line by line
 * (reference detikzify/infer/generate.py:246-282): ONE native thread per engine launches and collects the batched steps — two in
 * flight, so the device never waits for the host — appends every slot's token to that slot's ring, applies the sequence's own stop
 * rules (stop ids = EOS, token budget = max_length) and wakes the slot's reader only when a FLUSH token (the caller's newline table),
 * `flush_max` tokens or the end of the sequence has arrived.  Callers block in dtk_engine_read (no interpreter lock held) and are
 * woken once per source line instead of once per token; joins and leaves are queued and executed by the same thread between steps,
 * so nothing but that thread ever touches the context's main stream.  A slot's tokens do not depend on which other slots decode
 * next to it, so a sequence is the same ids as through dtk_decode_batch_launch / _wait driven from the host.
 * dtk_engine_create drives `ctx` (which must outlive the engine and must not be used for prefill / decode calls by anybody else
 * meanwhile; dtk_vit_encode from other threads stays legal: own stream, serialised with image prefills inside the library).
 * dtk_engine_create_ops drives a caller-supplied device (the CPU tests' scripted device: tests/test_native_engine.py). */
typedef struct dtk_engine dtk_engine;
typedef struct dtk_engine_ops {      /* the device under the loop: same contracts as the dtk_* entry points of the same names */
  void* dev;
  int (*launch)(void* dev, const int32_t* active /* [DTK_MAX_BATCH] */);
  int (*wait)(void* dev, int64_t* tokens_out /* [DTK_MAX_BATCH] */);
  int (*prefill_slot)(void* dev, int slot, const int64_t* ids, int T, const float* pixels, uint64_t image_key, int flags);
  int (*set_sampling_slot)(void* dev, int slot, const dtk_sampling* s);
  int (*kv_fork)(void* dev, int src_slot, int dst_slot, int n_tokens);
  int (*slot_lcp)(void* dev, int slot, const int64_t* ids, int n_tokens, uint64_t image_key, int* lcp_out);
  int (*resume_slot)(void* dev, int slot, const int64_t* ids, int n_tokens, uint64_t image_key);
  int (*context_len_slot)(void* dev, int slot);
  const char* (*last_error)(void* dev);
  int32_t max_positions;
  int32_t decode_slots;
} dtk_engine_ops;

/* how a join got its prompt into the slot (dtk_join.how_out) */
enum { DTK_JOIN_FULL = 0,        /* whole prompt prefilled (ViT included unless the image is cached)                  */
       DTK_JOIN_FORK_TAIL = 1,   /* image prefix forked from prefix_src, what follows prefilled                       */
       DTK_JOIN_FORK_WHOLE = 2,  /* prompt == prefix: KV and next-token logits forked, nothing computed                */
       DTK_JOIN_RESUMED = 3,     /* the slot still held all but the last prompt token: continued in place, no prefill */
       DTK_JOIN_IN_PLACE = 4 };  /* the slot still held the image prefix: only what follows prefilled                 */
/* sequence states (dtk_engine_read state_out) */
enum { DTK_SEQ_RUNNING = 1, DTK_SEQ_FINISHED = 2 /* stop id or budget */, DTK_SEQ_LEFT = 3 /* dtk_engine_leave / engine destroyed */ };

typedef struct dtk_join {
  int32_t slot;                  /* decoding slot of the sequence; with n_candidates > 0: the slot taken when no candidate can resume */
  int32_t n_ids;
  const int64_t* ids;            /* the whole prompt (host)                                                                        */
  const float* pixels;           /* 1 x 3 x S x S fp32 (host) or NULL, as dtk_prefill_slot                                         */
  uint64_t image_key;
  int32_t try_resume;            /* 1: dtk_resume_slot when the slot's cache holds ids[0, n_ids - 1) (an MCTS tree coming back)     */
  int32_t n_candidates;          /* > 0 (with try_resume): resume in the candidate with the longest such match (lowest index on ties) */
  int32_t candidates[DTK_MAX_BATCH];
  int32_t prefix_len;            /* leading ids that are the shareable image prefix; 0 = no sharing: a full prefill with full_flags */
  int32_t prefix_src;            /* slot to fork the prefix from (-1: none)                                                         */
  int32_t prefix_src_whole;      /* 1: prefix_src is a prefix-cache slot (holds exactly the prefix + its next-token logits)         */
  int32_t prefix_encode;         /* 1: first prefill ids[0, prefix_len) + pixels into prefix_src (greedy), then fork                */
  int32_t prefix_in_place;       /* 1: `slot` itself still holds the prefix: prefill with DTK_PREFILL_REUSE_*                       */
  int32_t full_flags;            /* DTK_PREFILL_* of the full prefill                                                               */
  dtk_sampling sampling;
  int32_t max_new_tokens;        /* token budget of the sequence (max_length - n_ids)                                               */
  int32_t n_stop;
  int64_t stop_ids[8];           /* the sequence ends WITH the first of these it emits (EOS)                                        */
  int32_t flush_mode;            /* 0: every token wakes the reader; 1: flush tokens (dtk_engine_set_flush_tokens) / flush_max      */
  int32_t flush_max;             /* mode 1: wake the reader after at most this many undelivered tokens (0 = 64)                     */
  int32_t slot_out;              /* out: the slot the sequence decodes in                                                           */
  int32_t how_out;               /* out: DTK_JOIN_*                                                                                 */
  char    error_out[240];        /* out: text of a failed join (the engine itself stays usable unless the device failed)            */
} dtk_join;

typedef struct dtk_engine_stats {
  uint64_t steps, tokens_out, joins, resumed, steps_below_half_occupancy, host_bound_steps, reader_wakeups, wasted_slot_steps;
  double   wait_s;               /* inside the device's wait: the step time the host sees                                           */
  double   launch_s, join_s;     /* inside launch / inside join execution (prefills, forks)                                         */
  double   idle_s;               /* sequences running but no step in flight (joins being executed, start-up)                        */
  double   drain_s;              /* waiting for steps in flight because a join / leave was queued                                    */
  double   first_launch_t, last_collect_t;   /* CLOCK_MONOTONIC seconds (0 = none yet)                                              */
} dtk_engine_stats;

int  dtk_engine_create(dtk_ctx* ctx, dtk_engine** out);
int  dtk_engine_create_ops(const dtk_engine_ops* ops, dtk_engine** out);
void dtk_engine_destroy(dtk_engine* e);            /* stops the loop (steps in flight are collected); readers see DTK_SEQ_LEFT     */
const char* dtk_engine_last_error(const dtk_engine* e);   /* text of the device failure that stopped the loop ("" = none)          */
/* tokens that end a reader's burst (the newline table of DetikzifyGenerator.rollout, reference infer/generate.py:262-274) */
int  dtk_engine_set_flush_tokens(dtk_engine* e, const int64_t* ids, int n);
/* option "depth" = steps kept in flight (1 | 2, default 2) */
int  dtk_engine_set_option(dtk_engine* e, const char* name, int value);
/* n sequences are about to join: no step before all of them have, or timeout_ms have passed (rollouts started together move together) */
int  dtk_engine_expect(dtk_engine* e, int n, int timeout_ms);
/* queue a sequence and block until the loop has put its prompt into the slot (DTK_OK) or failed to (error_out).  Joins execute in
 * the order they were queued; dtk_engine_submit queues without waiting (a caller that plans joins against its own bookkeeping —
 * which slot holds which image prefix — queues under the lock that protects the bookkeeping and waits outside it);
 * dtk_engine_await blocks for that join's result.  `j` must stay valid until await returns; every ticket must be awaited once. */
int  dtk_engine_join(dtk_engine* e, dtk_join* j);
int  dtk_engine_submit(dtk_engine* e, dtk_join* j, uint64_t* ticket_out);
int  dtk_engine_await(dtk_engine* e, uint64_t ticket);
/* block until slot's sequence has undelivered flushed tokens, has ended, or timeout_ms passed (< 0: no timeout); copies up to cap
 * tokens; *state_out = DTK_SEQ_*.  A sequence that ended hands out its remaining tokens first: FINISHED / LEFT is reported
 * together with the last of them.  A device failure returns its error code (text: dtk_engine_last_error). */
int  dtk_engine_read(dtk_engine* e, int slot, int64_t* tokens_out, int cap, int32_t* n_out, int32_t* state_out, int timeout_ms);
/* end slot's sequence now (its undelivered tokens stay readable); the slot can be joined again at once */
int  dtk_engine_leave(dtk_engine* e, int slot);
int  dtk_engine_get_stats(dtk_engine* e, dtk_engine_stats* out);

/* Tuning aids (tools/, bench): time one decode GEMV role (0 qkv, 1 o_proj, 2 gate/up, 3 down,
 * 4 lm_head; 5 / 6 = the batched gate/up kernel / its LDS-DMA twin with parts switched off, tools/probe_batch.py) in kernel
 * variant `variant` over all layers with HIP events (clobbers the decode state); select the variant the decode step uses for a
 * role: epi 1 = residual roles (down, and o_proj unless slot 5 is set), 2 qkv, 3 gate/up, 4 lm_head, 5 = o_proj alone (-1: as
 * epi 1), 6 = o_proj with the attention-partials prologue.  Variant 0 = the measured default of the model's width;
 * dtk_bench_gemv variant 0xff = whatever the decode step itself launches for that role (bench.py's roofline leg). */
int  dtk_bench_gemv(dtk_ctx* ctx, int role, int variant, int reps, float* avg_us);
int  dtk_set_gemv_variant(dtk_ctx* ctx, int epi, int variant);
/* Tuning switches; every default is the measured-best setting.  Single-sequence decode: "attn_threads" (0 =
 * contiguous key range per split | 256 | 512 | 1024 = tile-interleaved splits), "attn_splits" (1..16), "attn_combine" (0 = the
 * split partials are reduced in o_proj's prologue, 1 = by the last-arriving split block, 2 = by an own kernel),
 * "attn_full_max" (contexts below it use the one-block-per-head kernel).  Batched decode: "tail_threads" (64 | 128 | 256 | 512) and "prefix_mfma" (1: the prefix a group of <= 16 forked slots
 * shares — same source slot, same length — is scored once for the group on the matrix cores by k_attn_prefix_g, the slots' private
 * keys per slot; 0 = every slot walks its whole context) default by the CONTEXT's size: 128 / 1 with 64 decoding slots, 256 / 0 below;
 * "pfx_splits" (1..4 key splits of that kernel, default 4), "gemv_bus" (64-slot qkv / gate-up by k_gemv_bus — a block per CU whose 8
 * waves are the 8 K slices, every operand straight from memory into the wave's registers, the slice sums met in LDS once: 0 off, 128 = the measured
 * default per role and weight format, else bit 0 qkv, bit 1 gate/up), "gemv_bc" (64-slot qkv / gate-up / lm_head by k_gemv_bc — a compute wave per 16-slot
 * column tile, x from L2 into registers, the weights through an LDS ring: 0 off, 128 = the measured default per role and weight format, else bit 0 qkv, bit 1
 * gate/up, bit 2 lm_head, bits 4..6 units per block), "gemv_b_wide" (0..6: row tiles per block), "gemm_b" (0 = x fragments in
 * registers, 1..4 = x through LDS by LDS-DMA), "gemv_bx" (0 off, 1 = x once per CU for gate/up + lm_head at 49..64 slots, 2..4 =
 * forced units per block), "gemv_bk" (K split across CUs for o_proj / down; slower, off), "gqa_fused" (GQA models: 0 = an attention block per query
 * head, 1 = per K/V head, 2 = per pair of query heads), "share_prefix_reads", "resid_split" (o_proj / down: two row tiles x half of
 * the slots per block), "resid_kparts" (o_proj / down at 49..64 slots: K-slice partials stored, reduced by the RMSNorm kernel that
 * follows), "gemv_bkl" (that kernel with LDS-DMA operand rings), "gemv_bl" (bit 0: loader-wave kernel for gate/up + lm_head, bit 1:
 * qkv by pair units, bit 2: fp8 weights too, bit 3 / 4: qkv as a RoPE pair unit + a V row tile per block, bit 5: fp8 weights at K = 4096 through registers), "gemv_xw" (x fragments
 * by an extra wave's ordinary loads instead of LDS-DMA), "attn_nt" (non-temporal K / V loads), "mv_slots" (0..4: contexts with at most
 * that many + 1 slots decode with the multi-vector kernels; 0 = the MFMA kernels for every context), "mv_tail_threads" (256 | 512 |
 * 1024: attention block of that step), "mv_shape_qkv|o|gu|down|lm_head" (-1 = measured default, 0 = the single-sequence block
 * shape, 1..3 = persistent with 8 / 4 / 16 waves).  Prefill / ViT: "attn_impl" (0 auto,
 * 1 VALU, 2 MFMA flash), "gemm_tile" (0 auto, 1 64x64, 2 128x64, 3 128x128, 4 64x32, 5 32x32), "gemm_bk" (64 | 128), "gemm_stages"
 * (1..4), "gemm_impl" (0 register-staged, 1 LDS-DMA fragment order, 2 128x128 LDS-DMA row order, 3 auto), "gemm_ring" (2..4),
 * "gemm_glds_min_tiles".  fp8 models: "act_fp8" (0 = default: bf16 activations, the fp8 weights widened in registers; 1 = OPT-IN: the MFMA-family
 * step runs on the fp8 matrix cores with MXFP8 activations — 15-26 % shorter steps, logits ~0.12 rel-L2 from the bf16-activation step:
 * tests/test_gpu_parity_batched.py::test_mxfp8_activations_against_bf16_activations asserts <= 0.20), "mx_nc_qkv" / "mx_nc_gu" / "mx_nc_lm_head" (0..4 compute waves per block of its unit kernel; 0 = from the CU count).
 * DESIGN.md 3.4 has the defaults and what each switch measured.
 * Diagnostic: "vit_feature_layer" (0..depth-1) = the block whose normed output dtk_vit_encode returns as features (tests walk
 * the tower block by block with it); every cached image prefix is dropped.
 * Decoder prefill: "prefill_sk" (1 = roles sliced along K by their weight shape, the default; 0 = one-chain GEMMs — another fp32
 * summation order, cached prefixes are dropped; 2 / 4 / 8 = a cap on the slices), and three bit-identical choices: "gemm_wt" (k_gemm_g3's W stage from the
 * fragment-major weight copy), "gemm_epi_direct" (k_gemm_g3 stores from the accumulator layout instead of through LDS), "qkv_rope_fused" (the
 * sliced q/k/v role reduces inside the RoPE + KV-append kernel), "swiglu_fused" (SiLU*mul is the epilogue of the gate/up GEMM), "gemm_sk_tile" (0 = 256 x 128, 1 = 128 x 256, 2 = by M).  "sk_sl_min_rows": above this many prefill rows a sliced role is ONE launch that folds
 * its slices in registers (k_gemm_g3<.., SL>) instead of (tile, slice) blocks + reduction per 512 rows; bit-identical.
 * The environment variable DTK_OPTIONS="name=value,name=value" applies the same switches at dtk_create.
 * SCOPE: the switches that select a kernel VARIANT ("gemv_*", "resid_*", "gemm_*", "mx_*", "attn_impl") are process-wide — they live in the
 * launchers, not in the context: a later context of the same process inherits what an earlier one set, and an A/B inside one process must set
 * the switch on both sides.  The per-context ones: "act_fp8", "prefix_mfma", "tail_threads", "pfx_splits", "share_prefix_reads", "mv_slots",
 * "attn_threads" / "attn_splits" / "attn_combine", "vit_feature_layer", "gemm_naive", "resid_kparts", "prefill_sk", "qkv_rope_fused", "swiglu_fused". */
int  dtk_set_option(dtk_ctx* ctx, const char* name, int value);

/* Op-level entry points used by the parity tests (tests/): run ONE kernel of the
 * hot path on host buffers.  All matrices row-major; bf16 as uint16.  They exist so a
 * failing kernel can be isolated on the GPU box; the product path never calls them. */
#define DTK_EPI_NONE 0
#define DTK_EPI_BIAS 1        /* + bias[n]                         */
#define DTK_EPI_GELU 2        /* gelu(acc + bias)                   */
#define DTK_EPI_RESIDUAL 4    /* bf16(acc + bias) + residual[m,n]   */
#define DTK_GEMM_NAIVE 256    /* use the non-MFMA reference kernel  */
#define DTK_GEMM_WT 512       /* also hand the kernel W as fragment-major tiles (what the decoder prefill does): same result, bit for bit */
#define DTK_GEMM_SL 1024      /* with K slices: ONE launch that folds the slices in registers (what large M takes) instead of (tile, slice) blocks + reduction */
#define DTK_GEMM_KSLICES_SHIFT 12   /* flags |= S << 12 (S = 1..8): the sliced-K family of the decoder prefill — K's 64-wide k-tiles cut into S
                                     * runs, a chain per run, the runs' sums added in order (fp32); either kernel (MFMA / naive) */
/* C[M,N] = A[M,K] . W[N,K]^T (+epilogue) */
int  dtk_op_gemm(dtk_ctx* ctx, const uint16_t* A, const uint16_t* W, const uint16_t* bias,
                 const uint16_t* residual, int M, int N, int K, int flags, uint16_t* C);
/* mode 0: y = W.x ; mode 1: y = W.rmsnorm(x, norm_w) ; fp32 result of the bf16-rounded output */
int  dtk_op_gemv(dtk_ctx* ctx, const uint16_t* W, const uint16_t* x, const uint16_t* norm_w,
                 int N, int K, int mode, float eps, uint16_t* y);
/* the multi-vector GEMV of the <= 4-slot step on nb (1, 2, 4) row-major vectors X[nb][K] -> Y[nb][N]; per vector bit-identical
 * to dtk_op_gemv */
int  dtk_op_gemv_mv(dtk_ctx* ctx, const uint16_t* W, const uint16_t* X, const uint16_t* norm_w,
                    int N, int K, int mode, float eps, int nb, uint16_t* Y);
/* The fp8 matrix-core GEMVs of the batched step of an fp8 model (csrc/kernels_batch_mx.hip; BASELINE config 5's "CDNA4 fp8 MFMA"):
 * W8 [N][K] e4m3 bytes + wscale [N] (per-row power of two); X [nslots][K] bf16 rows, quantised to MXFP8 in groups of G = 32 | 16
 * consecutive k by the step's own quantiser; nslots = 16 | 32 | 64.  mode 0: unit kernel with the logits epilogue (G = 32):
 * Y [nslots][N] = bf16-rounded sums; mode 1: the K-slice kernel of the N = d roles: Y = the 8 slice partials added in order (fp32);
 * mode 2: unit kernel with the SwiGLU epilogue (N = 2 ff): y8_out / ys_out = the activation as MXFP8 groups of 16 (64 ff bytes /
 * ceil(ff / 256) KiB).  x8_out (64 K bytes) / xs_out (ceil(K / (16 G)) KiB), optional: the quantised input in the kernels' order. */
/* byte offsets of value k of `slot` and of its group's E8M0 scale in the MXFP8 activation buffers (G = 32 | 16); host arithmetic only */
int  dtk_mx_layout(int G, int slot, int k, int64_t* data_off, int64_t* scale_off);
int  dtk_op_gemv_mx(dtk_ctx* ctx, const uint8_t* W8, const float* wscale, const uint16_t* X, int N, int K, int G, int nslots, int mode,
                    float* Y, uint8_t* x8_out, uint8_t* xs_out, uint8_t* y8_out, uint8_t* ys_out);
/* softmax(Q K^T * scale [+ causal mask with q_offset]) V, heads-major [H][T][hd] bf16 */
int  dtk_op_attention(dtk_ctx* ctx, const uint16_t* Q, const uint16_t* K, const uint16_t* V,
                      int H, int Tq, int Tk, int hd, int causal, int q_offset, uint16_t* O);
int  dtk_op_layernorm(dtk_ctx* ctx, const uint16_t* X, const uint16_t* w, const uint16_t* b,
                      int M, int D, float eps, uint16_t* Y);
/* run the sampler on host logits with the context's sampling config; step = draw index */
int  dtk_op_sample(dtk_ctx* ctx, const float* logits, int V, int step, int64_t* token_out,
                   float* filtered_probs_out);

#ifdef __cplusplus
}
#endif
#endif /* DTK_H */
