/* tools/probe/step_bench.c — decode steps of a full-size model through the C ABI, without Python (a fresh GPU box pays 1-2 minutes
 * for its first `import torch`; this starts in a second).  For every option set on the command line ("" = the defaults) a FRESH
 * context runs the same greedy decode, times it, and hashes the logits — kernel variants that claim to be bit-identical must print
 * the same hash; rocprofv3 --kernel-trace --stats around it gives the per-kernel times of exactly this step.
 *   gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,'$ORIGIN/../../detikzify_amd/lib'
 *   tools/probe/step_bench "" "prefix_mfma=1" "prefix_mfma=1,pfx_splits=4"
 * Environment:
 *   STEP_BENCH_MODEL   cl-7b-fp8 (default) | ds-7b | ds-1.3b (bf16): the presets of detikzify_amd/model/config.py
 *   STEP_BENCH_SLOTS   64 (default: the batched step, slots forked from one prefix) | 32 | 16 | 1 (the single-sequence step, dtk_decode_*)
 *   STEP_BENCH_IMAGES  prefix sources the slots are forked from, round-robin (default 1; 8 = BASELINE config 5's shape on one GPU)
 *   STEP_BENCH_WARM    untimed steps before the timed ones (default 4): > 4 grows every slot's PRIVATE context first
 *   STEP_BENCH_STEPS   timed steps (default 48)
 *   STEP_BENCH_LAYERS  decoder layers (default: the model's) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "dtk.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DTK_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dtk_last_error(ctx)); return 1; } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

static int run_variant(const char* model, const char* opts) {
  dtk_ctx* ctx = NULL;
  dtk_config cfg;
  memset(&cfg, 0, sizeof cfg);
  const int small = !strcmp(model, "ds-1.3b"), fp8 = !strcmp(model, "cl-7b-fp8");
  const int layers = env_int("STEP_BENCH_LAYERS", small ? 24 : 32), steps = env_int("STEP_BENCH_STEPS", 48), warm = env_int("STEP_BENCH_WARM", 4), T = 243;
  int nslots = env_int("STEP_BENCH_SLOTS", 64), images = env_int("STEP_BENCH_IMAGES", 1);
  if (nslots != 1 && nslots != 16 && nslots != 32 && nslots != 64) { fprintf(stderr, "STEP_BENCH_SLOTS must be 1, 16, 32 or 64\n"); return 1; }
  if (images < 1 || images > 8) images = 1;
  cfg.hidden = small ? 2048 : 4096; cfg.layers = layers; cfg.heads = small ? 16 : 32; cfg.head_dim = 128; cfg.ffn = small ? 5504 : 11008;
  cfg.vocab = fp8 ? 32024 : 32256; cfg.max_positions = T + warm + steps + 8 > 512 ? 1024 : 512;
  cfg.rms_eps = fp8 ? 1e-5f : 1e-6f; cfg.rope_theta = fp8 ? 1000000.f : 100000.f; cfg.rope_factor = fp8 ? 1.f : 4.f;
  cfg.vit_dim = 1152; cfg.vit_depth = 1; cfg.vit_heads = 16; cfg.vit_mlp = 4304; cfg.vit_patch = 14; cfg.vit_image = 384;
  cfg.vit_feature_layer = 0; cfg.vit_ln_eps = 1e-6f; cfg.concat_patches = 3; cfg.image_token_id = 1;
  cfg.reserved[0] = nslots == 1 ? 0 : nslots + images; cfg.reserved[1] = fp8;
  double t0 = now();
  if (dtk_create(&cfg, 0, &ctx) != DTK_OK) { fprintf(stderr, "dtk_create: %s\n", dtk_last_error(NULL)); return 1; }
  CHECK(dtk_fill_synthetic(ctx, 1234));
  CHECK(dtk_synchronize(ctx));
  const double t_load = now() - t0;
  char buf[512];
  strncpy(buf, opts, sizeof buf - 1); buf[sizeof buf - 1] = 0;
  for (char* tok = strtok(buf, ","); tok; tok = strtok(NULL, ",")) {
    char* eq = strchr(tok, '=');
    if (!eq) continue;
    *eq = 0;
    CHECK(dtk_set_option(ctx, tok, atoi(eq + 1)));
  }
  dtk_sampling greedy;
  memset(&greedy, 0, sizeof greedy);
  greedy.temperature = 1.f; greedy.top_p = 1.f;
  static int64_t ids[512];
  static float logits[32256];
  unsigned long long h = 1469598103934665603ull;
  double ms = 0;
  long long last = -1;
  if (nslots == 1) {
    for (int t = 0; t < T; ++t) ids[t] = 3 + (int64_t)((t * 7919 + 13) % 30000);
    int64_t tok = 0;
    CHECK(dtk_set_sampling(ctx, &greedy));
    CHECK(dtk_prefill(ctx, ids, T, NULL, 0, 0, NULL));
    for (int i = 0; i < warm; ++i) CHECK(dtk_decode(ctx, &tok));
    CHECK(dtk_synchronize(ctx));
    t0 = now();
    CHECK(dtk_decode_launch(ctx));
    for (int i = 1; i < steps; ++i) { CHECK(dtk_decode_launch(ctx)); CHECK(dtk_decode_wait(ctx, &tok)); }     /* one step ahead, like generate() */
    CHECK(dtk_decode_wait(ctx, &tok));
    CHECK(dtk_synchronize(ctx));
    ms = 1e3 * (now() - t0) / steps;
    last = (long long)tok;
    CHECK(dtk_get_logits(ctx, logits));
    const unsigned char* p = (const unsigned char*)logits;
    for (size_t i = 0; i < (size_t)cfg.vocab * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
  } else {
    int32_t active[DTK_MAX_BATCH];
    memset(active, 0, sizeof active);
    for (int im = 0; im < images; ++im) {        /* one prefix-cache slot per image (slots nslots ..): a different prompt each */
      for (int t = 0; t < T; ++t) ids[t] = 3 + (int64_t)((t * 7919 + 13 + 101 * im) % 30000);
      CHECK(dtk_set_sampling_slot(ctx, nslots + im, &greedy));
      CHECK(dtk_prefill_slot(ctx, nslots + im, ids, T, NULL, 0, 0, NULL));
    }
    for (int s = 0; s < nslots; ++s) { CHECK(dtk_set_sampling_slot(ctx, s, &greedy)); CHECK(dtk_kv_fork(ctx, nslots + s % images, s, T)); active[s] = 1; }
    int64_t tok[DTK_MAX_BATCH];
    for (int i = 0; i < warm; ++i) { CHECK(dtk_decode_batch_launch(ctx, active)); CHECK(dtk_decode_batch_wait(ctx, tok)); }
    CHECK(dtk_synchronize(ctx));
    t0 = now();
    CHECK(dtk_decode_batch_launch(ctx, active));
    for (int i = 1; i < steps; ++i) { CHECK(dtk_decode_batch_launch(ctx, active)); CHECK(dtk_decode_batch_wait(ctx, tok)); }
    CHECK(dtk_decode_batch_wait(ctx, tok));
    CHECK(dtk_synchronize(ctx));
    ms = 1e3 * (now() - t0) / steps;
    last = (long long)tok[0];
    const int watch[3] = {0, nslots / 3, nslots - 1};
    for (int w = 0; w < 3; ++w) {
      CHECK(dtk_get_logits_slot(ctx, watch[w], logits));
      const unsigned char* p = (const unsigned char*)logits;
      for (size_t i = 0; i < (size_t)cfg.vocab * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
    }
  }
  dtk_stats st;
  CHECK(dtk_get_stats(ctx, &st));
  printf("[%s] %s, %d layers, %d slot(s), %d image(s), context %d + %d: %.3f ms/step over %d steps (%.1f tok/s); logits hash %016llx; last token of slot 0: %lld; "
         "fp8 matrix cores %u; device errors %u; context + weights %.1f s\n", opts, model, layers, nslots, images, T, warm, ms, steps, 1e3 * nslots / ms, h, last,
         st.last_batch_step_fp8_mfma, st.device_errors, t_load);
  fflush(stdout);
  dtk_destroy(ctx);
  return 0;
}

int main(int argc, char** argv) {
  const char* model = getenv("STEP_BENCH_MODEL") ? getenv("STEP_BENCH_MODEL") : "cl-7b-fp8";
  for (int v = 1; v < (argc > 1 ? argc : 2); ++v)
    if (run_variant(model, argc > 1 ? argv[v] : "")) return 1;
  return 0;
}
