/* tools/probe/step_bench.c — the 64-slot decode step of the full cl-7b fp8 model through the C ABI, without Python (a fresh GPU box pays
 * 1-2 minutes for its first `import torch`; this starts in a second): for every option set on the command line ("" = defaults) the
 * same greedy decode is run and timed, and the logits of three slots are hashed — kernel variants that claim to be bit-identical must
 * print the same hash.
 *   gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,'$ORIGIN/../../detikzify_amd/lib'
 *   tools/probe/step_bench "" "mx_k_overlap=1" "mx_k_tpg4=1" "mx_k_overlap=1,mx_k_tpg4=1" */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "dtk.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DTK_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dtk_last_error(ctx)); return 1; } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
  dtk_ctx* ctx = NULL;
  dtk_config cfg;
  memset(&cfg, 0, sizeof cfg);
  /* STEP_BENCH_MODEL = cl-7b-fp8 (default) | ds-7b (bf16) | ds-1.3b (bf16): the presets of detikzify_amd/model/config.py */
  const char* model = getenv("STEP_BENCH_MODEL") ? getenv("STEP_BENCH_MODEL") : "cl-7b-fp8";
  const int small = !strcmp(model, "ds-1.3b"), fp8 = !strcmp(model, "cl-7b-fp8");
  const int layers = getenv("STEP_BENCH_LAYERS") ? atoi(getenv("STEP_BENCH_LAYERS")) : (small ? 24 : 32), steps = 48, T = 243;
  cfg.hidden = small ? 2048 : 4096; cfg.layers = layers; cfg.heads = small ? 16 : 32; cfg.head_dim = 128; cfg.ffn = small ? 5504 : 11008;
  cfg.vocab = fp8 ? 32024 : 32256; cfg.max_positions = 512;
  cfg.rms_eps = fp8 ? 1e-5f : 1e-6f; cfg.rope_theta = fp8 ? 1000000.f : 100000.f; cfg.rope_factor = fp8 ? 1.f : 4.f;
  cfg.vit_dim = 1152; cfg.vit_depth = 1; cfg.vit_heads = 16; cfg.vit_mlp = 4304; cfg.vit_patch = 14; cfg.vit_image = 384;
  cfg.vit_feature_layer = 0; cfg.vit_ln_eps = 1e-6f; cfg.concat_patches = 3; cfg.image_token_id = 1;
  cfg.reserved[0] = 65; cfg.reserved[1] = fp8;
  double t0 = now();
  if (dtk_create(&cfg, 0, &ctx) != DTK_OK) { fprintf(stderr, "dtk_create: %s\n", dtk_last_error(NULL)); return 1; }
  CHECK(dtk_fill_synthetic(ctx, 1234));
  CHECK(dtk_synchronize(ctx));
  printf("%s, %d layers: context + synthetic weights %.1f s\n", model, layers, now() - t0);
  dtk_sampling greedy;
  memset(&greedy, 0, sizeof greedy);
  greedy.temperature = 1.f; greedy.top_p = 1.f;
  static int64_t ids[512];
  for (int t = 0; t < T; ++t) ids[t] = 3 + (int64_t)((t * 7919 + 13) % 30000);
  static float logits[32256];
  for (int v = 1; v < (argc > 1 ? argc : 2); ++v) {
    const char* opts = argc > 1 ? argv[v] : "";
    char buf[256];
    strncpy(buf, opts, sizeof buf - 1); buf[sizeof buf - 1] = 0;
    for (char* tok = strtok(buf, ","); tok; tok = strtok(NULL, ",")) {
      char* eq = strchr(tok, '=');
      if (!eq) continue;
      *eq = 0;
      CHECK(dtk_set_option(ctx, tok, atoi(eq + 1)));
    }
    int32_t active[DTK_MAX_BATCH];
    CHECK(dtk_set_sampling_slot(ctx, 64, &greedy));
    CHECK(dtk_prefill_slot(ctx, 64, ids, T, NULL, 0, 0, NULL));
    for (int s = 0; s < 64; ++s) { CHECK(dtk_set_sampling_slot(ctx, s, &greedy)); CHECK(dtk_kv_fork(ctx, 64, s, T)); active[s] = 1; }
    int64_t tok[DTK_MAX_BATCH];
    for (int i = 0; i < 4; ++i) { CHECK(dtk_decode_batch_launch(ctx, active)); CHECK(dtk_decode_batch_wait(ctx, tok)); }
    CHECK(dtk_synchronize(ctx));
    t0 = now();
    CHECK(dtk_decode_batch_launch(ctx, active));
    for (int i = 1; i < steps; ++i) { CHECK(dtk_decode_batch_launch(ctx, active)); CHECK(dtk_decode_batch_wait(ctx, tok)); }
    CHECK(dtk_decode_batch_wait(ctx, tok));
    CHECK(dtk_synchronize(ctx));
    const double ms = 1e3 * (now() - t0) / steps;
    unsigned long long h = 1469598103934665603ull;
    const int watch[3] = {0, 21, 63};
    for (int w = 0; w < 3; ++w) {
      CHECK(dtk_get_logits_slot(ctx, watch[w], logits));
      const unsigned char* p = (const unsigned char*)logits;
      for (size_t i = 0; i < (size_t)cfg.vocab * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
    }
    dtk_stats st;
    CHECK(dtk_get_stats(ctx, &st));
    printf("[%s] %.3f ms/step over %d steps; logits hash %016llx; last token of slot 0: %lld; fp8 matrix cores %u; device errors %u\n", opts, ms, steps, h,
           (long long)tok[0], st.last_batch_step_fp8_mfma, st.device_errors);
    /* back to the defaults for the next variant */
    strncpy(buf, opts, sizeof buf - 1);
    for (char* t2 = strtok(buf, ","); t2; t2 = strtok(NULL, ",")) { char* eq = strchr(t2, '='); if (eq) { *eq = 0; CHECK(dtk_set_option(ctx, t2, !strcmp(t2, "act_fp8") ? 1 : (!strcmp(t2, "mx_ring") ? 3 : 0))); } }
  }
  dtk_destroy(ctx);
  return 0;
}
