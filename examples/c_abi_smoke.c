/* examples/c_abi_smoke.c — the C ABI of include/dtk.h from plain C, no Python: a two-layer decoder of the detikzify-cl-7b width with fp8
 * weights (BASELINE config 5's model, cut to two layers; seeded synthetic weights) and 17 slots; three text-only prompts are prefilled,
 * four batched decode steps run, and the program prints the tokens and which kernel family decoded them (dtk_stats).
 *   gcc -O2 -Iinclude examples/c_abi_smoke.c -o build/c_abi_smoke -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,'$ORIGIN/../detikzify_amd/lib'
 *   build/c_abi_smoke            # on an MI355X; exit code 0 = every call returned DTK_OK
 * This is the call sequence a cgo / JNI / N-API binding of the reference would make (INTEGRATION.md B shows the ctypes form). */
#include <stdio.h>
#include <string.h>
#include "dtk.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DTK_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dtk_last_error(ctx)); return 1; } } while (0)

int main(void) {
  dtk_ctx* ctx = NULL;
  dtk_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.hidden = 4096; cfg.layers = 2; cfg.heads = 32; cfg.head_dim = 128; cfg.ffn = 11008; cfg.vocab = 32024; cfg.max_positions = 128;
  cfg.rms_eps = 1e-6f; cfg.rope_theta = 10000.f; cfg.rope_factor = 1.f;
  cfg.vit_dim = 1152; cfg.vit_depth = 1; cfg.vit_heads = 16; cfg.vit_mlp = 4304; cfg.vit_patch = 14; cfg.vit_image = 384;
  cfg.vit_feature_layer = 0; cfg.vit_ln_eps = 1e-6f; cfg.concat_patches = 3; cfg.image_token_id = 1;
  cfg.reserved[0] = 17;      /* batch slots: slots 0..15 decode (one 16-slot MFMA tile) */
  cfg.reserved[1] = 1;       /* fp8 e4m3 decoder weights */
  if (dtk_abi_version() != DTK_ABI_VERSION || dtk_abi_struct_size(0) != (int)sizeof(dtk_config) || dtk_abi_struct_size(2) != (int)sizeof(dtk_stats)) {
    fprintf(stderr, "header / library mismatch\n"); return 1;
  }
  if (dtk_create(&cfg, 0, &ctx) != DTK_OK) { fprintf(stderr, "dtk_create: %s\n", dtk_last_error(NULL)); return 1; }
  CHECK(dtk_fill_synthetic(ctx, 4321));
  dtk_sampling greedy;
  memset(&greedy, 0, sizeof greedy);
  greedy.temperature = 1.f; greedy.top_p = 1.f;
  int32_t active[DTK_MAX_BATCH] = {0};
  for (int s = 0; s < 3; ++s) {
    int64_t ids[12];
    for (int t = 0; t < 6 + 3 * s; ++t) ids[t] = 3 + (int64_t)((t * 7919 + s * 104729) % 30000);
    CHECK(dtk_set_sampling_slot(ctx, s, &greedy));
    CHECK(dtk_prefill_slot(ctx, s, ids, 6 + 3 * s, NULL, 0, 0, NULL));
    active[s] = 1;
  }
  for (int step = 0; step < 4; ++step) {
    int64_t tok[DTK_MAX_BATCH];
    CHECK(dtk_decode_batch_launch(ctx, active));
    CHECK(dtk_decode_batch_wait(ctx, tok));
    printf("step %d: tokens %lld %lld %lld\n", step, (long long)tok[0], (long long)tok[1], (long long)tok[2]);
  }
  dtk_stats st;
  CHECK(dtk_get_stats(ctx, &st));
  printf("last step: %u-slot kernels, fp8 matrix cores: %u, device errors: %u, context lengths %d %d %d\n", st.last_batch_step_slots,
         st.last_batch_step_fp8_mfma, st.device_errors, dtk_context_len_slot(ctx, 0), dtk_context_len_slot(ctx, 1), dtk_context_len_slot(ctx, 2));
  const int ok = st.last_batch_step_slots == 16 && st.last_batch_step_fp8_mfma == 0 /* MXFP8 activations are opt-in (act_fp8) */ && st.device_errors == 0 && dtk_context_len_slot(ctx, 2) == 16;
  dtk_destroy(ctx);
  printf(ok ? "c_abi_smoke ok\n" : "c_abi_smoke: unexpected state\n");
  return ok ? 0 : 1;
}
