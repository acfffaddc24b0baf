"""
oracle/ — CPU restatement of the DeTikZify image->TikZ hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package; the product (detikzify_amd/) never does and fails loudly without its HIP library.

What is restated, and from where (reference = potamides/DeTikZify @ 2025-08-24):

* glue (vision features -> patch concat -> mm_projector -> embedding splice -> LLaMA ->
  lm_head): detikzify/model/v1/modeling_detikzify.py:63-72,132-137,144-200,218-257.
* the arithmetic itself lives in third-party wheels that are NOT under /root/reference:
    - transformers (pinned ~=4.52.4, pyproject.toml:12; 5.15.0 installed here):
      LlamaModel / LlamaRMSNorm / rotary / LlamaMLP (modeling_llama.py), GenerationMixin._sample
      and the logits processors (generation/utils.py, generation/logits_process.py);
    - timm (pinned ~=1.0.11, pyproject.toml:46-48; not installed): VisionTransformer
      `vit_so400m_patch14_siglip_384` incl. AttentionPoolLatent ('map' pooling).
  Their published algorithms are restated in llama.py / vit.py / sampling.py.

Pinning: the reference ships no tests, golden vectors or fixtures and its package cannot be
imported as a whole in this container (needs py3.11, transformers 4.52, timm, torchmetrics, ...), so the
restatement is pinned against what CAN run here (tests/golden/make_golden.py generates the fixtures,
tests/test_oracle_golden.py and tests/test_host_logic.py check against them):
  * the reference's OWN model code, both families — detikzify/model/modeling_detikzify.py (v2) and
    detikzify/model/v1/modeling_detikzify.py (v1; timm.create_model replaced by a timm-shaped shim over HF's
    SiglipVisionModel) loaded file by file and run on the CPU at toy size with the seeded synthetic weights: the
    oracle reproduces prefill and 16 cached greedy steps (logits to 6e-7, tokens and error messages identical);
  * the installed HuggingFace LlamaForCausalLM (MHA + linear rope, GQA + llama3 rope), SiglipVisionModel
    (architecture stand-in for the timm ViT of v1) and HF logits processors;
  * the reference's own host code: detikzify.mcts (imports cleanly), infer/generate.py, infer/tikz.py,
    util/image.py and the v2 processor, each executed with stubs for the packages that are absent.
Status for the v1 tower against timm itself and for every real checkpoint: PARITY UNPINNED (timm and weights
are absent offline; the GELU flavour of the timm tower is a config switch, see vit.py).

Precision policy ("bf16" mode, the product's): every tensor HF/timm materialise in bf16 is
rounded to bf16 at the same point; contractions accumulate in fp32; attention follows the fused
(SDPA / flash) semantics: fp32 scores and probabilities, one rounding of the head output.
"fp32" mode applies no rounding and is the mathematical definition.
"""
