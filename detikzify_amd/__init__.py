"""
detikzify_amd — MI355X (gfx950) native image->TikZ decoder behind DeTikZify's inference API.

Layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/dtk.h) -> lib/libdtk_hip.so
  _lib.py    ctypes binding of the C ABI (fails loudly when the library is missing)
  model/     load(), config, processor, tokenizer, the HF-shaped model object over the C ABI
  infer/     DetikzifyGenerator / DetikzifyPipeline (+ TikzGenerator alias), TikzDocument
  mcts/      UCT tree search used by the generator
  evaluate/  SelfSim (ImageSim) reward on the model's own vision tower
  util/      streamers, stopping criteria, image helpers, subprocess helper
  dist.py    one-process-per-GPU sharding of rollouts / images, RCCL gather of TikZ strings
"""
__version__ = "0.1.0"
