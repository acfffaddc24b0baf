#!/usr/bin/env bash
# The first time a box has weights:   scripts/real_parity.sh /path/to/nllg-detikzify-ds-7b   (an HF-layout checkpoint directory)
# Device vs the installed HuggingFace / timm classes fed from the same files: vision features, prefill logits, 32 greedy tokens from
# examples/sketch.png, and — v1 — which GELU the timm tower wants.  Prints one REAL-CHECKPOINT PARITY line (tests/real_checkpoint.py).
set -euo pipefail
cd "$(dirname "$0")/.."
DTK_REAL_CKPT="${1:?usage: scripts/real_parity.sh <checkpoint directory>}" python -m pytest tests/test_gpu_real_checkpoint.py -m gpu -s -q -k real_checkpoint_parity
