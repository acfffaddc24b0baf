"""
HF logits processors + sampler restated (test infrastructure, see oracle/__init__.py).

Mirrors transformers generation (un-vendored; installed 5.15.0), as configured by the reference
at detikzify/infer/generate.py:218-227 (bad_words_ids=[[image_token_id]], begin_suppress_tokens=[eos],
temperature/top_p/top_k, do_sample):
  NoBadWordsLogitsProcessor (single-token bad words)   logits_process.py:1395  -> -inf
  SuppressTokensAtBeginLogitsProcessor                   :1860-1866  only when len(input_ids)==begin_index
  TemperatureLogitsWarper                                :300-303    scores / T
  TopKLogitsWarper                                       :573-579    remove scores < k-th largest
  TopPLogitsWarper                                       :528-540    ascending sort, softmax, cumsum,
                                                                      remove cumsum <= 1-p, keep >= 1
  _sample: softmax -> multinomial | argmax               generation/utils.py:2920-2925
`processed_scores` is the literal HF algorithm (sort + cumsum in fp32).  `draw` is the
deterministic inverse-CDF draw the HIP sampler implements (torch.multinomial's RNG stream cannot
be reproduced on a different device, so token parity under sampling is defined against this
counter-based draw): integer probability mass q_i = floor(exp(z_i - zmax) * 2^31), kept set
= {i : mass strictly above z_i < top_p * total}, target = floor(kept_total * r / 2^32) with
r = splitmix64(seed ^ C*(n+1)) >> 32, first index whose running mass exceeds target.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import numpy as np
import torch

M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def rand32(seed: int, n: int) -> int:
    return splitmix64((seed ^ ((0xD1B54A32D192ED03 * (n + 1)) & M64)) & M64) >> 32


def mask_scores(logits: torch.Tensor, bad: Iterable[int], begin: Iterable[int], first: bool,
                always: Iterable[int] = ()) -> torch.Tensor:
    s = logits.clone().float()
    for i in list(bad) + list(always):
        s[i] = float("-inf")
    if first:
        for i in begin:
            s[i] = float("-inf")
    return s


def processed_scores(logits, temperature=1.0, top_k=0, top_p=1.0, bad=(), begin=(), first=False,
                     always=()) -> torch.Tensor:
    """Scores after the HF processor/warper chain (removed tokens = -inf)."""
    s = mask_scores(logits, bad, begin, first, always)
    s = s / temperature
    if top_k and top_k > 0:
        k = min(top_k, s.numel())
        kth = torch.topk(s, k)[0][-1]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(s, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[-1:] = False
        mask = torch.zeros_like(remove).scatter(0, sorted_indices, remove)
        s = s.masked_fill(mask, float("-inf"))
    return s


def greedy(logits, bad=(), begin=(), first=False, always=()) -> int:
    return int(torch.argmax(mask_scores(logits, bad, begin, first, always)))


def integer_masses(logits, temperature, bad=(), begin=(), first=False, always=()):
    z = mask_scores(logits, bad, begin, first, always)
    z = (z * np.float32(1.0 / np.float32(temperature))).float()   # device multiplies by 1/T
    zmax = z.max()
    e = torch.exp(z - zmax).double()
    q = torch.floor(e * 2147483648.0).to(torch.int64)
    return z, q


def kept_mask(z: torch.Tensor, q: torch.Tensor, top_k: int, top_p: float) -> torch.Tensor:
    keep = torch.ones_like(q, dtype=torch.bool)
    if top_k and 0 < top_k < z.numel():
        kth = torch.topk(z, top_k)[0][-1]
        keep &= z >= kth
    if top_p < 1.0:
        qk = torch.where(keep, q, torch.zeros_like(q))
        total = int(qk.sum())
        pq = int(np.float64(np.float32(top_p)) * np.float64(total))
        order = torch.argsort(z, descending=True, stable=True)
        zs, qs, ks = z[order], qk[order], keep[order]
        csum = torch.cumsum(qs, 0) - qs            # mass of the entries sorted before
        # mass strictly above = mass of strictly larger values (ties share the same bound)
        first_of_value = torch.ones_like(zs, dtype=torch.bool)
        first_of_value[1:] = zs[1:] != zs[:-1]
        idx = torch.where(first_of_value, torch.arange(len(zs)), torch.zeros(len(zs), dtype=torch.long))
        idx = torch.cummax(idx, 0)[0]
        above = csum[idx]
        keep_sorted = ks & (above < pq)
        if not bool(keep_sorted.any()):
            keep_sorted[0] = True
        keep = torch.zeros_like(keep).scatter(0, order, keep_sorted)
    return keep


def draw(logits, temperature, top_k, top_p, seed: int, n: int, bad=(), begin=(), first=False,
         always=()) -> Tuple[int, torch.Tensor]:
    """(token, filtered probabilities) of the deterministic sampler for draw index n."""
    z, q = integer_masses(logits, temperature, bad, begin, first, always)
    keep = kept_mask(z, q, top_k, top_p)
    qk = torch.where(keep, q, torch.zeros_like(q))
    kept_total = int(qk.sum())
    target = (kept_total * rand32(seed, n)) >> 32
    run = torch.cumsum(qk, 0)
    tok = int(torch.searchsorted(run, torch.tensor(target, dtype=torch.int64), right=True))
    return tok, (qk.double() / float(kept_total)).float()
