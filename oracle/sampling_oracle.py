"""Next-token selection, fp32 torch-CPU (oracle — see oracle/__init__.py).

Follows /root/reference:
  models/vaura_model.py:807-825   last-position logits, CFG mix ``u + (c-u)*scale``, temperature
                                  softmax, dispatch: top_p > 0 wins over top_k; greedy iff
                                  ``not use_sampling or temp <= 0`` (argmax of the *logits*)
  utils/utils.py:163-178          sample_top_k: threshold = k-th largest prob, keep ``>=`` (ties kept),
                                  renormalise, multinomial
  utils/utils.py:181-196          sample_top_p: sort desc, cumsum, drop where cumsum - p_i > p,
                                  renormalise, multinomial in sorted space, gather back
  utils/utils.py:139-160          multinomial: flatten to (rows, V), ``torch.multinomial(., 1)``

``torch.multinomial(p, 1)`` (no replacement / one draw) is implemented by ATen as
``argmax(p / q)`` with ``q = empty_like(p).exponential_(1)`` drawn from the generator
(aten/src/ATen/native/Sampling / TensorAdvancedIndexing "fast path"; verified equal to the call
itself on torch 2.10 CPU by tests/golden/make_golden.py).  Taking ``q`` as an explicit input is
what lets a GPU implementation be compared token-for-token.
"""
from __future__ import annotations

from typing import Optional

import torch


def cfg_mix(logits: torch.Tensor, cfg_scale: float) -> torch.Tensor:
    """logits (2B, K, V) stacked [cond; uncond] -> (B, K, V); vaura_model.py:810-813."""
    half = logits.shape[0] // 2
    c, u = logits[:half], logits[half:]
    return u + (c - u) * cfg_scale


def draw(probs: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """argmax(p / q) over the last dim, keepdim — the multinomial fast path."""
    return torch.argmax(probs / noise, dim=-1, keepdim=True)


def top_k_filter(probs: torch.Tensor, k: int) -> torch.Tensor:
    kth = torch.topk(probs, k, dim=-1).values[..., [-1]]
    p = probs * (probs >= kth).float()
    return p / p.sum(dim=-1, keepdim=True)


def top_p_sorted(probs: torch.Tensor, p: float):
    ps, order = torch.sort(probs, dim=-1, descending=True)
    cs = torch.cumsum(ps, dim=-1)
    ps = ps * (~(cs - ps > p)).float()
    return ps / ps.sum(dim=-1, keepdim=True), order


def next_token(logits: torch.Tensor, *, use_sampling: bool, temp: float, top_k: int, top_p: float,
               noise: Optional[torch.Tensor]) -> torch.Tensor:
    """logits (B, K, V) already CFG-mixed -> tokens (B, K, 1) int64.  ``noise`` (B*K, V) Exp(1)."""
    if not (use_sampling and temp > 0.0):
        return torch.argmax(logits, dim=-1, keepdim=True)
    B, K, V = logits.shape
    probs = torch.softmax(logits / temp, dim=-1)
    q = noise.reshape(B, K, V)
    if top_p > 0.0:
        ps, order = top_p_sorted(probs, top_p)
        return torch.gather(order, -1, draw(ps, q))
    if top_k > 0:
        return draw(top_k_filter(probs, top_k), q)
    return draw(probs, q)
