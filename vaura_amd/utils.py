"""Plugin factory + sampling helpers with the reference's names and semantics
(/root/reference/utils/utils.py:11-22, 139-196).  The sampling functions are the host-visible
form of the device sampler (``vaura_sample``): they take/return torch tensors on the HIP device."""
from __future__ import annotations

import importlib
from typing import Optional

import torch

from . import ops


def get_obj_from_str(string: str, reload: bool = False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    """``{"target": "pkg.mod.Class", "params": {...}}`` -> ``Class(**params)`` (utils/utils.py:19-22)."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def _draw_noise(shape, device, noise: Optional[torch.Tensor]):
    """Exp(1) draws in the reference's order: ``empty(rows, V).exponential_(1)`` from torch's global CPU
    generator — what ``torch.multinomial`` consumes on the reference's CPU path."""
    if noise is not None:
        return noise.to(device)
    rows = 1
    for s in shape[:-1]:
        rows *= s
    return torch.empty(rows, shape[-1]).exponential_(1).to(device)


def sample_from_logits(logits: torch.Tensor, *, use_sampling: bool, temp: float = 1.0, top_k: int = 0, top_p: float = 0.0,
                       cfg_scale: float = 1.0, noise: Optional[torch.Tensor] = None, batch: Optional[int] = None):
    """logits (rows, K, V) -> tokens (batch, K, 1) int64: the body of ``_sample_next_token`` after the sampler
    call (models/vaura_model.py:807-825) on the HIP device.  ``noise`` (batch*K, V) Exp(1) or None (drawn from
    torch's global CPU generator, like the reference's CPU path)."""
    rows, K, V = logits.shape
    B = batch if batch is not None else (rows // 2 if cfg_scale > 1.0 else rows)
    nz = None
    if use_sampling and temp > 0.0:
        nz = _draw_noise((B, K, V), logits.device, noise).reshape(1, B * K, V)
    return ops.sample(logits, B, use_sampling=use_sampling, temp=temp, top_k=top_k, top_p=top_p, cfg_scale=cfg_scale, noise=nz)


# ---- the reference's own names (utils/utils.py:139-196): probabilities in, token ids out, on the HIP device
def _sample_probs(probs: torch.Tensor, top_k: int, top_p: float, noise: Optional[torch.Tensor]) -> torch.Tensor:
    lead, V = tuple(probs.shape[:-1]), probs.shape[-1]
    rows = 1
    for d in lead:
        rows *= d
    nz = _draw_noise((rows, V), probs.device, noise).reshape(1, rows, V)
    tok = ops.sample(probs.reshape(rows, 1, V), rows, use_sampling=True, temp=1.0, top_k=top_k, top_p=top_p, noise=nz,
                     input_is_probs=True)
    return tok.reshape(*lead, 1)


def multinomial(input: torch.Tensor, num_samples: int = 1, replacement: bool = False, *, generator=None,
                noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``utils.multinomial`` (utils/utils.py:139-160) for ``num_samples=1``: one draw per row of the last dimension,
    ``argmax(p / Exp(1))`` like ``torch.multinomial``.  ``noise`` (rows, V): recorded Exp(1) draws; else they are taken from
    torch's global CPU generator in the reference's order (a ``generator`` object cannot be forwarded to the device sampler)."""
    if num_samples != 1 or generator is not None:
        raise NotImplementedError("the device sampler draws one token per row from torch's global CPU stream (or recorded noise)")
    return _sample_probs(input, 0, 0.0, noise)


def sample_top_k(probs: torch.Tensor, k: int, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``utils.sample_top_k`` (utils/utils.py:163-178): keep ``probs >= k-th largest`` (ties kept), renormalise, draw.
    (The reference also rewrites ``probs`` in place; this one leaves its argument alone.)"""
    return _sample_probs(probs, int(k), 0.0, noise)


def sample_top_p(probs: torch.Tensor, p: float, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``utils.sample_top_p`` (utils/utils.py:181-196): sort descending, keep while ``cumsum - p_i <= p``, renormalise, draw in
    sorted space, map back."""
    return _sample_probs(probs, 0, float(p), noise)
