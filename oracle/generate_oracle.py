"""The autoregressive driver, restated (oracle — see oracle/__init__.py).

Follows /root/reference/models/vaura_model.py:
  generate()            :410-597   gen_codes = -1, prompt fill, build_pattern_sequence(special=d_codebook),
                                   loop offset in [first_step(start_offset), S), revert(special=-1), slice [:T]
  _sample_next_token()  :775-827   CFG batch doubling [cond; null], last position only
  fix-up                :536-544   invalid mask -> special token; never overwrite known tokens

``mode="full"`` feeds the whole prefix each step like the reference (no cache);
``mode="cached"`` is the same loop over ``CachedDecoder`` (prompt positions are pre-filled).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import pattern_oracle as po
from . import sampling_oracle as so
from .decoder_oracle import CachedDecoder, DecoderOracle


@torch.no_grad()
def generate(dec: DecoderOracle, cond: torch.Tensor, max_new_tokens: int, *, prompt: Optional[torch.Tensor] = None,
             use_sampling: bool = False, temp: float = 1.0, top_k: int = 0, top_p: float = 0.0,
             cfg_scale: float = 1.0, noise: Optional[torch.Tensor] = None, mode: str = "cached",
             special: int = 1024, trace: Optional[dict] = None) -> torch.Tensor:
    """cond (B, Tv, 768) fp32; prompt (B, K, Tp) int64 or None; noise (steps, B*K, V) or None.
    Returns codes (B, K, max_new_tokens) int64."""
    B = cond.shape[0]
    K, T = dec.K, max_new_tokens
    Tp = 0 if prompt is None else prompt.shape[-1]
    assert Tp < T
    codes = np.full((B, K, T), -1, dtype=np.int64)
    if Tp:
        codes[..., :Tp] = prompt.numpy()
    seq_np, _, mask = po.build_sequence(codes, special)
    seq = torch.from_numpy(seq_np)
    S = seq.shape[-1]
    start = po.first_step_with_timestep(K, T, Tp)
    use_cfg = cfg_scale > 1.0
    cond_all = torch.cat([cond, dec.null_condition(cond)], dim=0) if use_cfg else cond

    cache = None
    if mode == "cached":
        cache = CachedDecoder(dec, cond_all, S)
        for p in range(start - 1):  # positions whose successors are already known (prompt)
            tk = seq[:, :, p]
            cache.step(torch.cat([tk, tk], 0) if use_cfg else tk, want_logits=False)

    for n, offset in enumerate(range(start, S)):
        if mode == "cached":
            tk = seq[:, :, offset - 1]
            logits = cache.step(torch.cat([tk, tk], 0) if use_cfg else tk)
        else:
            cur = seq[:, :, :offset]
            logits = dec.forward_full(cur.repeat(2, 1, 1) if use_cfg else cur, cond_all)[:, :, -1]
        if use_cfg:
            logits = so.cfg_mix(logits, cfg_scale)
        if trace is not None:
            trace.setdefault("logits", {})[offset] = logits.clone()
        nz = None if noise is None else noise[n]
        nt = so.next_token(logits, use_sampling=use_sampling, temp=temp, top_k=top_k, top_p=top_p, noise=nz)[..., 0]
        valid = torch.from_numpy(mask[:, offset])[None, :].expand(B, -1)
        nt = torch.where(valid, nt, torch.full_like(nt, special))
        cur_col = seq[:, :, offset]
        seq[:, :, offset] = torch.where(cur_col == -1, nt, cur_col)

    assert not (seq == -1).any()
    out, _, omask = po.revert_sequence(seq.numpy(), T, -1)
    assert omask.all() and (out >= 0).all() and (out <= special).all()
    return torch.from_numpy(out)
