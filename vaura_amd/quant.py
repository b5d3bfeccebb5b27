"""Host-side description of the fp8 weight format (``VAURA_W_FP8``, include/vaura_hip.h).

The device quantiser lives in libvaura_hip.so (``vaura_pack_weight(..., VAURA_W_FP8)``); these few torch
lines state the same rule so that a caller (and the parity tests) can ask "which fp32 matrix does the fp8
model actually multiply by?".  The reference has no fp8 path (BASELINE.json configs[4] is ours), so the
contract is: the fp8 engine generates exactly the tokens the bf16 engine generates for the checkpoint
whose four per-layer matrices are replaced by ``fp8_effective_weight(W)``.

Format: OCP e4m3 (``torch.float8_e4m3fn``), one scale per output row, the smallest power of two with
``max|W[n, :]| <= 448 * scale[n]``.  A power-of-two scale makes ``fp8 * scale`` exact in bf16, so the
dequantised matrix is bf16-representable and the engine's exactness argument carries over.
"""
from __future__ import annotations

import torch

FP8_MAX = 448.0


def fp8_row_scales(w: torch.Tensor) -> torch.Tensor:
    """(N, K) fp32 -> (N) fp32 power-of-two scales."""
    amax = w.detach().float().abs().amax(dim=1)
    m, e = torch.frexp(amax)                       # amax = m * 2^e, m in [0.5, 1);  448 = 0.875 * 2^9
    exp = torch.where(m <= 0.875, e - 9, e - 8)
    scale = torch.ldexp(torch.ones_like(amax), exp)
    return torch.where(amax > 0, scale, torch.ones_like(amax))


def fp8_effective_weight(w: torch.Tensor) -> torch.Tensor:
    """The fp32 matrix the fp8 engine multiplies by: round-to-nearest-even e4m3 of W/scale, times scale."""
    w = w.detach().float()
    s = fp8_row_scales(w)[:, None]
    return (w / s).to(torch.float8_e4m3fn).float() * s


FP8_LAYER_KEYS = ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w2.weight",
                  "feed_forward.w3.weight")


def fp8_effective_state_dict(sd: dict) -> dict:
    """Sampler state dict with every matrix the fp8 engine stores in fp8 replaced by its dequantised value.
    w1/w3 rows are scaled independently per row, so quantising them separately equals quantising the
    interleaved (w1, w3) matrix the engine streams."""
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith("layers.") and k.endswith(FP8_LAYER_KEYS):
            out[k] = fp8_effective_weight(v)
    return out
