"""DAC 44.1 kHz decode, fp32 torch-CPU restatement (oracle — see oracle/__init__.py).

PARITY UNPINNED BY THE REFERENCE: the arithmetic is in ``descript-audio-codec==1.0.0``
(/root/reference/conda_env_cuda12.1.yaml:298), which is not vendored; the reference's call
sites are models/modules/dac/model.py:41-48 (``quantizer.from_codes`` then ``model.decode``)
and it ships no vectors for them.  Restated from the published DAC architecture
(dac/model/dac.py, dac/nn/layers.py, dac/nn/quantize.py of that release):

  from_codes   z = sum_k out_proj_k(codebook_k[codes_k])        out_proj = WNConv1d(8 -> latent, k=1)
  Decoder      WNConv1d(latent -> C, k=7, pad=3)
               for r in rates:  Snake(C) -> WNConvTranspose1d(C -> C/2, k=2r, stride=r, pad=ceil(r/2))
                                -> 3 x ResidualUnit(C/2, dilation in (1,3,9));  C /= 2
               Snake(C) -> WNConv1d(C -> 1, k=7, pad=3) -> tanh
  ResidualUnit y = WNConv1d(k=1)(Snake(WNConv1d(k=7, dil=d, pad=3d)(Snake(x))));  out = x + y
  Snake        x + (alpha + 1e-9)^-1 * sin(alpha * x)^2,  alpha per channel
  WN*          weight = g * v / ||v||  (norm over all dims but 0)

Encode side (reference call site models/modules/dac/model.py:30-39: preprocess, model.encode):
  preprocess   right-pad the waveform with zeros to a multiple of hop_length = prod(encoder_rates)
  Encoder      WNConv1d(1 -> d, k=7, pad=3)
               for r in encoder_rates: 3 x ResidualUnit(d, dilation in (1,3,9)) -> Snake(d)
                                -> WNConv1d(d -> 2d, k=2r, stride=r, pad=ceil(r/2));  d *= 2
               Snake(d) -> WNConv1d(d -> latent, k=3, pad=1)
  RVQ.forward  residual = z; for each quantizer: z_e = in_proj(residual) (WNConv1d latent -> 8, k=1);
               e = normalize(z_e), c = normalize(codebook); dist = |e|^2 - 2 e.c^T + |c|^2; code = argmax(-dist);
               z_q = out_proj(z_e + (codebook[code] - z_e)); residual -= z_q            (dac/nn/quantize.py)

Structure cross-check: tests/golden/codec_hf.npz, codec_enc_hf.npz (transformers' independent DacModel).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def fold(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    g, v = sd[prefix + "weight_g"].float(), sd[prefix + "weight_v"].float()
    return v * (g / v.norm(2, dim=tuple(range(1, v.dim())), keepdim=True))


def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def from_codes(sd: Dict[str, torch.Tensor], codes: torch.Tensor) -> torch.Tensor:
    """codes (B, K, T) int64 -> z (B, latent, T)."""
    z = None
    for k in range(codes.shape[1]):
        p = f"quantizer.quantizers.{k}."
        e = F.embedding(codes[:, k], sd[p + "codebook.weight"].float()).transpose(1, 2)
        zk = F.conv1d(e, fold(sd, p + "out_proj."), sd[p + "out_proj.bias"].float())
        z = zk if z is None else z + zk
    return z


def decode_latent(sd: Dict[str, torch.Tensor], z: torch.Tensor, rates: Sequence[int] = (8, 8, 4, 2),
                  dilations: Sequence[int] = (1, 3, 9), act_quant=None, weight_quant=None, quant_input: bool = False,
                  quant_last: bool = True) -> torch.Tensor:
    """z (B, latent, T) -> wav (B, 1, T * prod(rates)).  The hooks are not in the reference; they model the build's reduced
    codec precisions: ``act_quant`` is applied to every Snake output (channels last) before the next convolution reads it
    (``quant_last=False``: not to the last one, whose consumer — the C -> 1 convolution — the kernels keep in fp32), to the
    quantizer's latent too when ``quant_input``; ``weight_quant`` to every folded convolution weight but the last."""
    aq = (lambda x: x) if act_quant is None else (lambda x: act_quant(x.transpose(1, 2)).transpose(1, 2))
    wq = (lambda w: w) if weight_quant is None else weight_quant
    act = lambda x, alpha: aq(snake(x, alpha))
    if quant_input:
        z = aq(z)
    x = F.conv1d(z, wq(fold(sd, "decoder.model.0.")), sd["decoder.model.0.bias"].float(), padding=3)
    for b, r in enumerate(rates):
        p = f"decoder.model.{b + 1}.block."
        x = act(x, sd[p + "0.alpha"].float())
        x = F.conv_transpose1d(x, wq(fold(sd, p + "1.")), sd[p + "1.bias"].float(), stride=r, padding=math.ceil(r / 2))
        for u, d in enumerate(dilations):
            q = p + f"{u + 2}.block."
            y = act(x, sd[q + "0.alpha"].float())
            y = F.conv1d(y, wq(fold(sd, q + "1.")), sd[q + "1.bias"].float(), dilation=d, padding=3 * d)
            y = act(y, sd[q + "2.alpha"].float())
            y = F.conv1d(y, wq(fold(sd, q + "3.")), sd[q + "3.bias"].float())
            x = x + y
    n = len(rates) + 1
    x = snake(x, sd[f"decoder.model.{n}.alpha"].float())
    if quant_last:
        x = aq(x)
    x = F.conv1d(x, fold(sd, f"decoder.model.{n + 1}."), sd[f"decoder.model.{n + 1}.bias"].float(), padding=3)
    return torch.tanh(x)


@torch.no_grad()
def decode(sd: Dict[str, torch.Tensor], codes: torch.Tensor, rates: Sequence[int] = (8, 8, 4, 2), act_quant=None, **kw) -> torch.Tensor:
    return decode_latent(sd, from_codes(sd, codes), rates, act_quant=act_quant, **kw)


def preprocess(wav: torch.Tensor, hop: int) -> torch.Tensor:
    """(B, 1, N) -> zero-padded on the right to a multiple of `hop` (DAC.preprocess)."""
    n = wav.shape[-1]
    return F.pad(wav, (0, math.ceil(n / hop) * hop - n))


def encode_latent(sd: Dict[str, torch.Tensor], wav: torch.Tensor, rates: Sequence[int] = (2, 4, 8, 8),
                  dilations: Sequence[int] = (1, 3, 9)) -> torch.Tensor:
    """wav (B, 1, N), N % prod(rates) == 0 -> z (B, latent, N / prod(rates))."""
    x = F.conv1d(wav, fold(sd, "encoder.block.0."), sd["encoder.block.0.bias"].float(), padding=3)
    for b, r in enumerate(rates):
        p = f"encoder.block.{b + 1}.block."
        for u, d in enumerate(dilations):
            q = p + f"{u}.block."
            y = snake(x, sd[q + "0.alpha"].float())
            y = F.conv1d(y, fold(sd, q + "1."), sd[q + "1.bias"].float(), dilation=d, padding=3 * d)
            y = snake(y, sd[q + "2.alpha"].float())
            y = F.conv1d(y, fold(sd, q + "3."), sd[q + "3.bias"].float())
            x = x + y
        x = snake(x, sd[p + "3.alpha"].float())
        x = F.conv1d(x, fold(sd, p + "4."), sd[p + "4.bias"].float(), stride=r, padding=math.ceil(r / 2))
    n = len(rates) + 1
    x = snake(x, sd[f"encoder.block.{n}.alpha"].float())
    return F.conv1d(x, fold(sd, f"encoder.block.{n + 1}."), sd[f"encoder.block.{n + 1}.bias"].float(), padding=1)


def quantize(sd: Dict[str, torch.Tensor], z: torch.Tensor, n_codebooks: int = 9, return_margin: bool = False):
    """z (B, latent, T) -> codes (B, K, T) int64 [, margin (B, K, T): best - second-best of -dist, for tie analysis]."""
    residual = z
    codes, margins = [], []
    for k in range(n_codebooks):
        p = f"quantizer.quantizers.{k}."
        z_e = F.conv1d(residual, fold(sd, p + "in_proj."), sd[p + "in_proj.bias"].float())
        B, D, T = z_e.shape
        enc = F.normalize(z_e.transpose(1, 2).reshape(B * T, D))
        cb_raw = sd[p + "codebook.weight"].float()
        cb = F.normalize(cb_raw)
        dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
        idx = (-dist).max(1)[1]
        if return_margin:
            top2 = (-dist).topk(2, dim=1).values
            margins.append((top2[:, 0] - top2[:, 1]).reshape(B, T))
        idx = idx.reshape(B, T)
        z_q = F.embedding(idx, cb_raw).transpose(1, 2)
        z_q = z_e + (z_q - z_e)
        z_q = F.conv1d(z_q, fold(sd, p + "out_proj."), sd[p + "out_proj.bias"].float())
        residual = residual - z_q
        codes.append(idx)
    codes = torch.stack(codes, dim=1)
    return (codes, torch.stack(margins, dim=1)) if return_margin else codes


@torch.no_grad()
def encode(sd: Dict[str, torch.Tensor], wav: torch.Tensor, rates: Sequence[int] = (2, 4, 8, 8), n_codebooks: int = 9) -> torch.Tensor:
    """DacModelWrapper.encode (models/modules/dac/model.py:30-39): wav (B, 1, N) -> codes (B, K, ceil(N / hop))."""
    wav = preprocess(wav.float(), int(math.prod(rates)))
    return quantize(sd, encode_latent(sd, wav, rates), n_codebooks)
