"""CPU restatement of the Segment-AVCLIP visual feature extractor — TEST INFRASTRUCTURE.

Follows, for the configuration every generate_*.yaml uses (configs/modules/feature_extractors/avclip_vggsound.yaml:
extract_features, factorize_space_time, agg_space_module=TransformerEncoderLayer, agg_time_module=Identity, no global
representation) and the `divided_224_16x4.yaml` backbone (what the reference builds when the checkpoint carries no other
backbone name, motionformer.py:96-114):

  MotionFormer.forward / forward_segments / restore_spatio_temp_dims    models/modules/feature_extractors/avclip/motionformer.py:252-364
  VisionTransformer.forward_features (3-D patch embedding, CLS, 'separate' positional embedding, 12 blocks)
                                                                          .../motionformer_src/video_model_builder.py:174-268
  PatchEmbed3D                                                            .../motionformer_src/vit_helper.py:523-557
  DividedSpaceTimeBlock / DividedAttention / qkv_attn / Mlp               .../motionformer_src/vit_helper.py:392-472, 80-172, 34-44, 475-498
  SpatialTransformerEncoderLayer -> BaseEncoderLayer -> nn.TransformerEncoderLayer (norm_first, GELU, eps 1e-6)
                                                                          motionformer.py:366-486
Plain fp32 torch-CPU, eval mode (dropout / DropPath are identities), no content mask (generate never passes one).
PINNED: tests/golden/avclip.npz is the output of the reference's own classes run on the same seeded weights
(tests/golden/make_golden.py avclip).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS = 1e-6          # video_model_builder.py:39 (norm_layer eps), motionformer.py:177 (layer_norm_eps)


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], EPS)


def _lin(x, sd, p):
    return F.linear(x, sd[p + "weight"], sd[p + "bias"])


def qkv_attn(q, k, v):
    """vit_helper.py:34-44 (no mask)."""
    sim = torch.einsum("bid,bjd->bij", q, k)
    return torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), v)


def divided_attention(x, sd, p, heads: int, mode: str, n: int, f: int):
    """DividedAttention.forward (vit_helper.py:98-172).  x (b, 1 + f*n, D); mode 'time': every patch token attends over the
    CLS token and the f tokens at its own spatial location; mode 'space': over CLS and the n tokens of its own frame.  The
    CLS query attends over everything."""
    b, N, D = x.shape
    d = D // heads
    q, k, v = _lin(x, sd, p + "qkv.").chunk(3, dim=-1)
    sh = lambda t: t.reshape(b, N, heads, d).permute(0, 2, 1, 3).reshape(b * heads, N, d)      # 'b n (h d) -> (b h) n d'
    q, k, v = sh(q), sh(k), sh(v)
    q = q * (d ** -0.5)
    cls_q, q_ = q[:, :1], q[:, 1:]
    cls_k, k_ = k[:, :1], k[:, 1:]
    cls_v, v_ = v[:, :1], v[:, 1:]
    cls_out = qkv_attn(cls_q, k, v)
    bh = b * heads
    if mode == "time":      # 'b (f n) d -> (b n) f d'
        re = lambda t: t.reshape(bh, f, n, d).permute(0, 2, 1, 3).reshape(bh * n, f, d)
        back = lambda t: t.reshape(bh, n, f, d).permute(0, 2, 1, 3).reshape(bh, f * n, d)
        r = n
    else:                   # 'b (f n) d -> (b f) n d'
        re = lambda t: t.reshape(bh * f, n, d)
        back = lambda t: t.reshape(bh, f * n, d)
        r = f
    q_, k_, v_ = re(q_), re(k_), re(v_)
    ck = cls_k.repeat_interleave(r, dim=0)       # 'b () d -> (b r) () d'
    cv = cls_v.repeat_interleave(r, dim=0)
    out = qkv_attn(q_, torch.cat((ck, k_), dim=1), torch.cat((cv, v_), dim=1))
    out = torch.cat((cls_out, back(out)), dim=1)
    out = out.reshape(b, heads, N, d).permute(0, 2, 1, 3).reshape(b, N, D)                      # '(b h) n d -> b n (h d)'
    return _lin(out, sd, p + "proj.")


def block(x, sd, p, heads, n, f):
    """DividedSpaceTimeBlock.forward (vit_helper.py:443-472): time attention, space attention, MLP — each with a residual."""
    tr = x + divided_attention(_ln(x, sd, p + "norm3."), sd, p + "timeattn.", heads, "time", n, f)
    sr = tr + divided_attention(_ln(tr, sd, p + "norm1."), sd, p + "attn.", heads, "space", n, f)
    h = F.gelu(_lin(_ln(sr, sd, p + "norm2."), sd, p + "mlp.fc1."))
    return sr + _lin(h, sd, p + "mlp.fc2.")


def tokens(frames, sd, f: int):
    """Tokenisation + positional embedding (video_model_builder.py:174-255, POS_EMBED 'separate', crop 224)."""
    w = sd["patch_embed_3d.proj.weight"]
    x = F.conv3d(frames, w, sd["patch_embed_3d.proj.bias"], stride=tuple(w.shape[2:]))         # (bs, D, f, h, w)
    n = x.shape[3] * x.shape[4]
    x = x.flatten(2).transpose(1, 2)                                                           # (bs, f*n, D), order (f, h, w)
    x = torch.cat((sd["cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
    pos = sd["pos_embed"]
    total = torch.cat([pos[:, :1], pos[:, 1:].repeat(1, f, 1) + sd["temp_embed"].repeat_interleave(n, 1)], dim=1)
    return x + total, n


def spatial_aggregate(y, sd, heads: int, p: str = "spatial_attn_agg."):
    """SpatialTransformerEncoderLayer (motionformer.py:488-512, 366-448): per frame, a CLS token is prepended to the n patch
    tokens and one pre-norm nn.TransformerEncoderLayer is applied; the CLS row is the frame's feature.  y (bs*t, n, D)."""
    b, n, D = y.shape
    d = D // heads
    x = torch.cat((sd[p + "cls_token"].expand(b, -1, -1), y), dim=1)
    z = _ln(x, sd, p + "norm1.")
    qkv = F.linear(z, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    sh = lambda t: t.reshape(b, n + 1, heads, d).transpose(1, 2)
    a = torch.softmax((sh(q) * (d ** -0.5)) @ sh(k).transpose(-1, -2), dim=-1) @ sh(v)        # (b, h, n+1, d)
    a = a.transpose(1, 2).reshape(b, n + 1, D)
    x = x + _lin(a, sd, p + "self_attn.out_proj.")
    x = x + _lin(F.gelu(_lin(_ln(x, sd, p + "norm2."), sd, p + "linear1.")), sd, p + "linear2.")
    return x[:, 0]


def forward(sd: Dict[str, torch.Tensor], frames: torch.Tensor, heads: int = 12, depth: Optional[int] = None,
            trace: Optional[dict] = None) -> torch.Tensor:
    """frames (B, S, 3, T=16, 224, 224) fp32 -> features (B, S, t=8, 768): MotionFormer.forward with for_loop=False
    (motionformer.py:252-303) under the configuration named in the module docstring."""
    B, S, C, T, H, W = frames.shape
    f = T // sd["patch_embed_3d.proj.weight"].shape[2]
    x, n = tokens(frames.reshape(B * S, C, T, H, W).float(), sd, f)
    if trace is not None:
        trace["tokens"] = x.clone()
    depth = depth if depth is not None else 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        x = block(x, sd, f"blocks.{i}.", heads, n, f)
        if trace is not None and i in (0, depth - 1):
            trace[f"block{i}"] = x.clone()
    x = _ln(x[:, 1:], sd, "norm.")                                   # motionformer.py:311-314 (CLS dropped before the norm)
    D = x.shape[-1]
    y = x.reshape(B * S * f, n, D)                                   # restore_spatio_temp_dims + 'BS D t h w -> (BS t) (h w) D'
    out = spatial_aggregate(y, sd, heads)
    return out.reshape(B, S, f, D)
