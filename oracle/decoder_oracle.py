"""fp32 CPU restatement of the reference's Llama-style 9-codebook decoder (oracle — see oracle/__init__.py).

Follows /root/reference/models/modules/sampler/llama.py:
  token embedding      :60-73, 455-460   emb_k[idx] -> weight-normed 1x1 conv 8->1024 (+bias), summed over k
  video MLP            :79-92, 136-141   fc2(gelu_tanh(fc1(x))), no bias
  repeat/pad video     :555-586          position p takes video[p // 7] (or empty_video_emb past Tv)
  channel concat       :472              h0 = [cond(512) | tok(1024)]
  RMSNorm              :147-158
  RoPE                 :593-603, 633-650 interleaved pairs, table = polar(1, outer(t, base^(-2i/hd)))
  attention            :219-260          wqkv -> split -> rope(q,k) -> causal SDPA -> wo
  SwiGLU               :161-177
  block                :272-283          pre-norm residual
  heads                :503-504          stack_k Linear_k(norm(h))

Two evaluation orders of the *same* function are provided:
  ``forward_full``  — what the reference does every step: the whole prefix, no cache
                      (models/vaura_model.py:504-506).  Used to pin the oracle to the goldens and
                      as the reference-faithful CPU baseline.
  ``CachedDecoder`` — one position per call with a K/V cache (what the HIP path does).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def rope_table(n_pos: int, head_dim: int, base: int = 10000) -> torch.Tensor:
    """(n_pos, head_dim/2, 2) cos/sin table, llama.py:593-603 (torch.polar on fp32 angles)."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(n_pos), inv)
    z = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([z.real, z.imag], dim=-1)


def apply_rope(x: torch.Tensor, tab: torch.Tensor) -> torch.Tensor:
    """x (B, L, H, hd), tab (L, hd/2, 2); rotates pairs (x[2i], x[2i+1]) — llama.py:633-650."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    c = tab[None, :, None, :, 0]
    s = tab[None, :, None, :, 1]
    out = torch.stack([xs[..., 0] * c - xs[..., 1] * s, xs[..., 1] * c + xs[..., 0] * s], dim=-1)
    return out.flatten(3)


def rmsnorm(x: torch.Tensor, gain: torch.Tensor, eps: float) -> torch.Tensor:
    """llama.py:153-158 — (x * rsqrt(mean(x^2) + eps)) * gain, in fp32."""
    return (x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)) * gain


class DecoderOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], num_layers: int, nhead: int, eps: float = 1e-5,
                 rope_base: int = 10000, block_size: int = 256, tokens_per_video_frame: int = 7):
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.L = num_layers
        self.H = nhead
        self.eps = eps
        self.D = self.sd["norm.weight"].numel()
        self.hd = self.D // nhead
        self.K = sum(1 for k in self.sd if k.startswith("lm_heads.") and k.endswith(".weight"))
        self.tpf = tokens_per_video_frame
        self.rope = rope_table(block_size, self.hd, rope_base)
        # weight-norm fold of the token projection (llama.py:70-73, 405-409): w = g * v / ||v||
        self.tok_w, self.tok_b = [], []
        for k in range(self.K):
            g = self.sd[f"tok_embeddings.{k}.out_proj.weight_g"]
            v = self.sd[f"tok_embeddings.{k}.out_proj.weight_v"]
            w = v * (g / v.norm(2, dim=(1, 2), keepdim=True))
            self.tok_w.append(w[:, :, 0])
            self.tok_b.append(self.sd[f"tok_embeddings.{k}.out_proj.bias"])
        self.head_w = torch.stack([self.sd[f"lm_heads.{k}.weight"] for k in range(self.K)])  # (K, V, D)

    # ---------------------------------------------------------------- input side
    def token_embedding(self, idx: torch.Tensor) -> torch.Tensor:
        """idx (Bs, K, L) int64 -> (Bs, L, tok_dim); llama.py:455-460."""
        out = None
        for k in range(self.K):
            e = F.embedding(idx[:, k], self.sd[f"tok_embeddings.{k}.emb.weight"])  # (Bs, L, 8)
            z = F.linear(e, self.tok_w[k], self.tok_b[k])
            out = z if out is None else out + z
        return out

    def cond_projection(self, cond: torch.Tensor) -> torch.Tensor:
        """(Bs, Tv, 768) -> (Bs, Tv, 512); llama.py:88-92 (GELU tanh approximation, bias-free)."""
        h = F.linear(cond, self.sd["cls_embeddings.projection.fc1.weight"])
        h = F.gelu(h, approximate="tanh")
        return F.linear(h, self.sd["cls_embeddings.projection.fc2.weight"])

    def cond_for_positions(self, cond_proj: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        """Closed form of ``_repeat_and_pad_video`` (llama.py:555-586): position p -> frame p // tpf,
        frames >= Tv read ``empty_video_emb``."""
        Bs, Tv, C = cond_proj.shape
        frame = positions // self.tpf
        ext = torch.cat([cond_proj, self.sd["empty_video_emb"].expand(Bs, 1, C)], dim=1)
        return ext[:, torch.clamp(frame, max=Tv)]

    def null_condition(self, cond: torch.Tensor) -> torch.Tensor:
        """CFG null branch, vaura_model.py:790-793."""
        return torch.zeros_like(cond) + self.sd["cls_embeddings.uncond_embedding"]

    # ---------------------------------------------------------------- one layer, op by op (pinned by tests/golden/ops.npz)
    def attention(self, x: torch.Tensor, l: int, tab: torch.Tensor) -> torch.Tensor:
        """Attention.forward (llama.py:219-260) of layer l on an already-normed x (Bs, L, D): wqkv -> rope(q, k) -> causal SDPA -> wo."""
        Bs, L, _ = x.shape
        p = f"layers.{l}."
        qkv = F.linear(x, self.sd[p + "attention.wqkv.weight"])
        q, k, v = qkv.split([self.D, self.D, self.D], dim=-1)
        q = apply_rope(q.view(Bs, L, self.H, self.hd), tab).transpose(1, 2)
        k = apply_rope(k.view(Bs, L, self.H, self.hd), tab).transpose(1, 2)
        v = v.view(Bs, L, self.H, self.hd).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v, is_causal=True)  # llama.py:246-255
        a = a.transpose(1, 2).contiguous().view(Bs, L, self.D)
        return F.linear(a, self.sd[p + "attention.wo.weight"])

    def feed_forward(self, x: torch.Tensor, l: int) -> torch.Tensor:
        """FeedForward.forward (llama.py:176-177) of layer l on an already-normed x: w2(silu(w1 x) * w3 x)."""
        p = f"layers.{l}."
        g = F.silu(F.linear(x, self.sd[p + "feed_forward.w1.weight"])) * F.linear(x, self.sd[p + "feed_forward.w3.weight"])
        return F.linear(g, self.sd[p + "feed_forward.w2.weight"])

    def block(self, h: torch.Tensor, l: int, tab: torch.Tensor) -> torch.Tensor:
        """TransformerBlock.forward (llama.py:272-283): pre-norm residual (DropPath / Dropout are the identity in eval mode)."""
        p = f"layers.{l}."
        h = h + self.attention(rmsnorm(h, self.sd[p + "attention_norm.weight"], self.eps), l, tab)
        return h + self.feed_forward(rmsnorm(h, self.sd[p + "ffn_norm.weight"], self.eps), l)

    # ---------------------------------------------------------------- full recompute
    def forward_full(self, idx: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """idx (Bs,K,L) int64, cond (Bs,Tv,768) -> logits (Bs,K,L,V); llama.py:445-504."""
        Bs, K, L = idx.shape
        tok = self.token_embedding(idx)
        cp = self.cond_for_positions(self.cond_projection(cond), torch.arange(L))
        h = torch.cat([cp, tok], dim=-1)
        tab = self.rope[:L]
        for l in range(self.L):
            h = self.block(h, l, tab)
        h = rmsnorm(h, self.sd["norm.weight"], self.eps)
        return torch.einsum("bld,kvd->bklv", h, self.head_w)


class CachedDecoder:
    """One position per call; K (rotated) and V kept per layer.  Mathematically the last row of
    ``forward_full`` under causal masking."""

    def __init__(self, dec: DecoderOracle, cond: torch.Tensor, max_len: int):
        self.d = dec
        self.Bs = cond.shape[0]
        self.cond_proj = dec.cond_projection(cond)  # hoisted: step-invariant
        self.k = [torch.zeros(self.Bs, dec.H, max_len, dec.hd) for _ in range(dec.L)]
        self.v = [torch.zeros(self.Bs, dec.H, max_len, dec.hd) for _ in range(dec.L)]
        self.pos = 0
        self.scale = 1.0 / math.sqrt(dec.hd)

    def step(self, tokens: torch.Tensor, want_logits: bool = True) -> Optional[torch.Tensor]:
        """tokens (Bs, K) int64 at position ``self.pos`` -> logits (Bs, K, V) for the next position."""
        d = self.d
        p = self.pos
        tok = d.token_embedding(tokens[:, :, None])
        cp = d.cond_for_positions(self.cond_proj, torch.tensor([p]))
        h = torch.cat([cp, tok], dim=-1)  # (Bs,1,D)
        tab = d.rope[p:p + 1]
        for l in range(d.L):
            q_ = f"layers.{l}."
            x = rmsnorm(h, d.sd[q_ + "attention_norm.weight"], d.eps)
            qkv = F.linear(x, d.sd[q_ + "attention.wqkv.weight"])
            q, k, v = qkv.split([d.D, d.D, d.D], dim=-1)
            q = apply_rope(q.view(self.Bs, 1, d.H, d.hd), tab).transpose(1, 2)   # (Bs,H,1,hd)
            k = apply_rope(k.view(self.Bs, 1, d.H, d.hd), tab).transpose(1, 2)
            v = v.view(self.Bs, 1, d.H, d.hd).transpose(1, 2)
            self.k[l][:, :, p] = k[:, :, 0]
            self.v[l][:, :, p] = v[:, :, 0]
            s = torch.matmul(q, self.k[l][:, :, :p + 1].transpose(-1, -2)) * self.scale
            a = torch.matmul(torch.softmax(s, dim=-1), self.v[l][:, :, :p + 1])
            a = a.transpose(1, 2).reshape(self.Bs, 1, d.D)
            h = h + F.linear(a, d.sd[q_ + "attention.wo.weight"])
            x = rmsnorm(h, d.sd[q_ + "ffn_norm.weight"], d.eps)
            g = F.silu(F.linear(x, d.sd[q_ + "feed_forward.w1.weight"])) * F.linear(
                x, d.sd[q_ + "feed_forward.w3.weight"])
            h = h + F.linear(g, d.sd[q_ + "feed_forward.w2.weight"])
        self.pos += 1
        if not want_logits:
            return None
        h = rmsnorm(h, d.sd["norm.weight"], d.eps)
        return torch.einsum("bd,kvd->bkv", h[:, 0], d.head_w)
