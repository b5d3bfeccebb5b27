"""Deterministic, name-keyed synthetic weights and inputs.

There are no checkpoints on the GPU box (no network), so benchmarks and parity
tests regenerate every tensor from ``(seed, state-dict key)``.  The key names
are the reference's own state-dict names so the very same dictionaries can be
loaded into the reference modules when goldens are generated
(``tests/golden/make_golden.py``):

  sampler keys   /root/reference/models/modules/sampler/llama.py:331-361, 387-412
  codec keys     descript-audio-codec==1.0.0 (``dac.model.dac.DAC``), un-vendored;
                 call sites /root/reference/models/modules/dac/model.py:41-48

Two deviations from the reference's *default* initialisation, both required
for a non-degenerate synthetic run:
  * ``lm_heads`` are N(0, 0.02²) instead of zeros (llama.py:383-385);
  * ``empty_video_emb`` is N(0, 0.02²) instead of ``torch.empty`` (llama.py:336-338).

``round_bf16=True`` makes every streamed GEMV weight exactly representable in
bf16 (round-to-nearest-even, kept as fp32).  The HIP path can then store those
matrices as bf16 (half the HBM bytes) while computing *the same real numbers*
as an fp32 reference run on the same dictionary.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch


# ----------------------------------------------------------------------------- configs
@dataclass(frozen=True)
class SamplerCfg:
    """Mirror of configs/modules/samplers/llama_9cbs.yaml:3-17 (+ llama.py derived sizes)."""

    num_layers: int = 24
    d_model: int = 1536
    nhead: int = 16
    d_codebook: int = 1024
    num_codebooks: int = 9
    block_size_audio: int = 256
    block_size_video: int = 64
    cond_feature_channel_scaler: int = 3
    layer_norm_eps: float = 1e-5
    rope_base: int = 10000
    cond_in: int = 768          # llama.py:332
    uncond_tokens: int = 32     # llama.py:105
    codebook_dim: int = 8       # DAC latent per codebook

    @property
    def head_dim(self) -> int:
        return self.d_model // self.nhead

    @property
    def cond_dim(self) -> int:
        return self.d_model // self.cond_feature_channel_scaler

    @property
    def tok_dim(self) -> int:
        # DAC latent dim; cond_dim + tok_dim == d_model (llama.py:472)
        return self.d_model - self.cond_dim

    @property
    def ffn_dim(self) -> int:
        # llama.py:164-169 (multiple_of=256)
        h = int(2 * (4 * self.d_model) / 3)
        return h if h % 256 == 0 else h + 256 - (h % 256)

    @property
    def block_size(self) -> int:
        return max(self.block_size_audio, self.block_size_video)

    def yaml_params(self) -> dict:
        """kwargs accepted by the reference ``Transformer.__init__`` (llama.py:287-306)."""
        return dict(
            num_layers=self.num_layers, d_model=self.d_model, d_codebook=self.d_codebook,
            nhead=self.nhead, dim_feedforward=4096, dropout=0.1, activation="gelu",
            layer_norm_eps=self.layer_norm_eps, batch_first=True, norm_first=True,
            num_codebooks=self.num_codebooks, block_size_audio=self.block_size_audio,
            block_size_video=self.block_size_video, positional_embedder="learned",
            cond_feature_channel_scaler=self.cond_feature_channel_scaler,
        )


@dataclass(frozen=True)
class CodecCfg:
    """DAC 44.1 kHz / 8 kbps decoder geometry."""

    latent_dim: int = 1024
    decoder_dim: int = 1536
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    n_codebooks: int = 9
    codebook_size: int = 1024
    codebook_dim: int = 8
    sample_rate: int = 44100
    dilations: Tuple[int, ...] = (1, 3, 9)
    encoder_dim: int = 64                           # DAC 44.1 kHz encoder (row f4): 64 -> 128 -> 256 -> 512 -> 1024
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)

    @property
    def hop(self) -> int:
        return int(math.prod(self.decoder_rates))


FULL_SAMPLER = SamplerCfg()
FULL_CODEC = CodecCfg()


def tiny_sampler(num_layers: int = 2, **kw) -> SamplerCfg:
    """Same widths as the real model, fewer layers: seconds on CPU."""
    return SamplerCfg(num_layers=num_layers, **kw)


def tiny_codec() -> CodecCfg:
    return CodecCfg(latent_dim=1024, decoder_dim=96, decoder_rates=(8, 8, 4, 2))


# ----------------------------------------------------------------------------- generators
def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(((zlib.crc32(key.encode("utf-8")) << 16) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFFFFFF)
    return g


def normal(key: str, shape, std: float, seed: int, mean: float = 0.0) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32).normal_(mean, std, generator=_gen(key, seed))


def uniform(key: str, shape, lo: float, hi: float, seed: int) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32).uniform_(lo, hi, generator=_gen(key, seed))


def to_bf16_exact(w: torch.Tensor) -> torch.Tensor:
    return w.to(torch.bfloat16).to(torch.float32)


GEMV_SUFFIXES = ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight",
                 "feed_forward.w2.weight", "feed_forward.w3.weight")


def is_streamed_weight(key: str) -> bool:
    return key.endswith(GEMV_SUFFIXES) or (key.startswith("lm_heads.") and key.endswith(".weight"))


# ----------------------------------------------------------------------------- sampler
def sampler_state_dict(cfg: SamplerCfg = FULL_SAMPLER, seed: int = 0,
                       round_bf16: bool = True) -> Dict[str, torch.Tensor]:
    """State dict with the key set of the reference sampler (SURVEY.md §5 'Checkpoint')."""
    std = 0.02
    D, F = cfg.d_model, cfg.ffn_dim
    sd: Dict[str, torch.Tensor] = {}
    sd["empty_video_emb"] = normal("empty_video_emb", (1, 1, cfg.cond_dim), std, seed)
    sd["cls_embeddings.uncond_embedding"] = normal(
        "cls_embeddings.uncond_embedding", (cfg.uncond_tokens, cfg.cond_in),
        1.0 / math.sqrt(cfg.cond_in), seed)
    sd["cls_embeddings.projection.fc1.weight"] = normal(
        "cls_embeddings.projection.fc1.weight", (cfg.cond_dim, cfg.cond_in), std, seed)
    sd["cls_embeddings.projection.fc2.weight"] = normal(
        "cls_embeddings.projection.fc2.weight", (cfg.cond_dim, cfg.cond_dim), std, seed)
    for i in range(cfg.num_codebooks):
        p = f"tok_embeddings.{i}."
        sd[p + "emb.weight"] = normal(p + "emb.weight", (cfg.d_codebook + 1, cfg.codebook_dim), 1.0, seed)
        sd[p + "out_proj.weight_g"] = uniform(p + "out_proj.weight_g", (cfg.tok_dim, 1, 1), 0.2, 0.5, seed)
        sd[p + "out_proj.weight_v"] = normal(p + "out_proj.weight_v", (cfg.tok_dim, cfg.codebook_dim, 1), 1.0, seed)
        sd[p + "out_proj.bias"] = normal(p + "out_proj.bias", (cfg.tok_dim,), std, seed)
    for l in range(cfg.num_layers):
        p = f"layers.{l}."
        sd[p + "attention.wqkv.weight"] = normal(p + "attention.wqkv.weight", (3 * D, D), std, seed)
        sd[p + "attention.wo.weight"] = normal(p + "attention.wo.weight", (D, D), std, seed)
        sd[p + "feed_forward.w1.weight"] = normal(p + "feed_forward.w1.weight", (F, D), std, seed)
        sd[p + "feed_forward.w3.weight"] = normal(p + "feed_forward.w3.weight", (F, D), std, seed)
        sd[p + "feed_forward.w2.weight"] = normal(p + "feed_forward.w2.weight", (D, F), std, seed)
        # norm gains are 1 in a fresh reference model; jitter them so a kernel that
        # forgets the gain cannot pass.
        sd[p + "attention_norm.weight"] = uniform(p + "attention_norm.weight", (D,), 0.8, 1.2, seed)
        sd[p + "ffn_norm.weight"] = uniform(p + "ffn_norm.weight", (D,), 0.8, 1.2, seed)
    sd["norm.weight"] = uniform("norm.weight", (D,), 0.8, 1.2, seed)
    for i in range(cfg.num_codebooks):
        sd[f"lm_heads.{i}.weight"] = normal(f"lm_heads.{i}.weight", (cfg.d_codebook, D), std, seed)
    if round_bf16:
        for k in list(sd):
            if is_streamed_weight(k):
                sd[k] = to_bf16_exact(sd[k])
    return sd


# ----------------------------------------------------------------------------- codec
def trained_like(sd: Dict[str, torch.Tensor], seed: int = 0, massive: float = 0.0) -> Dict[str, torch.Tensor]:
    """A sampler state dict with the STATISTICS of a trained transformer laid over a synthetic one (tests: there is no network for a
    real V-AURA checkpoint, and N(0, 0.02^2) weights with unit gains exercise none of this): heavy-tailed streamed matrices (one
    element in 2 000 scaled x 25: kurtosis in the hundreds), log-normal RMSNorm gains (sigma 0.5) with six outlier channels x 20, and
    two token-embedding output channels x 100 — "massive activations": a few residual-stream dimensions two orders of magnitude above
    the rest from the first layer on.  massive > 0 additionally scales two norm gains deep in the stack by that factor: activations
    then leave the fp16-plane range (what DecoderEngine.generate_codes_checked must survive).  Deterministic in (sd, seed)."""
    out = dict(sd)
    for k in sorted(sd):
        g = _gen("trained_like/" + k, seed)
        if is_streamed_weight(k):
            w = sd[k].clone()
            n = max(1, w.numel() // 2000)
            idx = torch.randint(0, w.numel(), (n,), generator=g)
            w.view(-1)[idx] *= 25.0
            out[k] = w
        elif k.endswith("attention_norm.weight") or k.endswith("ffn_norm.weight") or k == "norm.weight":
            gain = sd[k] * torch.exp(0.5 * torch.randn(sd[k].shape, generator=g))
            gain[torch.randperm(gain.numel(), generator=g)[:6]] *= 20.0
            out[k] = gain
        elif "tok_embeddings" in k and k.endswith("out_proj.weight_g"):
            wg = sd[k].clone()
            wg.view(-1)[[7, 300]] *= 100.0
            out[k] = wg
    if massive > 0.0:
        for k in ("layers.1.ffn_norm.weight", f"layers.{max(1, len([x for x in sd if x.endswith('ffn_norm.weight')]) * 2 // 3)}.attention_norm.weight"):
            out[k] = out[k] * massive
    return out


def _wn_conv(sd, prefix, cout, cin, k, seed, gain=1.0, transposed=False):
    if transposed:   # ConvTranspose1d weight (Cin, Cout, k); weight_norm dim=0 -> g (Cin,1,1)
        sd[prefix + "weight_v"] = normal(prefix + "weight_v", (cin, cout, k), 1.0, seed)
        sd[prefix + "weight_g"] = uniform(prefix + "weight_g", (cin, 1, 1), 0.85, 1.15, seed) * gain
    else:            # Conv1d weight (Cout, Cin, k); g (Cout,1,1)
        sd[prefix + "weight_v"] = normal(prefix + "weight_v", (cout, cin, k), 1.0, seed)
        sd[prefix + "weight_g"] = uniform(prefix + "weight_g", (cout, 1, 1), 0.85, 1.15, seed) * gain
    sd[prefix + "bias"] = normal(prefix + "bias", (cout,), 0.02, seed)


def codec_state_dict(cfg: CodecCfg = FULL_CODEC, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Decoder half of a ``dac.DAC`` state dict (weight-norm parametrised, as DAC 1.0.0 saves it)."""
    sd: Dict[str, torch.Tensor] = {}
    for i in range(cfg.n_codebooks):
        p = f"quantizer.quantizers.{i}."
        sd[p + "codebook.weight"] = normal(p + "codebook.weight", (cfg.codebook_size, cfg.codebook_dim), 1.0, seed)
        _wn_conv(sd, p + "out_proj.", cfg.latent_dim, cfg.codebook_dim, 1, seed, gain=0.35)
    ch = cfg.decoder_dim
    _wn_conv(sd, "decoder.model.0.", ch, cfg.latent_dim, 7, seed, gain=1.0)
    for b, r in enumerate(cfg.decoder_rates):
        cin, cout = ch // (2 ** b), ch // (2 ** (b + 1))
        p = f"decoder.model.{b + 1}.block."
        sd[p + "0.alpha"] = uniform(p + "0.alpha", (1, cin, 1), 0.5, 1.5, seed)
        _wn_conv(sd, p + "1.", cout, cin, 2 * r, seed, gain=math.sqrt(r / 2.0), transposed=True)
        for u in range(len(cfg.dilations)):
            q = p + f"{u + 2}.block."
            sd[q + "0.alpha"] = uniform(q + "0.alpha", (1, cout, 1), 0.5, 1.5, seed)
            _wn_conv(sd, q + "1.", cout, cout, 7, seed, gain=0.8)
            sd[q + "2.alpha"] = uniform(q + "2.alpha", (1, cout, 1), 0.5, 1.5, seed)
            _wn_conv(sd, q + "3.", cout, cout, 1, seed, gain=0.35)
    cl = ch // (2 ** len(cfg.decoder_rates))
    n = len(cfg.decoder_rates) + 1
    sd[f"decoder.model.{n}.alpha"] = uniform(f"decoder.model.{n}.alpha", (1, cl, 1), 0.5, 1.5, seed)
    _wn_conv(sd, f"decoder.model.{n + 1}.", 1, cl, 7, seed, gain=0.04)
    return sd


def codec_encoder_state_dict(cfg: CodecCfg = FULL_CODEC, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Encoder half of a ``dac.DAC`` state dict: ``encoder.block.*`` + ``quantizer.quantizers.N.in_proj.*`` (the
    codebooks / out_proj of ``codec_state_dict`` with the same seed complete the quantizer)."""
    sd: Dict[str, torch.Tensor] = {}
    for i in range(cfg.n_codebooks):
        _wn_conv(sd, f"quantizer.quantizers.{i}.in_proj.", cfg.codebook_dim, cfg.latent_dim, 1, seed, gain=1.0)
    d = cfg.encoder_dim
    _wn_conv(sd, "encoder.block.0.", d, 1, 7, seed, gain=1.0)
    for b, r in enumerate(cfg.encoder_rates):
        p = f"encoder.block.{b + 1}.block."
        for u in range(len(cfg.dilations)):
            q = p + f"{u}.block."
            sd[q + "0.alpha"] = uniform(q + "0.alpha", (1, d, 1), 0.5, 1.5, seed)
            _wn_conv(sd, q + "1.", d, d, 7, seed, gain=0.8)
            sd[q + "2.alpha"] = uniform(q + "2.alpha", (1, d, 1), 0.5, 1.5, seed)
            _wn_conv(sd, q + "3.", d, d, 1, seed, gain=0.35)
        sd[p + "3.alpha"] = uniform(p + "3.alpha", (1, d, 1), 0.5, 1.5, seed)
        _wn_conv(sd, p + "4.", 2 * d, d, 2 * r, seed, gain=1.0)
        d *= 2
    n = len(cfg.encoder_rates) + 1
    sd[f"encoder.block.{n}.alpha"] = uniform(f"encoder.block.{n}.alpha", (1, d, 1), 0.5, 1.5, seed)
    _wn_conv(sd, f"encoder.block.{n + 1}.", cfg.latent_dim, d, 3, seed, gain=1.0)
    return sd


# ----------------------------------------------------------------------------- Segment-AVCLIP extractor (row f2)
@dataclass(frozen=True)
class AvclipCfg:
    """Motionformer ViT-B/16 'divided_224_16x4' + spatial aggregation layer
    (models/modules/feature_extractors/avclip/motionformer_src/divided_224_16x4.yaml VIT section; motionformer.py:166-184)."""

    depth: int = 12
    embed_dim: int = 768
    num_heads: int = 12
    mlp_ratio: int = 4
    patch: int = 16
    patch_t: int = 2
    frames: int = 16
    img: int = 224
    in_chans: int = 3

    @property
    def t(self) -> int:
        return self.frames // self.patch_t

    @property
    def n(self) -> int:
        return (self.img // self.patch) ** 2


FULL_AVCLIP = AvclipCfg()


def avclip_state_dict(cfg: AvclipCfg = FULL_AVCLIP, seed: int = 0) -> Dict[str, torch.Tensor]:
    """State dict with the reference MotionFormer's key names for the generation-time configuration.  NOT the reference's
    default initialisation: DividedAttention starts at qkv = 0 / proj = 1 and the 3-D patch embedding at 0
    (vit_helper.py:88-93, video_model_builder.py:62) — degenerate; everything here is seeded N(0, 0.02) / jittered gains."""
    D, Hd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    sd: Dict[str, torch.Tensor] = {}
    sd["cls_token"] = normal("avclip.cls_token", (1, 1, D), 0.02, seed)
    sd["pos_embed"] = normal("avclip.pos_embed", (1, cfg.n + 1, D), 0.02, seed)
    sd["temp_embed"] = normal("avclip.temp_embed", (1, cfg.t, D), 0.02, seed)
    sd["patch_embed_3d.proj.weight"] = normal("avclip.pe3d.w", (D, cfg.in_chans, cfg.patch_t, cfg.patch, cfg.patch), 0.02, seed)
    sd["patch_embed_3d.proj.bias"] = normal("avclip.pe3d.b", (D,), 0.02, seed)

    def lin(p, out, inp):
        sd[p + "weight"] = normal("avclip." + p + "weight", (out, inp), 0.02, seed)
        sd[p + "bias"] = normal("avclip." + p + "bias", (out,), 0.02, seed)

    def ln(p):
        sd[p + "weight"] = uniform("avclip." + p + "weight", (D,), 0.8, 1.2, seed)
        sd[p + "bias"] = normal("avclip." + p + "bias", (D,), 0.02, seed)

    for i in range(cfg.depth):
        b = f"blocks.{i}."
        for nm in ("norm1.", "norm2.", "norm3."):
            ln(b + nm)
        for att in ("attn.", "timeattn."):
            lin(b + att + "qkv.", 3 * D, D)
            lin(b + att + "proj.", D, D)
        lin(b + "mlp.fc1.", Hd, D)
        lin(b + "mlp.fc2.", D, Hd)
    ln("norm.")
    a = "spatial_attn_agg."
    sd[a + "cls_token"] = normal("avclip." + a + "cls_token", (1, 1, D), 0.02, seed)
    sd[a + "self_attn.in_proj_weight"] = normal("avclip." + a + "in_proj_weight", (3 * D, D), 0.02, seed)
    sd[a + "self_attn.in_proj_bias"] = normal("avclip." + a + "in_proj_bias", (3 * D,), 0.02, seed)
    lin(a + "self_attn.out_proj.", D, D)
    lin(a + "linear1.", Hd, D)
    lin(a + "linear2.", D, Hd)
    ln(a + "norm1.")
    ln(a + "norm2.")
    return sd


def video_frames(batch: int, segments: int = 4, cfg: AvclipCfg = FULL_AVCLIP, seed: int = 0, first_clip: int = 0) -> torch.Tensor:
    """(B, S, 3, 16, 224, 224) N(0, 1) "normalised RGB" segments, keyed per clip index like ``video_features``."""
    return torch.stack([normal(f"frames.{first_clip + b}", (segments, cfg.in_chans, cfg.frames, cfg.img, cfg.img), 1.0, seed)
                        for b in range(batch)])


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (torch.nn.utils.weight_norm, dim=0)."""
    dims = tuple(range(1, v.dim()))
    return v * (g / v.norm(2, dim=dims, keepdim=True))


# ----------------------------------------------------------------------------- inputs
def video_features(batch: int, tokens: int = 32, dim: int = 768, seed: int = 0,
                   first_clip: int = 0) -> torch.Tensor:
    """(B, Tv, 768) N(0,1) Segment-AVCLIP-shaped features, keyed per *clip index*
    so that the result does not depend on how a batch is sharded over ranks."""
    return torch.stack([normal(f"clip.{first_clip + b}", (tokens, dim), 1.0, seed) for b in range(batch)])


def exp_noise(steps: int, rows: int, vocab: int, seed: int) -> torch.Tensor:
    """Exp(1) noise in the exact draw order of the reference's sampling path:
    one ``empty(rows, vocab).exponential_(1)`` per decode step from one generator
    (utils/utils.py:155-158 -> torch.multinomial fast path, SURVEY.md §7 'Sampling parity')."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.stack([torch.empty(rows, vocab).exponential_(1, generator=g) for _ in range(steps)])
