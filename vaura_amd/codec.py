"""Drop-in codec plugin: ``target: vaura_amd.codec.DacModelWrapper``.

Mirror of /root/reference/models/modules/dac/model.py:12-60, both directions.  The class
name must stay ``DacModelWrapper`` (models/vaura_model.py:87, scripts/generate.py:215).  ``.model``
is a parameter holder with the state-dict keys of ``dac.DAC`` 1.0.0 (weight-norm parametrised:
``weight_g`` / ``weight_v`` / ``bias`` / ``alpha`` / ``codebook.weight``), exposing what
``Transformer.initialize_embeddings`` reads (``quantizer.quantizers[i].codebook.weight`` and
``.out_proj``).  ``decode`` (codes -> waveform, the hot path's last stage) and ``encode`` (waveform -> codes: raw-audio
prompts, "compressed original" outputs) are executed by libvaura_hip.so (``CodecEngine`` /
``CodecEncoderEngine``); there is no CPU path.
"""
from __future__ import annotations

import os
import typing as tp
import warnings

import torch
import torch.nn as nn

from . import _lib as L
from . import synth
from .engine import CodecEncoderEngine, CodecEngine
from .sampler import _tree

MODEL_SR = [16000, 24000, 44000, 44100]


class _WNConvHolder(nn.Module):
    """Parameter holder of one weight-normed conv (``weight_g`` / ``weight_v`` / ``bias``) that also answers what the
    REFERENCE sampler's ``initialize_embeddings`` reads from a codec's projections (llama.py:403-408): the folded
    ``weight`` and ``in_channels`` / ``out_channels``."""

    @property
    def weight(self) -> torch.Tensor:
        return synth.fold_weight_norm(self.weight_g.detach().float(), self.weight_v.detach().float())

    @property
    def out_channels(self) -> int:
        return int(self.weight_v.shape[0])

    @property
    def in_channels(self) -> int:
        return int(self.weight_v.shape[1])


class _Holder(nn.Module):
    """``dac.DAC``-shaped parameter tree (encoder, quantizer, decoder)."""

    def __init__(self, cfg: synth.CodecCfg, sd: tp.Dict[str, torch.Tensor]):
        super().__init__()
        self.cfg = cfg
        self.sample_rate = cfg.sample_rate
        _tree(self, {k: tuple(v.shape) for k, v in sd.items()})
        for m in self.modules():
            if type(m) is nn.Module and "weight_g" in m._parameters and "weight_v" in m._parameters:
                m.__class__ = _WNConvHolder
        self.load_state_dict(sd, strict=True)


class DacModelWrapper(nn.Module):
    def __init__(self, model_sr: int = 24000, ckpt_path: tp.Optional[str] = None, synthetic: bool = False,
                 synthetic_seed: int = 0, precision: str = "f16pair", weights_from_state_dict: bool = False) -> None:
        """``model_sr`` / ``ckpt_path`` as in the reference (dac/model.py:12-25).  Extras understood by this plugin only:
        ``precision`` of the decode convolutions — "f16pair" (default), "f32" (exact fp32 MFMA), "f16" (plain fp16 operands with
        fp32 accumulate: what the reference's own `.half()` codec computes in, one matrix instruction per product), "f16pair_w8" (fp8 conv
        weights) or "mx8" (fp8 weights and block-scaled fp8 activations on the fp8 MFMA; both BASELINE configs[4]); ``synthetic=True`` (or env VAURA_SYNTHETIC_CODEC=1) asks for seeded synthetic weights
        (``synthetic_seed``) — benchmarks and tests on machines without checkpoints.  Without it a checkpoint is REQUIRED:
        the reference downloads one when ``ckpt_path`` is absent (dac/model.py:20-23) and never runs on random weights; this
        build has no network path, so it raises instead of silently producing noise."""
        super().__init__()
        self.precision = precision
        assert model_sr in MODEL_SR, "Invalid model samplerate"
        if model_sr not in (44000, 44100):
            raise L.VauraHipError("only the 44.1 kHz DAC geometry is built (configs/modules/audio_codecs/dac_8kbps_wrapper.yaml)")
        self.model_sr = model_sr
        self.cfg = synth.FULL_CODEC
        sd0 = dict(synth.codec_state_dict(self.cfg, seed=synthetic_seed))
        sd0.update(synth.codec_encoder_state_dict(self.cfg, seed=synthetic_seed))
        self.model = _Holder(self.cfg, sd0)
        synthetic = synthetic or os.environ.get("VAURA_SYNTHETIC_CODEC") == "1"
        if ckpt_path is not None:
            if not os.path.exists(ckpt_path):
                raise L.VauraHipError(f"DacModelWrapper: checkpoint {ckpt_path!r} does not exist (the reference would download "
                                      "one here, dac/model.py:23; there is no network path in this build)")
            self.model.load_state_dict(self.checkpoint_state_dict(ckpt_path, self.model.state_dict()), strict=True)
        elif weights_from_state_dict:
            # VAURAModel.load_from_checkpoint: the Lightning checkpoint carries audio_encoder.model.*; the strict load that
            # follows fills every tensor (or fails).  Until then the holder is poisoned so that nothing decodes by accident.
            for t in self.model.parameters():
                t.data.fill_(float("nan"))
        elif synthetic:
            warnings.warn("DacModelWrapper: seeded SYNTHETIC DAC weights (synthetic=True): output is not real audio")
        else:
            raise L.VauraHipError("DacModelWrapper: no checkpoint.  Pass ckpt_path=<DAC 44.1 kHz weights> (the reference downloads "
                                  "them, dac/model.py:20-23; this build cannot), or synthetic=True for seeded synthetic weights")
        self._engine: tp.Optional[CodecEngine] = None
        self._engine_key = None
        self._enc_engine: tp.Optional[CodecEncoderEngine] = None
        self._enc_engine_key = None

    @staticmethod
    def checkpoint_state_dict(ckpt_path: str, own: tp.Dict[str, torch.Tensor]) -> tp.Dict[str, torch.Tensor]:
        """Tensors of a ``dac.DAC.save`` file for exactly the keys of ``own``.  Accepts both weight-norm spellings
        (``weight_g`` / ``weight_v`` and torch >= 2.1's ``parametrizations.weight.original0`` / ``original1``); every key of
        ``own`` must be present with the right shape — a partial load would decode plausible-looking noise."""
        blob = torch.load(ckpt_path, map_location="cpu")
        sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
        ren = {}
        for k, v in sd.items():
            k = k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v")
            ren[k[len("model."):] if k.startswith("model.") and k[len("model."):] in own else k] = v
        missing = [k for k in own if k not in ren]
        bad = [k for k in own if k in ren and tuple(ren[k].shape) != tuple(own[k].shape)]
        if missing or bad:
            raise L.VauraHipError(f"DacModelWrapper: {ckpt_path!r} is not a DAC 44.1 kHz checkpoint this build can use: "
                                  f"{len(missing)} keys missing (e.g. {missing[:3]}), {len(bad)} shape mismatches (e.g. {bad[:3]})")
        return {k: ren[k] for k in own}

    def _weights_fingerprint(self, dev):
        return (str(dev), self.precision) + tuple((t._version, t.data_ptr()) for t in self.model.parameters())

    def engine(self) -> CodecEngine:
        dev = next(self.model.parameters()).device
        key = self._weights_fingerprint(dev)
        if self._engine is None or self._engine_key != key:
            if dev.type != "cuda":
                raise L.VauraHipError("vaura_amd.codec.DacModelWrapper decodes on a HIP device only; call .to('cuda')")
            self._engine = CodecEngine(self.cfg, {k: v.float() for k, v in self.model.state_dict().items()}, dev,
                                       precision=self.precision)
            self._engine_key = key
        return self._engine

    def forward(self, wav: torch.Tensor):
        return self.encode(wav)

    def encoder_engine(self) -> CodecEncoderEngine:
        dev = next(self.model.parameters()).device
        key = self._weights_fingerprint(dev)
        if self._enc_engine is None or self._enc_engine_key != key:
            if dev.type != "cuda":
                raise L.VauraHipError("vaura_amd.codec.DacModelWrapper encodes on a HIP device only; call .to('cuda')")
            self._enc_engine = CodecEncoderEngine(self.cfg, {k: v.float() for k, v in self.model.state_dict().items()}, dev)
            self._enc_engine_key = key
        return self._enc_engine

    @torch.no_grad()
    def encode(self, wav: torch.Tensor):
        """wav (N) | (C=1, N) | (B, 1, N) -> codes (B, 9, ceil(N / 512)) — dac/model.py:30-39 (unsqueeze to 3 dims,
        DAC.preprocess zero-pads to a multiple of the hop, DAC.encode)."""
        if wav.ndim < 2:
            wav = wav.unsqueeze(0)
        if wav.ndim < 3:
            wav = wav.unsqueeze(0)
        return self.encoder_engine().encode(wav)

    @torch.no_grad()
    def decode(self, codes: tp.Union[torch.Tensor, tp.List[tp.Tuple[torch.Tensor]]]):
        """codes (B, 9, T) or [(codes, None)] -> wav (B, 1, 512*T) — dac/model.py:41-48."""
        if type(codes) == list:  # EnCodec-style frame list
            codes = codes[0][0]
        return self.engine().decode(codes)

    @property
    def sample_rate(self):
        return self.model.sample_rate

    @property
    def channels(self):
        return 1

    @property
    def frame_rate(self):
        return None
