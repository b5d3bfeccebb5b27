"""Drop-in codec plugin: ``target: vaura_amd.codec.DacModelWrapper``.

Mirror of /root/reference/models/modules/dac/model.py:12-60 for the decode direction.  The class
name must stay ``DacModelWrapper`` (models/vaura_model.py:87, scripts/generate.py:215).  ``.model``
is a parameter holder with the state-dict keys of ``dac.DAC`` 1.0.0 (weight-norm parametrised:
``weight_g`` / ``weight_v`` / ``bias`` / ``alpha`` / ``codebook.weight``), exposing what
``Transformer.initialize_embeddings`` reads (``quantizer.quantizers[i].codebook.weight`` and
``.out_proj``).  Decoding is executed by libvaura_hip.so (``CodecEngine``) in fp32.

``encode`` (wav -> codes) is outside the hot path and raises.
"""
from __future__ import annotations

import os
import typing as tp
import warnings

import torch
import torch.nn as nn

from . import _lib as L
from . import synth
from .engine import CodecEngine
from .sampler import _tree

MODEL_SR = [16000, 24000, 44000, 44100]


class _Holder(nn.Module):
    """``dac.DAC``-shaped parameter tree (decoder half + quantizer codebooks / out_proj)."""

    def __init__(self, cfg: synth.CodecCfg, sd: tp.Dict[str, torch.Tensor]):
        super().__init__()
        self.cfg = cfg
        self.sample_rate = cfg.sample_rate
        _tree(self, {k: tuple(v.shape) for k, v in sd.items()})
        self.load_state_dict(sd, strict=True)


class DacModelWrapper(nn.Module):
    def __init__(self, model_sr: int = 24000, ckpt_path: tp.Optional[str] = None, synthetic_seed: int = 0) -> None:
        super().__init__()
        assert model_sr in MODEL_SR, "Invalid model samplerate"
        if model_sr not in (44000, 44100):
            raise L.VauraHipError("only the 44.1 kHz DAC geometry is built (configs/modules/audio_codecs/dac_8kbps_wrapper.yaml)")
        self.model_sr = model_sr
        self.cfg = synth.FULL_CODEC
        self.model = _Holder(self.cfg, synth.codec_state_dict(self.cfg, seed=synthetic_seed))
        if ckpt_path is not None and os.path.exists(ckpt_path):
            blob = torch.load(ckpt_path, map_location="cpu")
            sd = blob.get("state_dict", blob)
            own = self.model.state_dict()
            self.model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        else:
            # the reference downloads weights here (dac/model.py:23); there is no network on the
            # target machines, so fall back to seeded synthetic weights and say so.
            warnings.warn("DacModelWrapper: no checkpoint given; using seeded synthetic DAC weights")
        self._engine: tp.Optional[CodecEngine] = None
        self._engine_dev = None

    def engine(self) -> CodecEngine:
        dev = next(self.model.parameters()).device
        if self._engine is None or self._engine_dev != dev:
            if dev.type != "cuda":
                raise L.VauraHipError("vaura_amd.codec.DacModelWrapper decodes on a HIP device only; call .to('cuda')")
            self._engine = CodecEngine(self.cfg, {k: v.float() for k, v in self.model.state_dict().items()}, dev)
            self._engine_dev = dev
        return self._engine

    def forward(self, wav: torch.Tensor):
        return self.encode(wav)

    def encode(self, wav: torch.Tensor):
        raise NotImplementedError("DAC encode is outside the accelerated hot path (SURVEY.md §8 f4)")

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
