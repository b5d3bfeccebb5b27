"""Feature-extractor plugin: ``target: vaura_amd.feature_extractor.MotionFormer``.

Host-side mirror of the reference's Segment-AVCLIP visual encoder
(/root/reference/models/modules/feature_extractors/avclip/motionformer.py:49-342) for the configuration of
configs/modules/feature_extractors/avclip_vggsound.yaml: same class NAME (it gates the AVCLIP branch of the host,
models/vaura_model.py:73-76), same constructor keywords, same state-dict keys (``cls_token``, ``pos_embed``, ``temp_embed``,
``patch_embed_3d.proj.*``, ``blocks.N.{norm1,norm2,norm3,attn,timeattn,mlp}.*``, ``norm.*``, ``spatial_attn_agg.*``), same call:
``forward(frames (B, S, 3, 16, 224, 224)) -> (feats (B, S, 8, 768), None)`` (motionformer.py:252-303).  The arithmetic runs in
libvaura_hip.so (``AvclipEngine`` -> ``vaura_avclip_forward``); there is no CPU path.

Built: the 'divided_224_16x4' backbone (what the reference constructs when the checkpoint names no other one,
motionformer.py:96-114) with ``factorize_space_time=True``, ``agg_space_module='TransformerEncoderLayer'``,
``agg_time_module='torch.nn.Identity'``, ``add_global_repr=False``.  Not built: trajectory attention
('motionformer_224_16x4'), joint attention, average-pooling aggregators, the global representation, content masks.

Pre-extracted features (B, S, t, 768) are passed through unchanged (benchmarks / synthetic-feature runs).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import synth
from .sampler import _tree


class MotionFormer(nn.Module):
    def __init__(self, extract_features: bool = False, ckpt_path: Optional[str] = None, factorize_space_time: bool = True,
                 agg_space_module: str = "TransformerEncoderLayer", agg_time_module: str = "torch.nn.Identity",
                 add_global_repr: bool = False, agg_segments_module: Optional[str] = None, max_segments: Optional[int] = None):
        super().__init__()
        if add_global_repr or (agg_space_module != "TransformerEncoderLayer") or ("Identity" not in agg_time_module) or \
                not factorize_space_time:
            raise L.VauraHipError("vaura_amd.feature_extractor.MotionFormer builds the avclip_vggsound.yaml configuration only "
                                  "(factorize_space_time, TransformerEncoderLayer spatial aggregation, Identity in time, no global repr)")
        self.extract_features = extract_features
        self.ckpt_path = ckpt_path
        self.cfg = synth.FULL_AVCLIP
        self.embed_dim, self.num_heads = self.cfg.embed_dim, self.cfg.num_heads
        shapes = {k: tuple(v.shape) for k, v in synth.avclip_state_dict(synth.AvclipCfg(depth=self.cfg.depth), seed=0).items()}
        _tree(self, shapes)
        self._loaded = False
        self._engine = None
        self._engine_key = None
        if ckpt_path is not None:
            if not os.path.exists(ckpt_path):
                raise L.VauraHipError(f"MotionFormer: checkpoint {ckpt_path!r} does not exist (the reference would download it, "
                                      "motionformer.py:33-52; there is no network path in this build)")
            self.load_state_dict(self.checkpoint_state_dict(ckpt_path), strict=True)

    def checkpoint_state_dict(self, ckpt_path: str):
        """The extractor's tensors out of a Stage-I AVCLIP checkpoint (``state_dict`` with ``v_encoder.`` prefixes,
        motionformer.py:208-218) or a Motionformer checkpoint (``model_state``, :144-147); every key of this module must be there."""
        blob = torch.load(ckpt_path, map_location="cpu")
        if "state_dict" in blob:
            sd = {k.replace("module.", "").replace("v_encoder.", ""): v for k, v in blob["state_dict"].items()
                  if k.startswith(("module.v_encoder.", "v_encoder."))}
        else:
            sd = blob.get("model_state", blob)
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise L.VauraHipError(f"MotionFormer: {ckpt_path!r} lacks {len(missing)} tensors of the divided_224_16x4 + spatial-aggregation "
                                  f"configuration (e.g. {missing[:3]}): another backbone (trajectory / joint attention) is not built")
        return {k: sd[k] for k in own}

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        res = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._loaded = True
        return res

    def _fingerprint(self, dev):
        return (str(dev),) + tuple((t._version, t.data_ptr()) for t in self.parameters())

    def engine(self):
        from .engine import AvclipEngine
        dev = self.cls_token.device
        key = self._fingerprint(dev)
        if self._engine is None or self._engine_key != key:
            if dev.type != "cuda":
                raise L.VauraHipError("vaura_amd.feature_extractor.MotionFormer encodes frames on a HIP device only; call .to('cuda')")
            self._engine = AvclipEngine(self.cfg, {k: v for k, v in self.state_dict().items()}, dev)
            self._engine_key = key
        return self._engine

    @torch.no_grad()
    def forward(self, x: torch.Tensor, for_loop: bool = False, cont_mask: Optional[torch.Tensor] = None):
        if cont_mask is not None:
            raise NotImplementedError("content masks are a training-time feature (motionformer.py:188-214); generation passes none")
        if x.dim() == 4 and x.shape[-1] == self.embed_dim:          # pre-extracted features (B, S, t, 768)
            return x, None
        if x.dim() != 6:
            raise ValueError(f"MotionFormer expects frames (B, S, C, T, H, W) or features (B, S, t, {self.embed_dim}); got {tuple(x.shape)}")
        if not self.extract_features:
            raise L.VauraHipError("vaura_amd.feature_extractor.MotionFormer encodes frames to features only (extract_features=True, "
                                  "as in configs/modules/feature_extractors/avclip_vggsound.yaml:4); with extract_features=False the "
                                  "reference returns the classification head's output (motionformer.py:160-166, 311-320), which is not built")
        if not self._loaded:
            raise L.VauraHipError("MotionFormer has no weights: pass ckpt_path=... or load_state_dict(...) before encoding frames")
        return self.engine().forward(x), None
