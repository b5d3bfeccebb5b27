"""Feature-extractor slot: ``target: vaura_amd.feature_extractor.MotionFormer``.

The Segment-AVCLIP visual encoder (/root/reference/models/modules/feature_extractors/avclip/
motionformer.py:49-342) runs BEFORE the accelerated path and is out of scope.
This class keeps the slot's contract for pre-extracted / synthetic features: the class NAME gates
the AVCLIP branch of the host (models/vaura_model.py:73-76) and ``forward`` returns
``(feats (B, S, t, 768), None)`` (motionformer.py:252-303).  Input that is not already a feature
tensor of that shape is rejected loudly rather than silently mis-conditioning the decoder.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class MotionFormer(nn.Module):
    def __init__(self, **_ignored):
        super().__init__()
        self.register_buffer("_anchor", torch.zeros(1), persistent=False)

    def forward(self, x: torch.Tensor, *a, **k):
        if x.dim() != 4 or x.shape[-1] != 768:
            raise ValueError("vaura_amd.feature_extractor.MotionFormer passes pre-extracted Segment-AVCLIP features "
                             f"(B, S, t, 768) through; got {tuple(x.shape)}. RGB-frame encoding is outside the HIP path.")
        return x, None
