"""CPU restatement of the step right after the hot path: post-codec audio scaling
(SURVEY.md §8 row f3).  TEST INFRASTRUCTURE ONLY — nothing under vaura_amd/ imports this.

Follows, line for line in meaning:
  normalize_audio   /root/reference/utils/data_utils.py:407-466   ('peak' | 'clip' | 'rms' | 'none')
  _clip_wav         /root/reference/utils/data_utils.py:389-404
  scale_audio       /root/reference/scripts/generate.py:440-461   (dtype gate, normalize, reshape(1, -1).cpu())
The 'loudness' strategy calls torchaudio.transforms.Loudness (third-party, not in this image): not restated.
Pinned by tests/golden/post.npz, produced by the reference's own normalize_audio (make_golden.py post).
"""
from __future__ import annotations

import torch


def normalize_audio(wav: torch.Tensor, normalize: bool = True, strategy: str = "peak", peak_clip_headroom_db: float = 6,
                    rms_headroom_db: float = 18) -> torch.Tensor:
    scale_peak = 10 ** (-peak_clip_headroom_db / 20)          # data_utils.py:439
    scale_rms = 10 ** (-rms_headroom_db / 20)                 # :440
    if strategy == "peak":                                    # :441-444
        # NB `float / tensor` is Tensor.__rtruediv__ = tensor.reciprocal() * float32(float): the HIP kernel does the same
        rescaling = scale_peak / wav.abs().max()
        if normalize or rescaling < 1:
            wav = wav * rescaling
    elif strategy == "clip":                                  # :445-446
        wav = wav.clamp(-scale_peak, scale_peak)
    elif strategy == "rms":                                   # :447-452
        mono = wav.mean(dim=0)
        rescaling = scale_rms / mono.pow(2).mean().sqrt()
        if normalize or rescaling < 1:
            wav = wav * rescaling
        wav = wav.clamp(-1, 1)                                # _clip_wav :389-404 (in place there)
    else:                                                     # :459-463
        assert wav.abs().max() < 1
        assert strategy == "" or strategy == "none", f"Unexpected strategy: '{strategy}'"
    return wav


def scale_audio(audio: torch.Tensor, strategy: str = "clip", sample_rate: int = 44100, db: float = 6.0) -> torch.Tensor:
    if audio.dtype not in [torch.float32, torch.int32, torch.int16, torch.uint8]:      # generate.py:446-453
        audio = audio.to(torch.float32)
    audio = normalize_audio(audio, strategy=strategy, peak_clip_headroom_db=db)         # :455-458
    return audio.reshape(1, -1).to("cpu")                                               # :459
