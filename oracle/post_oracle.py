"""CPU restatement of the step right after the hot path: post-codec audio scaling
(SURVEY.md §8 row f3).  TEST INFRASTRUCTURE ONLY — nothing under vaura_amd/ imports this.

Follows, line for line in meaning:
  normalize_audio   /root/reference/utils/data_utils.py:407-466   ('peak' | 'clip' | 'rms' | 'none')
  _clip_wav         /root/reference/utils/data_utils.py:389-404
  scale_audio       /root/reference/scripts/generate.py:440-461   (dtype gate, normalize, reshape(1, -1).cpu())
Pinned by tests/golden/post.npz, produced by the reference's own normalize_audio (make_golden.py post) — EXCEPT the 'loudness'
strategy: normalize_loudness (data_utils.py:347-387) calls torchaudio.transforms.Loudness (torchaudio==2.2.1,
conda_env_cuda12.1.yaml:256), third-party and absent from /root/reference and from this image.  `loudness_lkfs` below restates the
published algorithm (ITU-R BS.1770-4 as torchaudio.functional.loudness implements it) in float64: PARITY UNPINNED.
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


def _biquad(x, b, a):
    """lfilter of one biquad in direct form I, then the clamp torchaudio's lfilter applies to what it returns (clamp=True)."""
    import numpy as np
    from scipy.signal import lfilter
    return np.clip(lfilter(np.asarray(b) / a[0], np.asarray(a) / a[0], x), -1.0, 1.0)


def loudness_lkfs(wav: torch.Tensor, sample_rate: int) -> float:
    """Integrated loudness of a mono (1, N) clip: K-weighting (treble shelf +4 dB at 1500 Hz, Q = 1/sqrt 2; high-pass 38 Hz, Q = 0.5),
    mean square over 400 ms blocks with 75 % overlap, absolute gate -70 LKFS, relative gate -10 LU, -0.691 + 10 log10(mean)."""
    import math
    import numpy as np
    x = wav.detach().double().reshape(-1).numpy()
    gate = int(round(0.4 * sample_rate))
    step = int(round(gate * 0.25))
    w0 = 2 * math.pi * 1500.0 / sample_rate
    A = math.exp(4.0 / 40 * math.log(10))
    alpha = math.sin(w0) / 2 / (1 / math.sqrt(2))
    t1, t2, t3 = 2 * math.sqrt(A) * alpha, (A - 1) * math.cos(w0), (A + 1) * math.cos(w0)
    y = _biquad(x, [A * ((A + 1) + t2 + t1), -2 * A * ((A - 1) + t3), A * ((A + 1) + t2 - t1)],
                [(A + 1) - t2 + t1, 2 * ((A - 1) - t3), (A + 1) - t2 - t1])
    w0 = 2 * math.pi * 38.0 / sample_rate
    alpha = math.sin(w0) / 2 / 0.5
    y = _biquad(y, [(1 + math.cos(w0)) / 2, -1 - math.cos(w0), (1 + math.cos(w0)) / 2], [1 + alpha, -2 * math.cos(w0), 1 - alpha])
    nblk = (len(y) - gate) // step + 1
    e = np.array([np.mean(y[b * step:b * step + gate] ** 2) for b in range(nblk)])
    l = -0.691 + 10 * np.log10(e)
    g1 = l > -70.0
    gamma_rel = -0.691 + 10 * np.log10(e[g1].mean()) - 10
    g2 = g1 & (l > gamma_rel)
    return float(-0.691 + 10 * np.log10(e[g2].mean()))


def normalize_loudness(wav: torch.Tensor, sample_rate: int, loudness_headroom_db: float = 12, loudness_compressor: bool = False,
                       energy_floor: float = 2e-3) -> torch.Tensor:
    """data_utils.py:347-387 followed by _clip_wav (:389-404), as normalize_audio(strategy='loudness') runs them (:453-458)."""
    energy = wav.pow(2).mean().sqrt().item()
    out = wav
    if energy >= energy_floor and wav.shape[-1] >= int(round(0.4 * sample_rate)):
        gain = 10.0 ** ((-loudness_headroom_db - loudness_lkfs(wav, sample_rate)) / 20.0)
        out = gain * wav
        if loudness_compressor:
            out = torch.tanh(out)
    return out.clamp(-1, 1)


def scale_audio(audio: torch.Tensor, strategy: str = "clip", sample_rate: int = 44100, db: float = 6.0) -> torch.Tensor:
    if audio.dtype not in [torch.float32, torch.int32, torch.int16, torch.uint8]:      # generate.py:446-453
        audio = audio.to(torch.float32)
    audio = normalize_audio(audio, strategy=strategy, peak_clip_headroom_db=db)         # :455-458
    return audio.reshape(1, -1).to("cpu")                                               # :459
