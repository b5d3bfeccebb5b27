"""The step after the hot path: post-codec audio scaling and the wav write of
``save_results`` (/root/reference/scripts/generate.py:392-461, utils/data_utils.py:407-466).

``normalize_audio`` / ``scale_audio`` keep the reference's names, keywords and return conventions; the arithmetic
runs in libvaura_hip.so (``vaura_audio_normalize`` / ``vaura_audio_loudness``) on the device tensor the codec produced — there is no
CPU path.  The 'loudness' strategy (``scale_audio``'s own default) is ITU-R BS.1770-4 integrated loudness as the reference's dependency
torchaudio 2.2.1 computes it (``transforms.Loudness``) — third-party and absent here, so it is restated from the published algorithm
and its parity is UNPINNED, like DAC's; the mp4 mux (PyAV) is host I/O outside this package.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L

_STRATEGIES = {"clip": 0, "peak": 1, "rms": 2, "": 3, "none": 3}


def normalize_audio(wav: torch.Tensor, normalize: bool = True, strategy: str = "peak", peak_clip_headroom_db: float = 6,
                    rms_headroom_db: float = 18, loudness_headroom_db: float = 12, loudness_compressor: bool = False,
                    log_clipping: bool = False, sample_rate: Optional[int] = None, stem_name: Optional[str] = None) -> torch.Tensor:
    """wav (C=1, N) or (B, 1, N) fp32 on a HIP device -> same shape; statistics are per clip (leading dims)."""
    if strategy == "loudness":
        assert sample_rate is not None, "Loudness normalization requires sample rate."        # data_utils.py:454
        return _normalize_loudness(wav, int(sample_rate), float(loudness_headroom_db), bool(loudness_compressor))
    if strategy not in _STRATEGIES:
        raise AssertionError(f"Unexpected strategy: '{strategy}'")
    if not wav.is_cuda:
        raise L.VauraHipError("normalize_audio runs on the HIP device that holds the decoded waveform; there is no CPU path")
    if wav.dim() >= 2 and wav.shape[-2] != 1:
        raise NotImplementedError("multi-channel audio: the codec is mono (dac_8kbps_wrapper.yaml)")
    x = wav.to(torch.float32).contiguous()
    n = x.shape[-1]
    clips = x.numel() // n
    out = torch.empty_like(x)
    scratch = torch.empty(L.lib().vaura_audio_scratch_elems(clips), dtype=torch.float32, device=x.device)
    L.check(L.lib().vaura_audio_normalize(L.ptr(x), L.ptr(out), clips, n, _STRATEGIES[strategy], int(bool(normalize)),
                                          float(peak_clip_headroom_db), float(rms_headroom_db), L.ptr(scratch),
                                          L.current_stream()), "vaura_audio_normalize")
    if strategy in ("", "none") :
        assert bool(out.abs().max() < 1)        # data_utils.py:460
    return out


def _normalize_loudness(wav: torch.Tensor, sample_rate: int, loudness_headroom_db: float, loudness_compressor: bool,
                        energy_floor: float = 2e-3) -> torch.Tensor:
    """normalize_loudness + _clip_wav (utils/data_utils.py:347-404): gain every clip to -loudness_headroom_db LKFS, optional tanh
    compressor, clamp to [-1, 1]; a clip below ``energy_floor`` rms or shorter than one 400 ms gating block is only clamped (the
    reference returns it unchanged from normalize_loudness — its unfold raises on the short one — and then clips)."""
    if not wav.is_cuda:
        raise L.VauraHipError("normalize_audio runs on the HIP device that holds the decoded waveform; there is no CPU path")
    if wav.dim() >= 2 and wav.shape[-2] != 1:
        raise NotImplementedError("multi-channel audio: the codec is mono (dac_8kbps_wrapper.yaml)")
    x = wav.to(torch.float32).contiguous()
    n = x.shape[-1]
    clips = x.numel() // n
    out = torch.empty_like(x)
    scratch = torch.empty(L.lib().vaura_audio_loudness_scratch_elems(clips), dtype=torch.float32, device=x.device)
    L.check(L.lib().vaura_audio_loudness(L.ptr(x), L.ptr(out), clips, n, sample_rate, loudness_headroom_db, int(loudness_compressor),
                                         float(energy_floor), L.ptr(scratch), L.current_stream()), "vaura_audio_loudness")
    g = scratch[:clips]
    out.loudness_untouched = g < 0                # the library marks clips the reference leaves alone (quiet / too short / no gated block) with a negative gain
    out.loudness_gains = torch.where(g < 0, torch.ones_like(g), g)       # the gains that were applied (1 = left alone)
    return out


def scale_audio(audio: torch.Tensor, strategy: str = "loudness", sample_rate: int = 24000, db: float = 6.0) -> torch.Tensor:
    """scripts/generate.py:440-461: one clip -> (1, N) tensor on the CPU, ready to be written."""
    if audio.dtype not in [torch.float32, torch.int32, torch.int16, torch.uint8]:
        audio = audio.to(torch.float32)
    audio = normalize_audio(audio, strategy=strategy, sample_rate=sample_rate, peak_clip_headroom_db=db)
    return audio.reshape(1, -1).to("cpu")


def save_wav(path: str, audio: torch.Tensor, sample_rate: int = 44100) -> None:
    """The ``torchaudio.save(audio_path, audio, fps)`` of save_results (generate.py:421): (1, N) fp32 -> 32-bit float wav."""
    from scipy.io import wavfile
    wavfile.write(path, int(sample_rate), audio.detach().to("cpu", torch.float32).reshape(-1).numpy())
