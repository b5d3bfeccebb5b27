"""The caller on the input side of the hot path: single-chunk / sliding-window generation
as driven by the reference's script (/root/reference/scripts/generate.py:236-370).

Durations beyond what one pass covers (2.56 s for the released configuration, :222-224) are generated in chunks:
every later chunk is prompted with the tail of the previous one (``max_gen_len - stride_tokens`` tokens) and
produces ``stride_tokens`` new ones; the video segments of a chunk are ``positions % n_segments`` (:336-341).  The
prompt of a chunk is teacher-forced by the batched prefill pass of the engine (32 positions per weight stream),
the rest by the captured decode-step graph; the waveform of the whole clip is decoded once at the end (:366-369).
"""
from __future__ import annotations

from math import ceil
from typing import List, Optional

import torch

COMPRESSION_MODEL_FRAME_RATE = 86   # scripts/generate.py:30


def chunk_schedule(duration: float, model_max_duration: float = 2.56, stride: float = 0.64, vfps: float = 25) -> List[dict]:
    """The (offset, length, video positions, prompt length) of every model.generate() call the reference's loop makes
    (scripts/generate.py:236-237, 304-365) — pure bookkeeping, no tensors."""
    total_gen_len = int(duration * COMPRESSION_MODEL_FRAME_RATE)
    stride_tokens = int(COMPRESSION_MODEL_FRAME_RATE * stride)
    if duration <= model_max_duration:
        return [dict(offset=0, max_gen_len=total_gen_len, prompt_len=0, positions=None, new_tokens=total_gen_len)]
    assert stride is not None, "Stride should be defined to generate beyond max_duration"
    assert stride < model_max_duration, "Cannot stride by more than max generation duration."
    out, current_gen_offset, prompt_length = [], 0, 0
    while current_gen_offset + prompt_length < total_gen_len:
        time_offset = current_gen_offset / COMPRESSION_MODEL_FRAME_RATE
        chunk_duration = min(duration - time_offset, model_max_duration)
        max_gen_len = ceil(chunk_duration * COMPRESSION_MODEL_FRAME_RATE)
        initial_position = ceil(time_offset * vfps)
        video_target_length = ceil(chunk_duration * vfps)
        out.append(dict(offset=current_gen_offset, max_gen_len=max_gen_len, prompt_len=prompt_length,
                        positions=(initial_position // 16, (initial_position + video_target_length) // 16),
                        new_tokens=max_gen_len - prompt_length))
        prompt_length = max_gen_len - stride_tokens
        current_gen_offset += stride_tokens
    return out


@torch.no_grad()
def generate_long(model, frames: torch.Tensor, duration: float, *, stride: float = 0.64, model_max_duration: Optional[float] = None,
                  vfps: float = 25, frame_step: int = 1, clip_indices=None, use_sampling: bool = True, temp: float = 1.0,
                  top_k: int = 128, top_p: float = 0.0, cfg_scale: float = 1.0) -> dict:
    """frames: whatever the feature-extractor plugin accepts, segments on dim 1 — raw (B, S, C, T, H, W) or, with the
    pass-through ``MotionFormer``, features (B, S, t, 768).  Returns {"generated_audio", "sampled_indices"}."""
    if model_max_duration is None:   # scripts/generate.py:221-226
        model_max_duration = 2.56 if model.sampler.config.block_size > 64 else 0.64
    sched = chunk_schedule(duration, model_max_duration, stride, vfps)
    kw = dict(clip_indices=clip_indices, return_sampled_indices=True, use_sampling=use_sampling, temp=temp, top_k=top_k,
              top_p=top_p, remove_prompts=False, prompt_is_encoded=True, cfg_scale=cfg_scale)
    if len(sched) == 1 and sched[0]["positions"] is None:     # single chunk (:309-324)
        selected = frames[:, :, ::frame_step, ...] if frame_step != 1 else frames
        item = model.generate(frames=selected, audio=None, max_new_tokens=sched[0]["max_gen_len"], **kw)
        return {"generated_audio": item["generated_audio"], "sampled_indices": item["sampled_indices"]}
    stride_tokens = int(COMPRESSION_MODEL_FRAME_RATE * stride)
    all_tokens, prompt_tokens = [], None
    for ch in sched:                                            # chunked generation (:327-365)
        lo, hi = ch["positions"]
        positions = torch.arange(lo, hi, device=frames.device)
        selected = frames[:, positions % frames.shape[1], ...]
        if frame_step != 1:
            selected = selected[:, :, :, ::frame_step, ...]
        # tokens only: the reference decodes every chunk inside generate() and throws the audio away (:344-357)
        gen_tokens = model.generate_tokens(frames=selected, audio=prompt_tokens, max_new_tokens=ch["max_gen_len"], **kw)
        all_tokens.append(gen_tokens if prompt_tokens is None else gen_tokens[:, :, prompt_tokens.shape[-1]:])
        prompt_tokens = gen_tokens[:, :, stride_tokens:]
    gen_tokens = torch.cat(all_tokens, dim=-1)
    audio = model.audio_encoder.decode([(gen_tokens[..., : model.num_codebooks, :], None)])   # :366-369
    return {"generated_audio": audio, "sampled_indices": gen_tokens}
