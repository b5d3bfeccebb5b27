"""Drop-in pattern plugin: ``target: vaura_amd.patterns.DelayedPatternProvider``.

Host-side mirror of /root/reference/models/modules/misc/codebook_patterns.py for the providers the
generate configs use (Delayed / Parallel; the Unrolled/VALLE/MusicLM providers at :422-603 are not
referenced by any config and are out of scope).  The layout bookkeeping (indexes, masks, first step
of a timestep) is closed-form host metadata; moving token VALUES (build / revert of a sequence)
runs on the device through ``vaura_pattern_build`` / ``vaura_pattern_revert`` and has no CPU path.

Delayed pattern, delays d_q (codebook_patterns.py:390-406): sequence step s (s >= 1) of codebook q
holds timestep t = s - 1 - d_q; step 0 is the all-special start step; S = T + max(d) + 1.
"""
from __future__ import annotations

import typing as tp
from functools import lru_cache

import torch

from . import _lib as L
from . import ops


class Pattern:
    def __init__(self, delays: tp.Sequence[int], timesteps: int):
        self.delays = list(delays)
        self.n_q = len(self.delays)
        self.timesteps = timesteps
        self.seq_steps = timesteps + max(self.delays) + 1
        self._unit_delays = self.delays == list(range(self.n_q))

    # ------------------------------------------------------------------ metadata (host)
    @property
    def num_sequence_steps(self) -> int:
        return self.seq_steps - 1

    @property
    def max_delay(self) -> int:
        return max(self.delays)

    def get_first_step_with_timesteps(self, t: int, q: tp.Optional[int] = None) -> tp.Optional[int]:
        """First sequence step whose coordinates contain timestep ``t`` (codebook_patterns.py:131-135)."""
        qs = range(self.n_q) if q is None else [q]
        # the layout lists every (step - 1 - d_q, q) with a non-negative timestep, including t >= T
        steps = [t + 1 + self.delays[k] for k in qs if t + 1 + self.delays[k] < self.seq_steps]
        return min(steps) if steps else None

    def _build_indexes(self, timesteps: int, device) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        s = torch.arange(self.seq_steps, device=device)[None, :]
        d = torch.tensor(self.delays, device=device)[:, None]
        q = torch.arange(self.n_q, device=device)[:, None]
        t = s - 1 - d
        mask = (t >= 0) & (t < timesteps)
        idx = torch.where(mask, t + q * timesteps, torch.full_like(t, self.n_q * timesteps))
        return idx, mask

    def _revert_indexes(self, seq_steps: int, device) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        t = torch.arange(self.timesteps, device=device)[None, :]
        d = torch.tensor(self.delays, device=device)[:, None]
        q = torch.arange(self.n_q, device=device)[:, None]
        s = t + 1 + d
        mask = s < seq_steps
        idx = torch.where(mask, s + q * seq_steps, torch.full_like(s, self.n_q * seq_steps))
        return idx, mask

    def _revert_logits_indexes(self, seq_steps: int, device) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        """Indexes of the model's OUTPUT positions (is_model_output=True, codebook_patterns.py:240-242): output position s predicts
        sequence step s + 1, i.e. timestep t = s - d_q of codebook q."""
        t = torch.arange(self.timesteps, device=device)[None, :]
        d = torch.tensor(self.delays, device=device)[:, None]
        q = torch.arange(self.n_q, device=device)[:, None]
        s = t + d
        mask = s < seq_steps
        idx = torch.where(mask, s + q * seq_steps, torch.full_like(s, self.n_q * seq_steps))
        return idx, mask

    def revert_pattern_logits(self, logits: torch.Tensor, special_token: float, keep_only_valid_steps: bool = False):
        """logits (B, card, K, S) on the pattern sequence -> (values (B, card, K, T), indexes (K, T), mask (K, T)) —
        codebook_patterns.py:287-313 (the training loss's view of the logits, vaura_model.py:185-187).  A float gather with no
        place on the decode path: plain torch indexing on whatever device the logits live on."""
        assert not keep_only_valid_steps
        B, card, K, S = logits.shape
        assert K == self.n_q and S <= self.seq_steps - 1 + 1
        idx, mask = self._revert_logits_indexes(S, logits.device)
        flat = logits.reshape(B, card, -1)
        flat = torch.cat([flat, torch.zeros_like(flat[:, :, :1]) + special_token], dim=-1)
        values = flat[:, :, idx.view(-1)].view(B, card, K, idx.shape[-1])
        return values, idx, mask

    # ------------------------------------------------------------------ values (device)
    def _check(self, x: torch.Tensor):
        if not x.is_cuda:
            raise L.VauraHipError("pattern sequences are built on the HIP device only (no CPU path)")
        if not self._unit_delays:
            raise L.VauraHipError("the HIP kernels implement delays = 0..K-1 (DelayedPatternProvider default)")

    def build_pattern_sequence(self, z: torch.Tensor, special_token: int, keep_only_valid_steps: bool = False):
        """z (B, K, T) -> (values (B, K, S), indexes (K, S), mask (K, S)) — codebook_patterns.py:180-207."""
        assert not keep_only_valid_steps, "keep_only_valid_steps is a training-time option (out of scope)"
        B, K, T = z.shape
        assert K == self.n_q and T <= self.timesteps
        self._check(z)
        if T != self.timesteps:
            raise L.VauraHipError("build_pattern_sequence expects T == pattern timesteps")
        idx, mask = self._build_indexes(T, z.device)
        return ops.pattern_build(z, special_token), idx, mask

    def revert_pattern_sequence(self, s: torch.Tensor, special_token: int, keep_only_valid_steps: bool = False):
        """s (B, K, S') -> (values (B, K, T), indexes (K, T), mask (K, T)) — codebook_patterns.py:260-285."""
        assert not keep_only_valid_steps
        B, K, S = s.shape
        assert K == self.n_q and S <= self.seq_steps
        self._check(s)
        idx, mask = self._revert_indexes(S, s.device)
        return ops.pattern_revert(s, self.timesteps, special_token), idx, mask


class DelayedPatternProvider:
    def __init__(self, n_q: int, delays: tp.Optional[tp.List[int]] = None, flatten_first: int = 0,
                 empty_initial: int = 0):
        assert n_q > 0
        if flatten_first or empty_initial:
            raise NotImplementedError("flatten_first / empty_initial are not used by any V-AURA config")
        self.n_q = n_q
        self.delays = list(range(n_q)) if delays is None else list(delays)
        assert len(self.delays) == n_q and sorted(self.delays) == self.delays
        self.get_pattern = lru_cache(100)(self.get_pattern)  # type: ignore

    def get_pattern(self, timesteps: int) -> Pattern:
        return Pattern(self.delays, timesteps)


class ParallelPatternProvider(DelayedPatternProvider):
    def __init__(self, n_q: int):
        super().__init__(n_q, [0] * n_q)
