"""Delay-pattern bookkeeping, numpy integers (oracle — see oracle/__init__.py).

Follows /root/reference/models/modules/misc/codebook_patterns.py:
  layout of the delayed pattern        :390-406  step s (s>=1) holds (t = s-1-q, q) for every q with t >= 0
                                                 (``delays = range(n_q)``, :377-378); step 0 is empty
  build indexes + mask                 :137-178  flat gather index t + q*T, invalid -> K*T (= special slot)
  build_pattern_sequence               :180-207
  revert indexes + mask                :209-258  flat gather index s + q*S, invalid -> K*S
  revert_pattern_sequence              :260-285
  get_first_step_with_timesteps        :131-135  first step whose coordinates contain timestep t

``*_loops`` functions walk the layout exactly like the reference does (small sizes only);
the closed forms are what the HIP path implements and are checked against the loops and
against goldens made by the reference.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def delayed_layout(n_q: int, timesteps: int) -> List[List[Tuple[int, int]]]:
    """List over sequence steps of (t, q) coordinates; codebook_patterns.py:390-406 with
    delays = [0..n_q-1], flatten_first = 0, empty_initial = 0."""
    out: List[List[Tuple[int, int]]] = [[]]
    for t in range(0, timesteps + (n_q - 1)):
        out.append([(t - q, q) for q in range(n_q) if t - q >= 0])
    return out


def build_indexes_loops(n_q: int, timesteps: int):
    lay = delayed_layout(n_q, timesteps)
    S = len(lay)
    idx = np.full((n_q, S), n_q * timesteps, dtype=np.int64)
    mask = np.zeros((n_q, S), dtype=bool)
    for s, coords in enumerate(lay):
        for (t, q) in coords:
            if t < timesteps:
                idx[q, s] = t + q * timesteps
                mask[q, s] = True
    return idx, mask


def build_indexes(n_q: int, timesteps: int):
    """Closed form: S = T + n_q; (q, s) valid iff 0 <= s-1-q < T."""
    S = timesteps + n_q
    s = np.arange(S)[None, :]
    q = np.arange(n_q)[:, None]
    t = s - 1 - q
    mask = (t >= 0) & (t < timesteps)
    idx = np.where(mask, t + q * timesteps, n_q * timesteps).astype(np.int64)
    return idx, mask


def build_sequence(codes: np.ndarray, special: int):
    """codes (B,K,T) -> (seq (B,K,S), idx, mask); codebook_patterns.py:198-207."""
    B, K, T = codes.shape
    idx, mask = build_indexes(K, T)
    flat = np.concatenate([codes.reshape(B, -1), np.full((B, 1), special, dtype=codes.dtype)], axis=1)
    return flat[:, idx.reshape(-1)].reshape(B, K, -1), idx, mask


def revert_indexes_loops(n_q: int, timesteps: int, seq_steps: int):
    lay = delayed_layout(n_q, timesteps)
    idx = np.full((n_q, timesteps), n_q * seq_steps, dtype=np.int64)
    mask = np.zeros((n_q, timesteps), dtype=bool)
    for s, coords in enumerate(lay):
        if s < seq_steps:
            for (t, q) in coords:
                if t < timesteps:
                    idx[q, t] = s + q * seq_steps
                    mask[q, t] = True
    return idx, mask


def revert_indexes(n_q: int, timesteps: int, seq_steps: int):
    """Closed form: code (q, t) sits at step s = t + 1 + q, valid iff s < seq_steps."""
    t = np.arange(timesteps)[None, :]
    q = np.arange(n_q)[:, None]
    s = t + 1 + q
    mask = s < seq_steps
    idx = np.where(mask, s + q * seq_steps, n_q * seq_steps).astype(np.int64)
    return idx, mask


def revert_sequence(seq: np.ndarray, timesteps: int, special: int):
    """seq (B,K,S) -> (codes (B,K,T), idx, mask); codebook_patterns.py:277-285."""
    B, K, S = seq.shape
    idx, mask = revert_indexes(K, timesteps, S)
    flat = np.concatenate([seq.reshape(B, -1), np.full((B, 1), special, dtype=seq.dtype)], axis=1)
    return flat[:, idx.reshape(-1)].reshape(B, K, -1), idx, mask


def first_step_with_timestep(n_q: int, timesteps: int, t: int) -> Optional[int]:
    """codebook_patterns.py:131-135 — first sequence step containing timestep ``t`` (any codebook).
    For the delayed pattern that is the step where codebook 0 holds t: s = t + 1; ``None`` if t
    is not in the layout at all (t >= T + n_q - 1 ... but the layout keeps t' = t - q >= 0 only
    for t < T + n_q - 1, and coordinates with t' >= T still count)."""
    lay = delayed_layout(n_q, timesteps)
    for s, coords in enumerate(lay):
        for (tt, _q) in coords:
            if tt == t:
                return s
    return None
