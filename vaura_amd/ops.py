"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/vaura_hip.h).
All tensors must live on a HIP device; nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.VauraHipError("HIP path only: tensor is not on a HIP device (no CPU fallback)")


def pack_weight(w: torch.Tensor, wdtype: int) -> torch.Tensor:
    _cuda(w)
    N, K = w.shape
    src = w.float().contiguous()
    dst = torch.empty(L.lib().vaura_packed_weight_bytes(N, K, wdtype), dtype=torch.uint8, device=w.device)
    L.check(L.lib().vaura_pack_weight(L.ptr(src), L.ptr(dst), N, K, wdtype, L.current_stream()), "vaura_pack_weight")
    torch.cuda.current_stream().synchronize()
    return dst


def pack_rows(x: torch.Tensor) -> torch.Tensor:
    _cuda(x)
    rows, Cc = x.shape
    src = x.float().contiguous()
    dst = torch.empty(((rows + 15) // 16 * 16) * Cc, dtype=torch.float32, device=x.device)
    L.check(L.lib().vaura_pack_rows(L.ptr(src), L.ptr(dst), rows, Cc, L.current_stream()), "vaura_pack_rows")
    torch.cuda.current_stream().synchronize()
    return dst


def unpack_rows(xp: torch.Tensor, rows: int, Cc: int) -> torch.Tensor:
    _cuda(xp)
    dst = torch.empty(rows, Cc, dtype=torch.float32, device=xp.device)
    L.check(L.lib().vaura_unpack_rows(L.ptr(xp), L.ptr(dst), rows, Cc, L.current_stream()), "vaura_unpack_rows")
    return dst


def gemv(wp: torch.Tensor, wdtype: int, xp: torch.Tensor, rows: int, N: int, K: int, epilogue: int = L.EPI_STORE,
         gain: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """Returns packed rows (rows x N') except for EPI_LOGITS (row-major (rows, N))."""
    _cuda(wp, xp, gain, residual)
    n_out = N // 2 if epilogue == L.EPI_SWIGLU else N
    rp = (rows + 15) // 16 * 16
    if epilogue == L.EPI_LOGITS:
        out = torch.zeros(rows, N, dtype=torch.float32, device=xp.device)
    else:
        out = torch.zeros(rp * n_out, dtype=torch.float32, device=xp.device)
    L.check(L.lib().vaura_gemv(L.ptr(wp), wdtype, L.ptr(xp), L.ptr(gain), L.ptr(residual), L.ptr(out), rows, N, K,
                               epilogue, eps, L.current_stream()), "vaura_gemv")
    return out


def attention_step(qkv_p: torch.Tensor, rope: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, rows: int,
                   n_head: int, head_dim: int, pos: int) -> torch.Tensor:
    _cuda(qkv_p, rope, kcache, vcache)
    max_len = kcache.shape[-2]
    out = torch.zeros(((rows + 15) // 16 * 16) * n_head * head_dim, dtype=torch.float32, device=qkv_p.device)
    L.check(L.lib().vaura_attention_step(L.ptr(qkv_p), L.ptr(rope), L.ptr(kcache), L.ptr(vcache), L.ptr(out), rows,
                                         n_head, head_dim, max_len, pos, L.current_stream()), "vaura_attention_step")
    return out


def attention_step_split(qkv_p: torch.Tensor, rope: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, rows: int,
                         n_head: int, head_dim: int, pos: int, n_split: int) -> torch.Tensor:
    """attention_step with the cached range of every (row, head) split over `n_split` workgroups + a combine pass."""
    _cuda(qkv_p, rope, kcache, vcache)
    max_len = kcache.shape[-2]
    out = torch.zeros(((rows + 15) // 16 * 16) * n_head * head_dim, dtype=torch.float32, device=qkv_p.device)
    part = torch.empty(rows * n_head * n_split * (head_dim + 8), dtype=torch.float32, device=qkv_p.device)
    L.check(L.lib().vaura_attention_step_split(L.ptr(qkv_p), L.ptr(rope), L.ptr(kcache), L.ptr(vcache), L.ptr(out),
                                               L.ptr(part), rows, n_head, head_dim, max_len, pos, n_split,
                                               L.current_stream()), "vaura_attention_step_split")
    return out


def sample(logits: torch.Tensor, batch: int, *, use_sampling: bool, temp: float = 1.0, top_k: int = 0,
           top_p: float = 0.0, cfg_scale: float = 1.0, noise: Optional[torch.Tensor] = None, seed: int = 0,
           clip_base: int = 0, step: int = 0, input_is_probs: bool = False) -> torch.Tensor:
    """logits (rows, K, V) with rows = batch (or 2*batch when cfg_scale > 1) -> tokens (batch, K, 1) int64.
    ``input_is_probs``: the rows already are probabilities (no temperature / softmax / CFG mix)."""
    _cuda(logits, noise)
    rows, K, V = logits.shape
    lg = logits.float().contiguous()
    sp = L.Sampling(int(use_sampling), float(temp), int(top_k), float(top_p), float(cfg_scale), int(seed), int(clip_base),
                    int(bool(input_is_probs)), 0)
    out = torch.zeros(batch, K, dtype=torch.int32, device=logits.device)
    nz = None if noise is None else noise.float().contiguous()
    L.check(L.lib().vaura_sample(L.ptr(lg), batch, K, V, C.byref(sp), L.ptr(nz), step, L.ptr(out), L.current_stream()),
            "vaura_sample")
    torch.cuda.current_stream().synchronize()
    return out.to(torch.int64)[..., None]


def pattern_build(codes: torch.Tensor, special: int) -> torch.Tensor:
    _cuda(codes)
    B, K, T = codes.shape
    ci = codes.to(torch.int32).contiguous()
    seq = torch.empty(B, K, T + K, dtype=torch.int32, device=codes.device)
    L.check(L.lib().vaura_pattern_build(L.ptr(ci), L.ptr(seq), B, K, T, special, L.current_stream()), "vaura_pattern_build")
    torch.cuda.current_stream().synchronize()
    return seq.to(codes.dtype)


def pattern_revert(seq: torch.Tensor, timesteps: int, fill: int) -> torch.Tensor:
    _cuda(seq)
    B, K, S = seq.shape
    si = seq.to(torch.int32).contiguous()
    codes = torch.empty(B, K, timesteps, dtype=torch.int32, device=seq.device)
    L.check(L.lib().vaura_pattern_revert(L.ptr(si), L.ptr(codes), B, K, timesteps, S, fill, L.current_stream()),
            "vaura_pattern_revert")
    torch.cuda.current_stream().synchronize()
    return codes.to(seq.dtype)


def split_rows(xp: torch.Tensor, rows: int, Cc: int, gain: Optional[torch.Tensor] = None, want_ss: bool = False):
    """packed rows fp32 -> (split rows int16 tensor, partial sums of squares or None)."""
    _cuda(xp, gain)
    rp = (rows + 15) // 16 * 16
    dst = torch.zeros(rp * 2 * Cc, dtype=torch.int16, device=xp.device)
    ss = torch.zeros((rp // 16) * (Cc // 16) * 16, dtype=torch.float32, device=xp.device) if want_ss else None
    L.check(L.lib().vaura_split_rows(L.ptr(xp), L.ptr(dst), L.ptr(gain), L.ptr(ss), rows, Cc, L.current_stream()),
            "vaura_split_rows")
    return dst, ss


def unsplit_rows(sp: torch.Tensor, rows: int, Cc: int) -> torch.Tensor:
    """split rows -> (2, rows, C) fp32 values of the hi and lo fp16 planes (layout check helper; plain tensor ops on the device)."""
    rp = (rows + 15) // 16 * 16
    f = sp.view(torch.float16).view(rp // 16, 2, Cc // 8, 16, 8).float()
    return f.permute(1, 0, 3, 2, 4).reshape(2, rp, Cc)[:, :rows]


def gemv_pair(wp: torch.Tensor, x_split: torch.Tensor, rows: int, N: int, K: int, epilogue: int = L.EPI_STORE,
              ss_in: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
              gain_out: Optional[torch.Tensor] = None, want_split: bool = False, want_ss: bool = False, eps: float = 1e-5,
              wdtype: int = L.W_H1, out_khalf2: Optional[torch.Tensor] = None):
    """Returns (out, out_split, ss_out); out is packed rows, or row-major (rows, N) for EPI_LOGITS."""
    _cuda(wp, x_split, ss_in, residual, gain_out)
    n_out = N // 2 if epilogue == L.EPI_SWIGLU else N
    rp = (rows + 15) // 16 * 16
    dev = x_split.device
    out = torch.zeros(rows, N, dtype=torch.float32, device=dev) if epilogue == L.EPI_LOGITS else \
        torch.zeros(rp * n_out, dtype=torch.float32, device=dev)
    osp = torch.zeros(rp * 2 * n_out, dtype=torch.int16, device=dev) if want_split else None
    oss = torch.zeros((rp // 16) * (n_out // 16) * 16, dtype=torch.float32, device=dev) if want_ss else None
    n_ss = 0 if ss_in is None else K // 16
    L.check(L.lib().vaura_gemv_pair(L.ptr(wp), wdtype, L.ptr(x_split), L.ptr(ss_in), n_ss, L.ptr(residual), L.ptr(out), L.ptr(out_khalf2), L.ptr(osp),
                                    L.ptr(gain_out), L.ptr(oss), rows, N, K, epilogue, eps, L.current_stream()),
            "vaura_gemv_pair")
    return out, osp, oss
