"""Host-side owners of device state for the HIP hot path (torch = memory + streams only).

``DecoderEngine``  packs a sampler state dict (reference key names) into the layouts of
                   include/vaura_hip.h, owns K/V cache + workspaces, and drives
                   vaura_prefill_cond / vaura_pattern_* / vaura_generate_loop.
``CodecEngine``    folds weight-norm of a DAC state dict and drives vaura_dac_decode.

Neither has a CPU path: construction fails if libvaura_hip.so cannot be loaded or the
tensors are not on a HIP device.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import Dict, Optional, Sequence

import torch

from . import _lib as L
from .synth import CodecCfg, SamplerCfg, fold_weight_norm


def rope_table(n_pos: int, head_dim: int, base: int = 10000) -> torch.Tensor:
    """cos/sin table with the reference's own construction (llama.py:593-603) so that the
    table entries are bit-identical to what the reference multiplies by."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(n_pos), inv)
    z = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([z.real, z.imag], dim=-1).contiguous()


_PRIVATE_STREAMS: Dict[torch.device, "torch.cuda.Stream"] = {}


@contextlib.contextmanager
def off_null_stream(dev: torch.device):
    """Run the enclosed enqueues on a private stream when the caller sits on HIP's legacy null stream.
    hipGraph capture is impossible there, and every launch on it synchronises implicitly with all blocking
    streams (measured: 268 -> 255 ms per configs[1] batch once the whole pass left the null stream).
    Ordered after the caller's pending work and before its later work.  Yields the caller's stream when a
    switch happened (results allocated inside must ``record_stream`` it), else None."""
    dev = torch.device(dev)
    cur = torch.cuda.current_stream(dev)
    if cur.cuda_stream != 0:
        yield None
        return
    side = _PRIVATE_STREAMS.get(dev)
    if side is None:
        side = _PRIVATE_STREAMS[dev] = torch.cuda.Stream(dev)
    side.wait_stream(cur)
    try:
        with torch.cuda.stream(side):
            yield cur
    finally:
        # also on an exception: kernels already enqueued on the side stream must still be ordered before later null-stream work
        cur.wait_stream(side)


def h_row_scales(w: torch.Tensor) -> torch.Tensor:
    """The power-of-two row scales of the fp16-plane storages (csrc/gemv3.hip h_row_scale_kernel): 2^(e - 14) with
    max|W[n, :]| = m * 2^e, m in [0.5, 1), i.e. max|W[n, :]| / scale in [2^13, 2^14); 1 for an all-zero row."""
    amax = w.detach().float().abs().amax(dim=1)
    _, e = torch.frexp(amax)
    return torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), e - 14), torch.ones_like(amax))


def h_effective_weight(w: torch.Tensor, planes: int) -> torch.Tensor:
    """The real-number matrix that ``planes`` fp16 planes hold: hi = fp16(W / scale), lo = fp16(W / scale - hi), times scale."""
    w = w.detach().float()
    sc = h_row_scales(w)[:, None]
    v = w / sc
    hi = v.half().float()
    q = hi if planes == 1 else hi + (v - hi).half().float()
    return q * sc


def h1_lossless(w: torch.Tensor) -> bool:
    """True when ONE fp16 plane (with its power-of-two row scale) holds every element of ``w``: half the HBM bytes, same real
    numbers.  bf16-representable weights qualify down to 2^-31 of their row's largest; a smaller element is held to 2^-39 of
    that maximum — below the resolution of any fp32 sum it enters — and does not disqualify the matrix."""
    w = w.detach().float()
    err = (h_effective_weight(w, 1) - w).abs()
    tol = w.abs().amax(dim=1, keepdim=True) * 2.0 ** -38
    return bool((err <= tol).all())


def fold_gain(g: torch.Tensor, plane_shift: int = 0):
    """(g 2^-E, 2^E) with E = max(0, ceil(log2 max|g|)) + plane_shift: the RMSNorm gain as the planes carry it and the factor its
    consuming matrix takes instead (both exact: powers of two)."""
    g = g.detach().float()
    m = float(g.abs().max())
    e = (max(0, math.ceil(math.log2(m))) if m > 0 and math.isfinite(m) else 0) + int(plane_shift)
    return g * (2.0 ** -e), 2.0 ** e


def streamed_matrices(sd: Dict[str, torch.Tensor]):
    """The matrices the decode loop streams every step (98 % of the bytes)."""
    from .synth import is_streamed_weight
    return [k for k in sd if is_streamed_weight(k)]


WEIGHT_DTYPES = ("h1", "h2", "fp8", "fp8h", "f32")


def resolve_weight_dtype(sd: Dict[str, torch.Tensor], wdtype: str) -> str:
    """Storage of the streamed matrices:
      "h2"   (hi, lo) fp16 planes, 22 significand bits, 4 bytes per weight: any fp32 checkpoint on the pair kernels — what
             "auto" picks for a real V-AURA checkpoint (fp32 master weights; the reference runs the sampler in fp32,
             configs/vaura_defaults.yaml `precision: 32`).  Logits within ~1e-5 of the fp32 reference, greedy tokens of every
             reference golden unchanged.
      "h1"   one fp16 plane, 2 bytes per weight — "auto" picks it when that loses nothing (``h1_lossless``: e.g. a
             bf16-representable checkpoint); forced on another checkpoint it ROUNDS the weights to 11 bits (not token-exact).
      "fp8"  e4m3 + row scales for the per-layer matrices (BASELINE configs[4]; a different model), multiplied against both
             activation planes: the "h1" arithmetic on ``quant.fp8_effective_state_dict(sd)``, token-exact against the oracle on it.
      "fp8h" (round 6) the same packed bytes against the HI activation plane only — 11-bit activations under 4-bit-significand
             weights, half the plane bytes each CU takes in: configs[4]'s measured configuration; tolerance against "fp8" / "h1"
             is REPORTED (tests, bench.py), not bit parity.
      "f32"  fp32 tiles on the exact-fp32-MFMA GEMVs (gemv_kernel.h): bit-for-bit fp32 products, 1/16 of the MFMA rate —
             the cross-check the pair kernels are tested against.
    ("bf16" is accepted as an alias of "h1": round 2's name for the 2-byte storage.)"""
    if wdtype == "bf16":
        wdtype = "h1"
    if wdtype != "auto":
        if wdtype not in WEIGHT_DTYPES:
            raise L.VauraHipError(f"weight_dtype must be auto | {' | '.join(WEIGHT_DTYPES)}, got {wdtype!r}")
        return wdtype
    return "h1" if all(h1_lossless(sd[k]) for k in streamed_matrices(sd)) else "h2"


def _require_cuda(dev: torch.device):
    if torch.device(dev).type != "cuda":
        raise L.VauraHipError(f"the V-AURA hot path runs on a HIP device only (got {dev}); there is no CPU fallback")
    L.lib()


# Near-tie detector (csrc/step.hip sample_kernel, include/vaura_hip.h vaura_sampling.tie_eps): relative bound on the error of a logit the
# plane storages deliver against the reference's fp32 (x the row's largest |logit|, x (2 cfg - 1) through the CFG mix).  Calibrated on
# the reference's own goldens (tools/near_tie_sweep.py -> profiles/r06_near_tie_sweep.txt; tests/test_gpu_generate.py::test_near_tie_*):
# bounds in [1.2e-6, 2e-6) flag the two literal near-ties the goldens hold (configs[3] step 578: margin 5.5e-6; the later chunk's step
# 175: 3.8e-6) and nothing else in any golden run (4 104 sampled + 4 104 greedy cfg-6 decisions, 7 920 + 1 134 greedy ones).
NEAR_TIE_EPS = 1.5e-6


class DecoderEngine:
    # hand-off epochs carry 10 bits of this counter (csrc/common.h va_handoff_epoch) and the arrival words of the in-launch hand-offs live
    # in LDS, which outlives processes: start every process somewhere else, so that two processes taking turns on one GPU do not
    # produce identical tag sequences
    _sequence_id = (os.getpid() * 0x9E37 + int.from_bytes(os.urandom(2), "little")) & 0x3FF

    def __init__(self, cfg: SamplerCfg, sd: Dict[str, torch.Tensor], device="cuda:0", wdtype: str = "auto",
                 one_launch_mlp: bool = True, range_fallback: bool = True, plane_shift: int = 0,
                 near_tie: str = "report", near_tie_eps: Optional[float] = None, kv_dtype: str = "f32"):
        """wdtype: storage of the streamed matrices — "auto" | "h1" | "h2" | "fp8" | "fp8h" | "f32" (``resolve_weight_dtype``).
        one_launch_mlp: let the library run w1||w3 -> w2 of a layer as ONE launch with an in-launch hand-off where the shape
        is eligible (1..16 decoder rows, fp16-plane weights, >= 256 CUs; csrc/mlp_engine.h: bit-identical results, -5..7 % on the
        decode loop).  Its consumers wait for producers of the SAME launch, so every workgroup must become resident: pass False
        when several processes share this GPU, or when several DecoderEngines of ONE process run on different streams at the same
        time (each launch wants all 256 CUs, one workgroup per CU: two such launches can starve each other until their bounded waits
        give up).  A give-up is reported by ``check_status`` — which callers of ``generate_codes`` / ``run`` must call themselves,
        ``VAURAModel.generate_tokens`` does.
        plane_shift: S in 0..24 (fp16-plane storages only).  EVERY activation plane set is stored times 2^-S and the matrix that consumes
        it times 2^S (h planes: through ``fold_gain``; attention output and SwiGLU output: ``vaura_decoder.plane_shift``) — the planes
        overflow at |activation| = 65504 * 2^S instead of 65504, at the same speed.  The price is at the other end of fp16: the lo
        plane's absolute resolution is 2^-24 * 2^S (its subnormal step), so S = 0 is what every parity number is quoted on, and a
        checkpoint KNOWN to carry massive activations is the reason to pass S = 4 .. 8 rather than pay the exact-fp32 twin on every call."""
        self.one_launch_mlp = bool(one_launch_mlp)
        self.handoff_fallbacks = 0           # calls re-run on the separate launches after an in-launch hand-off gave up (GPU shared)
        # near-tie detector: "off" | "report" (count decisions inside the arithmetic's noise: ``near_ties`` / ``last_near_ties``) |
        # "rerun" (``generate_codes_checked`` re-runs a flagged call on the exact-fp32 twin, like an overflow)
        if near_tie not in ("off", "report", "rerun"):
            raise L.VauraHipError(f"near_tie must be off | report | rerun, got {near_tie!r}")
        self.near_tie = near_tie
        self.near_tie_eps = float(NEAR_TIE_EPS if near_tie_eps is None else near_tie_eps)
        self.near_ties = 0                   # decisions flagged so far (all calls)
        self.last_near_ties = (0, None)      # (count, first flagged step) of the last checked call
        self.near_tie_reruns = 0
        # K / V cache: "f32" (every parity number) or "f16" (round 6; the low-precision serving configuration, BASELINE configs[4] together
        # with weight_dtype="fp8h": fp16(rotated k) / fp16(v), half the attention's stream; caches of at most 256 positions, plane storages)
        # "f8": OCP e4m3 bytes, unscaled and saturating — a quarter of the fp32 stream, a ~1e-2-class approximation of the attention (still an
        # order of magnitude below the fp8 weights' own error); an OPTION of that configuration, reported beside "f16"
        if kv_dtype not in ("f32", "f16", "f8"):
            raise L.VauraHipError(f"kv_dtype must be f32 | f16 | f8, got {kv_dtype!r}")
        self.kv_dtype = kv_dtype
        self.plane_shift = int(plane_shift)
        if not 0 <= self.plane_shift <= 24:
            raise L.VauraHipError(f"plane_shift must be in 0..24, got {plane_shift}")
        # Range safety (``generate_codes_checked``): the fp16-plane activation format ends at |x| = 65504; a call that overflows it is
        # DETECTED on the device (sticky status bit) and re-run from its start on a twin engine with weight_dtype="f32" (fp32
        # activations and weights on the exact-fp32 matrix instruction: no range limit, the reference's own arithmetic), built lazily
        # from the same state dict the first time it is needed.  The dict is kept by reference only (no copy).
        self._twin_sd = sd if range_fallback else None
        self._range_twin: Optional["DecoderEngine"] = None
        self.range_fallbacks = 0             # calls re-run on the exact-fp32 twin so far
        self._forward_on_twin = False        # forward_cached moved to the twin for good (the host's per-step calls)
        _require_cuda(device)
        self.cfg = cfg
        self.dev = torch.device(device)
        self.requested_wdtype = wdtype
        wdtype = resolve_weight_dtype(sd, wdtype)
        self.planes = wdtype != "f32"          # activations travel as (hi, lo) fp16 planes; "f32" = the exact-fp32-MFMA step
        if self.plane_shift and not self.planes:
            raise L.VauraHipError("plane_shift applies to the fp16-plane storages (h1 | h2 | fp8) only")
        S, pw = self.plane_shift, 2.0 ** self.plane_shift
        # "fp8": e4m3 + power-of-two row scales for the four per-layer matrices (BASELINE configs[4]); the codebook
        # heads stay one fp16 plane.  The model then IS the one with weights quant.fp8_effective_weight(W): same kernels,
        # same activation arithmetic.
        self.wd = {"f32": L.W_F32, "h1": L.W_H1, "h2": L.W_H2, "fp8": L.W_FP8, "fp8h": L.W_FP8H}[wdtype]
        head_wd = L.W_H1 if wdtype in ("fp8", "fp8h") else self.wd
        self.wdtype = wdtype
        # lossy storages (fp8 / fp8h; one plane FORCED on a checkpoint it cannot hold) are a different model than the state dict: their
        # exact-fp32 twin must hold the numbers the storage holds, or a call that falls back would be decoded by another model than
        # the other calls of the same job.  Resolved lazily in _twin().
        self._twin_lossy_cached: Optional[bool] = None
        self.lib = L.lib()
        D, F, K = cfg.d_model, cfg.ffn_dim, cfg.num_codebooks
        with torch.cuda.device(self.dev):
            self._keep = []  # device tensors referenced by raw pointers
            lw = (L.LayerWeights * cfg.num_layers)()
            for l in range(cfg.num_layers):
                p = f"layers.{l}."
                w1 = sd[p + "feed_forward.w1.weight"].float()
                w3 = sd[p + "feed_forward.w3.weight"].float()
                w13 = torch.stack([w1.view(F // 16, 16, D), w3.view(F // 16, 16, D)], dim=1).reshape(2 * F, D)
                # Norm gains leave the planes (range): rmsnorm's gain g multiplies the residual stream BEFORE it is split into fp16 planes
                # (the producer's epilogue), so a large gain eats the planes' range.  Fold a power of two out of it: the planes hold
                # (g 2^-E) * h with max |g 2^-E| <= 1 and the consuming matrix is stored as W 2^E — exact on both sides (its power-of-two
                # row scales absorb it; W (g * x) == (W 2^E) ((g 2^-E) * x) bit for bit), E = 0 for gains <= 1 (every golden).
                ga, ea = fold_gain(sd[p + "attention_norm.weight"], S)
                gf, ef = fold_gain(sd[p + "ffn_norm.weight"], S)
                lw[l].wqkv = L.ptr(self._pack(sd[p + "attention.wqkv.weight"].float() * ea, self.wd))
                lw[l].wo = L.ptr(self._pack(sd[p + "attention.wo.weight"].float() * pw, self.wd))
                lw[l].w13 = L.ptr(self._pack(w13 * ef, self.wd))
                lw[l].w2 = L.ptr(self._pack(sd[p + "feed_forward.w2.weight"].float() * pw, self.wd))
                lw[l].attn_norm = L.ptr(self._dev(ga))
                lw[l].ffn_norm = L.ptr(self._dev(gf))
            self.layers = lw
            gn, en = fold_gain(sd["norm.weight"], S)
            heads = torch.cat([sd[f"lm_heads.{k}.weight"].float() for k in range(K)], dim=0) * en
            self.heads = self._pack(heads, head_wd)
            self.final_norm = self._dev(gn)
            self.fc1 = self._pack(sd["cls_embeddings.projection.fc1.weight"], L.W_F32)
            self.fc2 = self._pack(sd["cls_embeddings.projection.fc2.weight"], L.W_F32)
            self.uncond = self._dev(sd["cls_embeddings.uncond_embedding"])
            self.empty_video = self._dev(sd["empty_video_emb"].reshape(-1))
            self.tok_emb = self._dev(torch.stack([sd[f"tok_embeddings.{k}.emb.weight"].float() for k in range(K)]))
            self.tok_w = self._dev(torch.stack([
                fold_weight_norm(sd[f"tok_embeddings.{k}.out_proj.weight_g"].float(),
                                 sd[f"tok_embeddings.{k}.out_proj.weight_v"].float())[:, :, 0] for k in range(K)]))
            self.tok_b = self._dev(torch.stack([sd[f"tok_embeddings.{k}.out_proj.bias"].float() for k in range(K)]))
            self.tok_table = torch.empty(K, cfg.d_codebook + 1, cfg.tok_dim, dtype=torch.float32, device=self.dev)
            self._keep.append(self.tok_table)
            L.check(self.lib.vaura_build_token_table(L.ptr(self.tok_emb), L.ptr(self.tok_w), L.ptr(self.tok_b),
                                                     L.ptr(self.tok_table), K, cfg.d_codebook + 1, cfg.codebook_dim,
                                                     cfg.tok_dim, L.current_stream(self.dev)), "vaura_build_token_table")
            torch.cuda.synchronize(self.dev)
        self.dims = L.Dims(cfg.num_layers, D, cfg.nhead, F, K, cfg.d_codebook, cfg.cond_dim, cfg.tok_dim, cfg.cond_in,
                           cfg.codebook_dim, 7, cfg.layer_norm_eps)
        self.dec: Optional[L.Decoder] = None
        self.state = None
        self._carried_status = 0             # status bits read off a state buffer that prepare() was about to replace
        self._shape = None
        self._graph, self._graph_key = None, None
        self.weight_bytes = sum(t.numel() * t.element_size() for t in self._keep)

    # ------------------------------------------------------------------ helpers
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        d = t.detach().to(self.dev, torch.float32).contiguous()
        self._keep.append(d)
        return d

    def _pack(self, w: torch.Tensor, wd: int) -> torch.Tensor:
        N, K = w.shape
        src = w.detach().to(self.dev, torch.float32).contiguous()
        dst = torch.empty(self.lib.vaura_packed_weight_bytes(N, K, wd), dtype=torch.uint8, device=self.dev)
        L.check(self.lib.vaura_pack_weight(L.ptr(src), L.ptr(dst), N, K, wd, L.current_stream(self.dev)), "vaura_pack_weight")
        torch.cuda.current_stream(self.dev).synchronize()  # src may be freed right after
        self._keep.append(dst)
        return dst

    @staticmethod
    def _rows_padded(r: int) -> int:
        return (r + 15) // 16 * 16

    # ------------------------------------------------------------------ per-call state
    # prompt positions teacher-forced per pass (bf16 / fp8 storage); the workspaces scale with it.  192 covers the
    # sliding-window caller's 166-token prompt in one GEMM pass (96 ms per chunk; 106 ms at 64, 120 ms at 32)
    PREFILL_POSITIONS = 192

    def prepare(self, batch: int, timesteps: int, n_cond_tokens: int, cfg_on: bool, tokens_per_frame: int = 7,
                block_size: Optional[int] = None):
        c = self.cfg
        K = c.num_codebooks
        S = timesteps + K
        rows = 2 * batch if cfg_on else batch
        max_len = (max(S, block_size or 0) + 31) // 32 * 32
        key = (batch, timesteps, n_cond_tokens, cfg_on, tokens_per_frame, max_len)
        self._fc = None                       # whoever prepares the engine is about to overwrite the K/V cache
        if self._shape == key:
            return
        if getattr(self, "state", None) is not None:      # an unread status bit must survive the reallocation below
            self._carried_status |= int(self.state[4].item())
        with torch.cuda.device(self.dev):
            pp = min(self.PREFILL_POSITIONS, S) if self.planes else 1
            rp = self._rows_padded(rows) * pp     # decode uses the first position's worth of row blocks
            f32 = dict(dtype=torch.float32, device=self.dev)
            self.rope = rope_table(max_len, c.head_dim, c.rope_base).to(self.dev)
            if self.kv_dtype != "f32":
                if max_len > 256 or not self.planes:
                    raise L.VauraHipError(f"kv_dtype={self.kv_dtype!r} serves caches of at most 256 positions on the fp16-plane storages (got max_len "
                                          f"{max_len}, weights {self.wdtype})")
                self.kcache = torch.zeros(c.num_layers, rows, c.nhead, max_len, c.head_dim, device=self.dev,
                                          dtype=torch.float16 if self.kv_dtype == "f16" else torch.uint8)
            else:
                self.kcache = torch.zeros(c.num_layers, rows, c.nhead, max_len, c.head_dim, **f32)
            self.vcache = torch.zeros_like(self.kcache)
            self.seq = torch.zeros(batch, K, S, dtype=torch.int32, device=self.dev)
            self.state = torch.zeros(8, dtype=torch.int32, device=self.dev)   # include/vaura_hip.h: position, arrivals, step, id, STATUS, spare
            self.ws_h = torch.zeros(rp * c.d_model, **f32)
            self.ws_qkv = torch.zeros(rp * 3 * c.d_model, **f32)
            self.ws_qkv2 = torch.zeros(self._rows_padded(rows) * 3 * c.d_model, **f32)   # decode step only
            self.ws_attn = torch.zeros(rp * c.d_model, **f32)
            self.ws_ffn = torch.zeros(rp * c.ffn_dim, **f32)
            self.ws_logits = torch.zeros(rows, K * c.d_codebook, **f32)
            self._prefill_positions = pp if pp > 1 else 0
            i16 = dict(dtype=torch.int16, device=self.dev)
            self.ws_h_split = torch.zeros(rp * 2 * c.d_model, **i16)       # (hi, lo) fp16 planes
            self.ws_attn_split = torch.zeros(rp * 2 * c.d_model, **i16)
            self.ws_ffn_split = torch.zeros(rp * 2 * c.ffn_dim, **i16)
            self.ws_ss = torch.zeros((rp // 16) * (c.d_model // 16) * 16, **f32)
            self.ws_attn_part = torch.zeros(rows * c.nhead * 8 * (c.d_model // c.nhead + 8), **f32)
            self.ws_sync = torch.zeros(768, dtype=torch.int32, device=self.dev)
            crp = self._rows_padded(rows * n_cond_tokens)
            self.cond_in = torch.zeros(crp * c.cond_in, **f32)
            self.cond_tmp = torch.zeros(crp * c.cond_dim, **f32)
            self.cond_proj = torch.zeros(crp * c.cond_dim, **f32)
            self.codes_i32 = torch.zeros(batch, K, timesteps, dtype=torch.int32, device=self.dev)
        d = L.Decoder()
        d.dims = self.dims
        d.dims.tokens_per_frame = tokens_per_frame
        d.wdtype, d.batch, d.rows, d.max_len = self.wd, batch, rows, max_len
        d.timesteps, d.seq_len, d.n_cond_tokens = timesteps, S, n_cond_tokens
        d.prefill_positions = self._prefill_positions
        d.plane_shift = self.plane_shift
        d.kv_dtype = {"f32": 0, "f16": 1, "f8": 2}[self.kv_dtype]
        d.layers_host = C.cast(self.layers, C.POINTER(L.LayerWeights))
        d.heads, d.final_norm = L.ptr(self.heads), L.ptr(self.final_norm)
        d.tok_emb, d.tok_proj_w, d.tok_proj_b = L.ptr(self.tok_emb), L.ptr(self.tok_w), L.ptr(self.tok_b)
        d.tok_table = L.ptr(self.tok_table)
        d.empty_video, d.rope, d.cond_proj = L.ptr(self.empty_video), L.ptr(self.rope), L.ptr(self.cond_proj)
        d.kcache, d.vcache, d.seq, d.state = L.ptr(self.kcache), L.ptr(self.vcache), L.ptr(self.seq), L.ptr(self.state)
        d.noise = 0
        d.ws_h, d.ws_qkv, d.ws_attn = L.ptr(self.ws_h), L.ptr(self.ws_qkv), L.ptr(self.ws_attn)
        d.ws_qkv2 = L.ptr(self.ws_qkv2) if self.planes else 0
        d.ws_ffn, d.ws_logits = L.ptr(self.ws_ffn), L.ptr(self.ws_logits)
        if self.planes:
            d.ws_h_split, d.ws_attn_split = L.ptr(self.ws_h_split), L.ptr(self.ws_attn_split)
            d.ws_ffn_split, d.ws_ss = L.ptr(self.ws_ffn_split), L.ptr(self.ws_ss)
        else:   # NULL split workspaces select the exact-fp32-MFMA step (api.hip enqueue_step)
            d.ws_h_split = d.ws_attn_split = d.ws_ffn_split = d.ws_ss = 0
        d.first_norm = self.layers[0].attn_norm
        d.ws_attn_part = L.ptr(self.ws_attn_part)
        d.ws_sync = L.ptr(self.ws_sync) if self.one_launch_mlp else 0      # producer flags of the one-launch MLP (csrc/mlp_engine.h); NULL -> two launches
        self.dec = d
        self._shape = key
        self._graph_key = None
        self.batch, self.rows, self.T, self.S, self.Tv, self.max_len = batch, rows, timesteps, S, n_cond_tokens, max_len

    def kv_bytes_per_position(self) -> int:
        c = self.cfg
        return 2 * c.num_layers * self.rows * c.d_model * 4

    def set_condition(self, feats: torch.Tensor):
        """feats (B, Tv, 768) fp32 on device.  Rows [B, 2B) get the CFG null embedding
        (models/vaura_model.py:790-793) when the engine was prepared with cfg_on."""
        assert self.dec is not None
        self._fc = None
        B, Tv, Cin = feats.shape
        assert B == self.batch and Tv == self.Tv and Cin == self.cfg.cond_in
        x = feats.to(self.dev, torch.float32)
        if self.rows == 2 * B:
            if Tv != self.uncond.shape[0]:
                raise L.VauraHipError(f"CFG null embedding has {self.uncond.shape[0]} tokens, condition has {Tv}")
            x = torch.cat([x, torch.zeros_like(x) + self.uncond], dim=0)
        x = x.reshape(self.rows * Tv, Cin).contiguous()
        st = L.current_stream(self.dev)
        n = self.rows * Tv
        L.check(self.lib.vaura_pack_rows(L.ptr(x), L.ptr(self.cond_in), n, Cin, st), "vaura_pack_rows")
        L.check(self.lib.vaura_prefill_cond(C.byref(self.dims), L.ptr(self.cond_in), L.ptr(self.fc1), L.ptr(self.fc2),
                                            L.W_F32, L.ptr(self.cond_tmp), L.ptr(self.cond_proj), n, st),
                "vaura_prefill_cond")
        self._cond_keepalive = x

    def cond_projection(self) -> torch.Tensor:
        """(rows, Tv, cond_dim) row-major view of the hoisted video MLP output (tests)."""
        n = self.rows * self.Tv
        out = torch.empty(n, self.cfg.cond_dim, dtype=torch.float32, device=self.dev)
        L.check(self.lib.vaura_unpack_rows(L.ptr(self.cond_proj), L.ptr(out), n, self.cfg.cond_dim, L.current_stream(self.dev)),
                "vaura_unpack_rows")
        return out.view(self.rows, self.Tv, self.cfg.cond_dim)

    # ------------------------------------------------------------------ generation
    def _sampling(self, use_sampling, temp, top_k, top_p, cfg_scale, seed, clip_base) -> L.Sampling:
        if use_sampling and temp > 0.0 and not top_p > 0.0 and int(top_k) > self.cfg.d_codebook:
            # the reference's sample_top_k is torch.topk(probs, k) (utils/utils.py:172): k beyond the codebook raises there too
            raise L.VauraHipError(f"top_k = {top_k} exceeds the codebook size {self.cfg.d_codebook} (the reference's torch.topk raises as well)")
        tie = self.near_tie_eps if (self.near_tie != "off" and self.planes) else 0.0      # the exact-fp32 engine IS the reference's arithmetic
        return L.Sampling(int(bool(use_sampling)), float(temp), int(top_k), float(top_p),
                          float(cfg_scale if self.rows == 2 * self.batch else 1.0), int(seed), int(clip_base), 0, float(tie))

    def start_sequence(self, prompt: Optional[torch.Tensor]):
        """codes = -1 everywhere but the prompt -> pattern sequence on device; returns Tp."""
        K, T = self.cfg.num_codebooks, self.T
        self.codes_i32.fill_(-1)
        Tp = 0
        if prompt is not None and prompt.shape[-1] > 0:
            Tp = prompt.shape[-1]
            assert Tp < T, "gt audio prompt can not be longer than max_new_tokens"
            self.codes_i32[..., :Tp] = prompt.to(self.dev, torch.int32)
        L.check(self.lib.vaura_pattern_build(L.ptr(self.codes_i32), L.ptr(self.seq), self.batch, K, T,
                                             self.cfg.d_codebook, L.current_stream(self.dev)), "vaura_pattern_build")
        self._reset_state()
        return Tp

    def run(self, n_prefill: int, n_steps: int, sp: L.Sampling, noise: Optional[torch.Tensor] = None,
            use_graph: bool = True):
        """Enqueue `n_prefill` teacher-forced positions and `n_steps` sampled ones (asynchronous).
        With `use_graph` one decode step is captured once into a hipGraph and replayed per step; HIP
        cannot capture on the legacy default stream, so the loop then runs on a private stream that is
        ordered after / before the caller's current stream."""
        self.dec.noise = L.ptr(noise)
        self._noise_keepalive = noise
        use_graph = bool(use_graph and n_steps > 0)
        with (off_null_stream(self.dev) if use_graph else contextlib.nullcontext()):
            st = L.current_stream(self.dev)
            if use_graph:
                key = (self._shape, L.ptr(noise), bytes(sp))
                if self._graph_key != key:       # the captured step is tied to these buffers / parameters
                    self._free_graph()
                    handle = C.c_void_p()
                    L.check(self.lib.vaura_step_graph_build(C.byref(self.dec), C.byref(sp), st, C.byref(handle)),
                            "vaura_step_graph_build")
                    self._graph, self._graph_key = handle, key
            L.check(self.lib.vaura_generate_loop(C.byref(self.dec), C.byref(sp), n_prefill, n_steps,
                                                 self._graph if use_graph else None, st), "vaura_generate_loop")

    def _free_graph(self):
        if getattr(self, "_graph", None):
            self.lib.vaura_step_graph_free(self._graph)
        self._graph, self._graph_key = None, None

    def __del__(self):
        try:
            self._free_graph()
        except Exception:
            pass

    @property
    def _twin_lossy(self) -> bool:
        if self._twin_lossy_cached is None:
            self._twin_lossy_cached = self.wdtype in ("fp8", "fp8h") or (
                self.wdtype == "h1" and self.requested_wdtype != "auto" and self._twin_sd is not None and
                not all(h1_lossless(self._twin_sd[k]) for k in streamed_matrices(self._twin_sd)))
        return self._twin_lossy_cached

    def _twin(self) -> "DecoderEngine":
        """The exact-fp32 engine of the SAME model: fp32 tiles of what this engine's storage holds (the state dict itself for the
        lossless storages; the dequantised matrices for fp8 / a forced single plane).  Built on first use: ~2.7 GB of weights."""
        if self._range_twin is None:
            sd = self._twin_sd
            if self._twin_lossy:
                from . import quant
                if self.wdtype in ("fp8", "fp8h"):
                    sd = quant.fp8_effective_state_dict(sd)
                else:
                    from .synth import is_streamed_weight
                    sd = {k: (h_effective_weight(v, 1) if is_streamed_weight(k) else v) for k, v in sd.items()}
            try:
                self._range_twin = DecoderEngine(self.cfg, sd, self.dev, wdtype="f32", range_fallback=False, near_tie="off")
            except torch.cuda.OutOfMemoryError as e:
                raise L.VauraHipError("the exact-fp32 twin engine (2.7 GB of fp32 tiles + its K/V cache) does not fit next to this engine: "
                                      f"{e}") from e
        return self._range_twin

    def _reset_state(self):
        """position / arrivals / step back to 0; state[3] carries a sequence id (reserved for in-launch hand-off epochs)."""
        self._fc = None                       # position 0 again: a cached forward() prefix no longer matches the K/V cache
        DecoderEngine._sequence_id = (DecoderEngine._sequence_id + 1) & 0x3FF      # 10 bits enter the hand-off epoch (csrc/common.h)
        self.state[:4].zero_()                               # tiny device fills: no host-device synchronisation
        self.state[3:4].fill_(DecoderEngine._sequence_id)    # state[4] (status bits) is sticky until check_status() reads it
        self.state[6:8].zero_()                              # near-tie count / first flagged step of THIS sequence

    def check_status(self):
        """Read AND clear the sticky device status word (one host-device synchronisation; ``VAURAModel.generate`` folds it into
        the reference's own post-condition checks, vaura_model.py:550-572) and raise ONE error naming every condition it holds:
        a non-finite logit at the sampler (what an activation beyond the fp16-plane range turns into — the tokens decoded after it
        are garbage), and / or a hand-off of the one-launch MLP that gave up (every later hand-off then returns at once: wrong
        tokens AND an optimistic time).  ``generate_codes`` / ``run`` are asynchronous and do NOT call this: whoever drives them
        directly (bench.py, tools/, tests) must call it after synchronising — ``VAURAModel.generate_tokens``, ``forward_cached`` and
        ``logits_all_positions`` do."""
        words = self.state.tolist()                       # ONE transfer: status word + the near-tie counters
        st = int(words[4]) | self._carried_status
        self._carried_status = 0
        if st:
            self.state[4:5].zero_()
        self.last_near_ties = (0, None)
        if st & 4:                                        # informational (csrc/step.hip near-tie detector): never an error
            self.last_near_ties = (int(words[6]), int(words[7]) - 1 if words[7] else None)
            self.near_ties += int(words[6])
            self.state[6:8].zero_()
            st &= ~4
        msgs = []
        if st & 2:
            msgs.append(
                "a consumer of the one-launch MLP gave up waiting for its producers (csrc/mlp_engine.h: its 256 workgroups must all become "
                "resident — is another process, or a second DecoderEngine of this process on another stream, running the same kernels on "
                "this GPU?); every hand-off after the give-up returned without waiting, so the tokens of this call are not valid and its "
                "timing is meaningless.  DecoderEngine(..., one_launch_mlp=False) keeps the two-launch path")
        if st & 1:
            msgs.append(
                "non-finite logits reached the sampler — an activation left even the pre-scaled fp16-plane range, or the checkpoint holds "
                "non-finite weights; weight_dtype='f32' keeps fp32 activations (exact-fp32-MFMA path)")
        if msgs:
            err = L.VauraHipError(f"decode loop (status word {st:#x}): " + "; ALSO: ".join(msgs))
            err.status = st
            raise err

    def revert(self) -> torch.Tensor:
        K, T = self.cfg.num_codebooks, self.T
        L.check(self.lib.vaura_pattern_revert(L.ptr(self.seq), L.ptr(self.codes_i32), self.batch, K, T, self.S, -1,
                                              L.current_stream(self.dev)), "vaura_pattern_revert")
        return self.codes_i32

    @torch.no_grad()
    def generate_codes(self, feats: torch.Tensor, max_new_tokens: int, *, prompt: Optional[torch.Tensor] = None,
                       use_sampling=False, temp=1.0, top_k=0, top_p=0.0, cfg_scale=1.0, noise=None, seed=0,
                       clip_base=0, use_graph=True, tokens_per_frame=7) -> torch.Tensor:
        """The hot loop of generate(): (B, Tv, 768) -> codes (B, K, T) int64 (device)."""
        B, Tv, _ = feats.shape
        cfg_on = cfg_scale > 1.0
        self._fc = None                       # the K/V cache is about to be reused: forward_cached must start over
        with off_null_stream(self.dev) as caller:
            self.prepare(B, max_new_tokens, Tv, cfg_on, tokens_per_frame, block_size=self.cfg.block_size)
            self.set_condition(feats)
            Tp = self.start_sequence(prompt)
            start = Tp + 1  # Pattern.get_first_step_with_timesteps(Tp) for the delayed pattern
            sp = self._sampling(use_sampling, temp, top_k, top_p, cfg_scale, seed, clip_base)
            if noise is not None:
                noise = noise.to(self.dev, torch.float32).contiguous()
                assert noise.shape == (self.S - start, B * self.cfg.num_codebooks, self.cfg.d_codebook), noise.shape
            self.run(start - 1, self.S - start, sp, noise, use_graph)
            out = self.revert().to(torch.int64)
        if caller is not None:
            out.record_stream(caller)
        return out

    @torch.no_grad()
    def generate_codes_checked(self, feats: torch.Tensor, max_new_tokens: int, **kw) -> torch.Tensor:
        """``generate_codes`` + ``check_status`` (ONE host-device synchronisation) with range safety BY CONSTRUCTION: when — and only
        when — the status word reports non-finite logits (an activation left the fp16-plane range somewhere in the loop: every such
        overflow arrives at the sampler as NaN), the whole call is run again on the exact-fp32 twin engine
        (weight_dtype="f32": fp32 activations between kernels, the reference's own arithmetic and range), with the same condition,
        prompt, sampling parameters and noise (an explicit tensor, or Philox keyed by (seed, clip, codebook, step) — not by the engine), so
        the result is what the fp32 path would have produced from the start.  A checkpoint with massive activations therefore decodes
        correctly (at the f32 engine's speed) instead of raising.  Counted in ``range_fallbacks``.  The other status bit — a hand-off of
        the one-launch MLP that gave up because the GPU is shared — switches this engine to the separate launches and re-runs the
        call as well (``handoff_fallbacks``): a slower run, not an error.  Anything else still raises."""
        codes = self.generate_codes(feats, max_new_tokens, **kw)
        try:
            self.check_status()
        except L.VauraHipError as e:
            st = getattr(e, "status", 0)
            if (st & 2) and self.one_launch_mlp:
                # A consumer of the one-launch MLP gave up waiting: its 256 workgroups were not all resident — the GPU is shared
                # (another process / tenant, a CU mask, a second engine on another stream).  Not an error of the model: switch THIS
                # engine to the separate launches for good (no in-launch hand-off, nothing to starve; same bits) and run the call again.
                import warnings
                warnings.warn("vaura_amd: the one-launch MLP's in-launch hand-off timed out (GPU shared?); this engine now uses the "
                              "separate launches (slower by 5-7 %, same results)")
                self.one_launch_mlp = False
                self.handoff_fallbacks += 1
                if self.dec is not None:
                    self.dec.ws_sync = 0
                self._free_graph()
                return self.generate_codes_checked(feats, max_new_tokens, **kw)
            if st != 1 or self.wdtype == "f32" or self._twin_sd is None:       # nothing wider to fall back to
                raise
            if self.range_fallbacks == 0:
                import warnings
                warnings.warn("vaura_amd: an activation left the fp16-plane range (non-finite logits at the sampler); this call is re-run "
                              "on the exact-fp32 twin engine (1.5 x slower, ~2.7 GB more)"
                              + (" — built from the DEQUANTISED matrices this lossy storage holds" if self._twin_lossy else "")
                              + ".  A checkpoint that does this on every call wants plane_shift=4..8 or weight_dtype='f32'")
            self.range_fallbacks += 1
            codes = self._twin().generate_codes(feats, max_new_tokens, **kw)
            self._range_twin.check_status()
            return codes
        if self.near_tie == "rerun" and self.last_near_ties[0] > 0 and self.wdtype != "f32" and self._twin_sd is not None:
            # >= 1 used decision of this call was inside the plane arithmetic's noise: the same call on the exact-fp32 twin, same noise
            self.near_tie_reruns += 1
            codes = self._twin().generate_codes(feats, max_new_tokens, **kw)
            self._range_twin.check_status()
        return codes

    # ------------------------------------------------------------------ the reference host's call pattern, with a cache
    def forward_cached(self, idx: torch.Tensor, feats: torch.Tensor, tokens_per_frame: int = 7) -> torch.Tensor:
        """``Transformer.forward`` as the REFERENCE host calls it — the whole prefix, every step ("no caching is implemented :(",
        models/vaura_model.py:504-506) — served incrementally: when ``idx[..., :n]`` and ``feats`` equal what the previous call
        fed (compared on the device), only positions n.. are run through the decode kernels and their logits are appended to a
        cache; the result is the (Bs, K, L, V) view the contract asks for.  A call that does not extend the cached prefix (new
        clip, other batch, changed condition) starts over.  The host's 228-call loop thus costs 228 decode steps, not 26 106.
        ``self.cached_forward_steps`` counts the decode steps actually run (tests)."""
        Bs, K, Lq = idx.shape
        c = self.cfg
        cap = c.block_size                                   # positions the model supports (scripts/generate.py:221-224)
        if Lq > cap:
            raise L.VauraHipError(f"sequence of {Lq} positions exceeds block_size {cap}")
        if self._forward_on_twin:         # an earlier call overflowed the fp16 planes: this engine's forward() stays on fp32
            return self._range_twin.forward_cached(idx, feats, tokens_per_frame)
        idx = idx.to(self.dev)
        feats = feats.to(self.dev, torch.float32)
        st = getattr(self, "_fc", None)
        n0 = 0
        if (st is not None and st["key"] == (Bs, K, feats.shape[1], tokens_per_frame) and self._shape == st["shape"]
                and st["n"] < Lq and torch.equal(st["feats"], feats) and torch.equal(st["idx"][:, :, :st["n"]], idx[:, :, :st["n"]])):
            n0 = st["n"]
        else:
            self.prepare(Bs, cap - K, feats.shape[1], False, tokens_per_frame, block_size=cap)    # S = cap positions, K/V capacity = cap
            self.set_condition(feats)
            self.seq.zero_()
            self._reset_state()
            st = self._fc = {"key": (Bs, K, feats.shape[1], tokens_per_frame), "shape": self._shape, "feats": feats.clone(),
                             "idx": torch.zeros(Bs, K, cap, dtype=idx.dtype, device=self.dev), "n": 0,
                             "logits": torch.empty(Bs, K, cap, c.d_codebook, dtype=torch.float32, device=self.dev)}
        sp = self._sampling(False, 1.0, 0, 0.0, 1.0, 0, 0)
        stream = L.current_stream(self.dev)
        for p in range(n0, Lq):
            self.seq[:, :, p] = idx[:, :, p].to(torch.int32)       # teacher-forced input of position p (state[0] == p)
            L.check(self.lib.vaura_decode_step(C.byref(self.dec), C.byref(sp), 1, stream), "vaura_decode_step")
            st["logits"][:, :, p] = self.ws_logits.view(Bs, K, -1)
            self.cached_forward_steps = getattr(self, "cached_forward_steps", 0) + 1
        try:
            self.check_status()              # the caller consumes these logits on the host side anyway (one synchronisation)
            st["idx"][:, :, n0:Lq] = idx[:, :, n0:Lq]       # committed only once the status word is clean: a caller that
            st["n"] = Lq                                     # catches the error and goes on must not be served the invalid prefix
        except L.VauraHipError as e:
            self._fc = None
            # range safety for the reference host's call pattern too (see generate_codes_checked): non-finite logits = an activation left
            # the fp16-plane range -> this and every later forward() of the engine runs on the exact-fp32 twin (whole prefix recomputed once)
            if getattr(e, "status", 0) != 1 or self.wdtype == "f32" or self._twin_sd is None:
                raise
            self.range_fallbacks += 1
            self._forward_on_twin = True
            return self._twin().forward_cached(idx, feats, tokens_per_frame)
        return st["logits"][:, :, :Lq].clone()      # a copy: the cache must survive in-place edits by the caller

    # ------------------------------------------------------------------ op-level access (tests)
    def logits_all_positions(self, idx: torch.Tensor, feats: torch.Tensor, tokens_per_frame: int = 7) -> torch.Tensor:
        """Teacher-forced pass: idx (Bs, K, Lq) int64, feats (Bs, Tv, 768) -> logits (Bs, K, Lq, V).
        One decode step per position, heads evaluated every step (API-compat path of
        ``Transformer.forward``; the generate loop never materialises this tensor)."""
        Bs, K, Lq = idx.shape
        self._fc = None
        self.prepare(Bs, Lq, feats.shape[1], False, tokens_per_frame, block_size=self.cfg.block_size)
        self.set_condition(feats)
        self.seq.zero_()
        self.seq[:, :, :Lq] = idx.to(self.dev, torch.int32)
        self._reset_state()
        out = torch.empty(Bs, K, Lq, self.cfg.d_codebook, dtype=torch.float32, device=self.dev)
        sp = self._sampling(False, 1.0, 0, 0.0, 1.0, 0, 0)
        st = L.current_stream(self.dev)
        keep = self.seq.clone()
        for p in range(Lq):
            L.check(self.lib.vaura_decode_step(C.byref(self.dec), C.byref(sp), 1, st), "vaura_decode_step")
            out[:, :, p] = self.ws_logits.view(Bs, K, -1)
            self.seq.copy_(keep)  # the sampler only fills -1 slots, but keep the input pristine anyway
        self.check_status()
        return out


class CodecEngine:
    """DAC decode (codes -> waveform) on the HIP path; weights from a DAC-1.0.0-keyed state dict."""

    def __init__(self, cfg: CodecCfg, sd: Dict[str, torch.Tensor], device="cuda:0", precision: str = "f16pair"):
        """precision: "f16pair" (default; activations/weights as (hi, lo) fp16 pairs on the fp16 MFMA, error
        ~1e-6 RMS), "f32" (exact fp32 MFMA), "f16" (plain fp16 operands, fp32 accumulate — one matrix instruction per product:
        the arithmetic class the reference runs DAC in, vaura_model.py:92; error ~1e-4 of the signal), or "f16pair_w8" (BASELINE configs[4]: conv weights quantised to fp8 e4m3 with
        a power-of-two scale per output channel — a different model, ``quant.fp8_effective_codec_state_dict`` says which;
        such weights are exact in one fp16 plane, so a product costs two MFMAs instead of three), or "mx8" (configs[4] on the
        fp8 matrix instruction: those fp8 weights AND e4m3 activations with one power-of-two scale per 32 channels of a row,
        ``quant.mx8_effective_activation``; one block-scaled MFMA per 128 products — tolerance reported, not 1e-4)."""
        _require_cuda(device)
        self.cfg, self.dev, self.lib = cfg, torch.device(device), L.lib()
        self._keep = []
        self.pairs = {"f32": 0, "f16pair": 1, "f16pair_w8": 2, "mx8": 3, "f16": 4}[precision]
        if self.pairs in (2, 3):
            from .quant import fp8_effective_codec_state_dict
            sd = fp8_effective_codec_state_dict(sd)
        c = L.Codec()
        c.precision = self.pairs
        c.n_codebooks, c.codebook_size, c.codebook_dim, c.latent_dim = (cfg.n_codebooks, cfg.codebook_size,
                                                                        cfg.codebook_dim, cfg.latent_dim)
        nb = len(cfg.decoder_rates)
        assert nb <= 4 and len(cfg.dilations) == 3
        c.n_blocks, c.n_units = nb, 3
        for i, r in enumerate(cfg.decoder_rates):
            c.rates[i] = r
        K = cfg.n_codebooks
        f = lambda p: fold_weight_norm(sd[p + "weight_g"].float(), sd[p + "weight_v"].float())
        c.codebooks = L.ptr(self._dev(torch.stack([sd[f"quantizer.quantizers.{k}.codebook.weight"].float() for k in range(K)])))
        c.out_proj_w = L.ptr(self._dev(torch.stack([f(f"quantizer.quantizers.{k}.out_proj.")[:, :, 0] for k in range(K)])))
        c.out_proj_b = L.ptr(self._dev(torch.stack([sd[f"quantizer.quantizers.{k}.out_proj.bias"].float() for k in range(K)])))
        self._conv(c.conv_in, sd, "decoder.model.0.", 1, 1, mx8=False)
        for b, r in enumerate(cfg.decoder_rates):
            p = f"decoder.model.{b + 1}.block."
            c.alpha_up[b] = L.ptr(self._dev(sd[p + "0.alpha"].reshape(-1)))
            self._conv(c.up[b], sd, p + "1.", 1, r)
            for u, dil in enumerate(cfg.dilations):
                q = p + f"{u + 2}.block."
                c.alpha_res[b][u][0] = L.ptr(self._dev(sd[q + "0.alpha"].reshape(-1)))
                self._conv(c.res[b][u][0], sd, q + "1.", dil, 1)
                c.alpha_res[b][u][1] = L.ptr(self._dev(sd[q + "2.alpha"].reshape(-1)))
                self._conv(c.res[b][u][1], sd, q + "3.", 1, 1)
        n = nb + 1
        c.alpha_out = L.ptr(self._dev(sd[f"decoder.model.{n}.alpha"].reshape(-1)))
        self._conv(c.conv_out, sd, f"decoder.model.{n + 1}.", 1, 1)
        self.c = c
        self._ws_key = None

    def _dev(self, t):
        d = t.detach().to(self.dev, torch.float32).contiguous()
        self._keep.append(d)
        return d

    def _conv(self, cv: L.Conv, sd, prefix: str, dilation: int, stride: int, mx8: bool = True):
        w = fold_weight_norm(sd[prefix + "weight_g"].float(), sd[prefix + "weight_v"].float())
        self._pack_conv(cv, w, sd[prefix + "bias"], dilation, stride, mx8)

    def _pack_conv(self, cv: L.Conv, w: torch.Tensor, bias: torch.Tensor, dilation: int, stride: int, mx8: bool = True):
        """Folded Conv1d (Cout, Cin, k) / ConvTranspose1d (Cin, Cout, 2r) weight -> the layout of self.pairs (= vaura_codec.precision)."""
        if stride > 1:   # ConvTranspose1d weight (Cin, Cout, 2r) -> [phase][tap][Cout][Cin]
            cin, cout, k = w.shape
            assert k == 2 * stride
            wl = w.permute(2, 1, 0).reshape(2, stride, cout, cin).permute(1, 0, 2, 3)
            taps = 2
        else:            # Conv1d weight (Cout, Cin, k) -> [tap][Cout][Cin]
            cout, cin, k = w.shape
            wl = w.permute(2, 0, 1)
            taps = k
        if self.pairs == 3 and mx8 and cout > 1:   # packed e4m3 stream + per-output-channel scales (quant.mx8_pack_conv_weight)
            from .quant import mx8_pack_conv_weight
            wp = wl.reshape(stride, 2, cout, cin) if stride > 1 else wl.reshape(1, taps, cout, cin)
            stream, scale = mx8_pack_conv_weight(wp.contiguous().to(self.dev))
            self._keep.append(stream)
            cv.w = L.ptr(stream)
            cv.wscale = L.ptr(self._dev(scale))
        elif self.pairs and cout > 1:   # (hi, lo) fp16 pair layout [.., Cout][Cin/8][plane][8]
            wl = wl.contiguous()
            hi = wl.half()
            lo = (wl - hi.float()).half()
            pr = torch.stack([hi.reshape(*wl.shape[:-1], cin // 8, 8), lo.reshape(*wl.shape[:-1], cin // 8, 8)], dim=-2)
            keep = pr.contiguous().to(self.dev)
            self._keep.append(keep)
            cv.w = L.ptr(keep)
        else:
            cv.w = L.ptr(self._dev(wl))
        cv.bias = L.ptr(self._dev(bias))
        cv.cin, cv.cout, cv.taps, cv.dilation, cv.stride = cin, cout, taps, dilation, stride

    @torch.no_grad()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (B, K, T) integer tensor on device -> wav (B, 1, T*hop) fp32."""
        B, K, T = codes.shape
        with off_null_stream(self.dev) as caller:
            ci = codes.to(self.dev, torch.int32).contiguous()
            need = self.lib.vaura_dac_workspace_elems(C.byref(self.c), B, T)
            if self._ws_key is None or self._ws_key < need:
                self._ws = [torch.empty(need, dtype=torch.float32, device=self.dev) for _ in range(4)]
                for i in range(4):
                    self.c.ws[i] = L.ptr(self._ws[i])
                self.c.ws_elems = need
                self._ws_key = need
            wav = torch.empty(B, 1, T * self.cfg.hop, dtype=torch.float32, device=self.dev)
            L.check(self.lib.vaura_dac_decode(C.byref(self.c), L.ptr(ci), B, T, L.ptr(wav), L.current_stream(self.dev)),
                    "vaura_dac_decode")
            self._codes_keepalive = ci
        if caller is not None:
            wav.record_stream(caller)
        return wav


class CodecConvOp:
    """ONE decoder convolution through ``vaura_dac_conv`` (op-level parity tests): folded Conv1d (Cout, Cin, k) or
    ConvTranspose1d (Cin, Cout, 2r; stride r) weight, in the arithmetic of a codec ``precision``.  For "f16pair_w8" / "mx8" the
    caller passes weights that are already fp8-representable (``quant.fp8_effective_weight``)."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, dilation: int = 1, stride: int = 1, precision: str = "f16pair",
                 device="cuda:0", mx8_weights: bool = True):
        _require_cuda(device)
        self.dev, self.lib, self._keep = torch.device(device), L.lib(), []
        self.pairs = {"f32": 0, "f16pair": 1, "f16pair_w8": 2, "mx8": 3, "f16": 4}[precision]
        self.cv = L.Conv()
        CodecEngine._pack_conv(self, self.cv, weight.float(), bias.float(), dilation, stride, mx8_weights)
        self.stride = stride

    _dev = CodecEngine._dev

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, L, Cin) fp32, channels last, already activated -> (B, L * stride, Cout) fp32 = conv(x) + bias."""
        B, Lin, cin = x.shape
        assert cin == self.cv.cin
        with off_null_stream(self.dev) as caller:
            xi = x.to(self.dev, torch.float32).contiguous()
            out = torch.empty(B, Lin * self.stride, self.cv.cout, dtype=torch.float32, device=self.dev)
            scratch = torch.empty(B * Lin * cin + 64, dtype=torch.float32, device=self.dev)
            L.check(self.lib.vaura_dac_conv(C.byref(self.cv), self.pairs, L.ptr(xi), L.ptr(out), L.ptr(scratch), B, Lin,
                                            L.current_stream(self.dev)), "vaura_dac_conv")
            self._keepalive = (xi, scratch)
        if caller is not None:
            out.record_stream(caller)
        return out


def snake(x: torch.Tensor, alpha: torch.Tensor, device="cuda:0") -> torch.Tensor:
    """The codec's activation through ``vaura_snake`` (op-level parity tests): x (..., C) fp32 channels-last, alpha (C)."""
    _require_cuda(device)
    dev = torch.device(device)
    with off_null_stream(dev) as caller:
        xi = x.to(dev, torch.float32).contiguous()
        al = alpha.to(dev, torch.float32).contiguous()
        y = torch.empty_like(xi)
        L.check(L.lib().vaura_snake(L.ptr(xi), L.ptr(al), L.ptr(y), xi.numel() // xi.shape[-1], xi.shape[-1], L.current_stream(dev)),
                "vaura_snake")
    if caller is not None:
        y.record_stream(caller)
    return y


class CodecEncoderEngine:
    """DAC encode (waveform -> codes) on the HIP path; weights from a DAC-1.0.0-keyed state dict
    (``encoder.block.*``, ``quantizer.quantizers.N.{in_proj,codebook,out_proj}``).  Convolutions run on (hi, lo) fp16
    pairs like the decoder's default precision."""

    def __init__(self, cfg: CodecCfg, sd: Dict[str, torch.Tensor], device="cuda:0"):
        _require_cuda(device)
        self.cfg, self.dev, self.lib = cfg, torch.device(device), L.lib()
        self._keep = []
        c = L.CodecEncoder()
        c.n_codebooks, c.codebook_size, c.codebook_dim, c.latent_dim = (cfg.n_codebooks, cfg.codebook_size,
                                                                        cfg.codebook_dim, cfg.latent_dim)
        nb = len(cfg.encoder_rates)
        assert nb <= 4 and len(cfg.dilations) == 3
        c.n_blocks, c.n_units, c.enc_dim = nb, 3, cfg.encoder_dim
        for i, r in enumerate(cfg.encoder_rates):
            c.rates[i] = r
        K = cfg.n_codebooks
        f = lambda p: fold_weight_norm(sd[p + "weight_g"].float(), sd[p + "weight_v"].float())
        q = "quantizer.quantizers."
        c.in_proj_w = L.ptr(self._dev(torch.stack([f(f"{q}{k}.in_proj.")[:, :, 0] for k in range(K)])))
        c.in_proj_b = L.ptr(self._dev(torch.stack([sd[f"{q}{k}.in_proj.bias"].float() for k in range(K)])))
        c.codebooks = L.ptr(self._dev(torch.stack([sd[f"{q}{k}.codebook.weight"].float() for k in range(K)])))
        c.out_proj_w = L.ptr(self._dev(torch.stack([f(f"{q}{k}.out_proj.")[:, :, 0] for k in range(K)])))
        c.out_proj_b = L.ptr(self._dev(torch.stack([sd[f"{q}{k}.out_proj.bias"].float() for k in range(K)])))
        w0 = f("encoder.block.0.")                                   # (C, 1, 7) -> [tap][C]
        c.conv_in_w = L.ptr(self._dev(w0[:, 0, :].t()))
        c.conv_in_b = L.ptr(self._dev(sd["encoder.block.0.bias"]))
        d = cfg.encoder_dim
        for b, r in enumerate(cfg.encoder_rates):
            p = f"encoder.block.{b + 1}.block."
            for u, dil in enumerate(cfg.dilations):
                qq = p + f"{u}.block."
                c.alpha_res[b][u][0] = L.ptr(self._dev(sd[qq + "0.alpha"].reshape(-1)))
                self._conv(c.res[b][u][0], f(qq + "1.").permute(2, 0, 1), sd[qq + "1.bias"], dil)
                c.alpha_res[b][u][1] = L.ptr(self._dev(sd[qq + "2.alpha"].reshape(-1)))
                self._conv(c.res[b][u][1], f(qq + "3.").permute(2, 0, 1), sd[qq + "3.bias"], 1)
            c.alpha_down[b] = L.ptr(self._dev(sd[p + "3.alpha"].reshape(-1)))
            self._conv(c.down[b], self.strided_as_three_taps(f(p + "4."), r), sd[p + "4.bias"], 1)
            d *= 2
        n = nb + 1
        c.alpha_out = L.ptr(self._dev(sd[f"encoder.block.{n}.alpha"].reshape(-1)))
        self._conv(c.conv_out, f(f"encoder.block.{n + 1}.").permute(2, 0, 1), sd[f"encoder.block.{n + 1}.bias"], 1)
        self.c = c
        self._ws_key = None

    @staticmethod
    def strided_as_three_taps(w: torch.Tensor, r: int) -> torch.Tensor:
        """Conv1d weight (2C, C, 2r) with stride r, pad r/2  ->  [3][2C][r*C]: the same sum written over rows of r*C
        channels (input (L, C) read as (L/r, r*C)): w'[tau+1][co][q*C + ci] = w[co][ci][tau*r + q + r/2]."""
        cout, cin, k = w.shape
        assert k == 2 * r and r % 2 == 0
        out = torch.zeros(3, cout, r * cin, dtype=w.dtype)
        for tau in (-1, 0, 1):
            for qi in range(r):
                t = tau * r + qi + r // 2
                if 0 <= t < k:
                    out[tau + 1, :, qi * cin:(qi + 1) * cin] = w[:, :, t]
        return out

    def _dev(self, t):
        d = t.detach().to(self.dev, torch.float32).contiguous()
        self._keep.append(d)
        return d

    def _conv(self, cv: L.Conv, wl: torch.Tensor, bias: torch.Tensor, dilation: int):
        """wl: [taps][Cout][Cin] fp32 -> (hi, lo) fp16 pair layout [taps][Cout][Cin/8][plane][8]."""
        wl = wl.contiguous().float()
        taps, cout, cin = wl.shape
        hi = wl.half()
        lo = (wl - hi.float()).half()
        pr = torch.stack([hi.reshape(taps, cout, cin // 8, 8), lo.reshape(taps, cout, cin // 8, 8)], dim=-2)
        keep = pr.contiguous().to(self.dev)
        self._keep.append(keep)
        cv.w = L.ptr(keep)
        cv.bias = L.ptr(self._dev(bias))
        cv.cin, cv.cout, cv.taps, cv.dilation, cv.stride = cin, cout, taps, dilation, 1

    @torch.no_grad()
    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        """wav (B, 1, N) / (B, N) / (N) on the device -> codes (B, K, ceil(N / hop)) int64.  The zero padding of
        ``DAC.preprocess`` is applied here."""
        if wav.dim() == 1:
            wav = wav[None]
        if wav.dim() == 3:
            wav = wav[:, 0]
        hop = int(math.prod(self.cfg.encoder_rates))
        B, N = wav.shape
        T = (N + hop - 1) // hop
        with off_null_stream(self.dev) as caller:
            x = torch.zeros(B, T * hop, dtype=torch.float32, device=self.dev)
            x[:, :N] = wav.to(self.dev, torch.float32)
            need = self.lib.vaura_dac_encode_workspace_elems(C.byref(self.c), B, T * hop)
            if self._ws_key is None or self._ws_key < need:
                self._ws = [torch.empty(need, dtype=torch.float32, device=self.dev) for _ in range(4)]
                for i in range(4):
                    self.c.ws[i] = L.ptr(self._ws[i])
                self.c.ws_elems = need
                self._ws_key = need
            codes = torch.empty(B, self.cfg.n_codebooks, T, dtype=torch.int32, device=self.dev)
            L.check(self.lib.vaura_dac_encode(C.byref(self.c), L.ptr(x), B, T * hop, L.ptr(codes), L.current_stream(self.dev)),
                    "vaura_dac_encode")
            out = codes.to(torch.int64)
        if caller is not None:
            out.record_stream(caller)
        return out


class AvclipEngine:
    """Segment-AVCLIP visual features on the HIP path: weights from a state dict with the reference
    MotionFormer's key names (``synth.avclip_state_dict`` lists them), frames (B, S, 3, 16, 224, 224) -> (B, S, 8, 768)."""

    def __init__(self, cfg, sd: Dict[str, torch.Tensor], device="cuda:0"):
        _require_cuda(device)
        self.cfg, self.dev, self.lib = cfg, torch.device(device), L.lib()
        self._keep = []
        D, Hd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
        v = L.Vit()
        v.depth, v.dim, v.heads, v.hidden = cfg.depth, D, cfg.num_heads, Hd
        v.n_patches, v.n_frames = cfg.n, cfg.t
        v.in_chans, v.frames, v.img, v.patch, v.patch_t = cfg.in_chans, cfg.frames, cfg.img, cfg.patch, cfg.patch_t
        v.patch_k = cfg.in_chans * cfg.patch_t * cfg.patch * cfg.patch
        v.eps = 1e-6
        f, p = self._f32, self._pair
        v.pe_w, v.pe_b = p(sd["patch_embed_3d.proj.weight"].reshape(D, -1)), f(sd["patch_embed_3d.proj.bias"])
        v.cls_token, v.pos_embed, v.temp_embed = f(sd["cls_token"]), f(sd["pos_embed"]), f(sd["temp_embed"])
        blocks = (L.VitBlock * cfg.depth)()
        for i in range(cfg.depth):
            b, q = blocks[i], f"blocks.{i}."
            b.ln1_w, b.ln1_b = f(sd[q + "norm1.weight"]), f(sd[q + "norm1.bias"])
            b.ln2_w, b.ln2_b = f(sd[q + "norm2.weight"]), f(sd[q + "norm2.bias"])
            b.ln3_w, b.ln3_b = f(sd[q + "norm3.weight"]), f(sd[q + "norm3.bias"])
            for att, name in ((b.space, "attn."), (b.time, "timeattn.")):
                att.qkv_w, att.qkv_b = p(sd[q + name + "qkv.weight"]), f(sd[q + name + "qkv.bias"])
                att.proj_w, att.proj_b = p(sd[q + name + "proj.weight"]), f(sd[q + name + "proj.bias"])
            b.fc1_w, b.fc1_b = p(sd[q + "mlp.fc1.weight"]), f(sd[q + "mlp.fc1.bias"])
            b.fc2_w, b.fc2_b = p(sd[q + "mlp.fc2.weight"]), f(sd[q + "mlp.fc2.bias"])
        self._blocks = blocks
        v.blocks_host = C.cast(blocks, C.POINTER(L.VitBlock))
        v.norm_w, v.norm_b = f(sd["norm.weight"]), f(sd["norm.bias"])
        a = "spatial_attn_agg."
        v.agg_cls = f(sd[a + "cls_token"])
        v.agg_ln1_w, v.agg_ln1_b = f(sd[a + "norm1.weight"]), f(sd[a + "norm1.bias"])
        v.agg_ln2_w, v.agg_ln2_b = f(sd[a + "norm2.weight"]), f(sd[a + "norm2.bias"])
        v.agg_in_w, v.agg_in_b = p(sd[a + "self_attn.in_proj_weight"]), f(sd[a + "self_attn.in_proj_bias"])
        v.agg_out_w, v.agg_out_b = p(sd[a + "self_attn.out_proj.weight"]), f(sd[a + "self_attn.out_proj.bias"])
        v.agg_l1_w, v.agg_l1_b = p(sd[a + "linear1.weight"]), f(sd[a + "linear1.bias"])
        v.agg_l2_w, v.agg_l2_b = p(sd[a + "linear2.weight"]), f(sd[a + "linear2.bias"])
        self.v = v
        self._ws_segs = 0

    def _f32(self, t) -> int:
        d = t.detach().to(self.dev, torch.float32).contiguous()
        self._keep.append(d)
        return L.ptr(d)

    def _pair(self, w) -> int:
        """(Cout, Cin) fp32 -> (hi, lo) fp16 pair layout [Cout][Cin/8][plane][8] (the codec's weight layout, one tap)."""
        w = w.detach().to(self.dev, torch.float32).contiguous()
        cout, cin = w.shape
        hi = w.half()
        lo = (w - hi.float()).half()
        pr = torch.stack([hi.reshape(cout, cin // 8, 8), lo.reshape(cout, cin // 8, 8)], dim=-2).contiguous()
        self._keep.append(pr)
        return L.ptr(pr)

    # segments per pass: bounds the workspaces (a pass of 32 segments = 8 clips holds ~1.7 GB)
    MAX_SEGMENTS = 32

    def _workspaces(self, n_seg: int):
        if n_seg <= self._ws_segs:
            return
        names = ["ws_x", "ws_qkv", "ws_a", "ws_h", "ws_p", "ws_z", "ws_s"]
        self._ws = []
        for i, nm in enumerate(names):
            nbytes = self.lib.vaura_avclip_workspace_bytes(C.byref(self.v), n_seg, i)
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            self._ws.append(t)
            setattr(self.v, nm, L.ptr(t))
        self._ws_segs = n_seg

    @torch.no_grad()
    def forward(self, frames: torch.Tensor) -> torch.Tensor:
        """frames (B, S, 3, 16, 224, 224) on the device -> features (B, S, 8, 768) fp32."""
        c = self.cfg
        B, S = frames.shape[:2]
        if tuple(frames.shape[2:]) != (c.in_chans, c.frames, c.img, c.img):
            raise L.VauraHipError(f"AvclipEngine: segments must be ({c.in_chans}, {c.frames}, {c.img}, {c.img}), got {tuple(frames.shape[2:])}")
        with off_null_stream(self.dev) as caller:
            x = frames.to(self.dev, torch.float32).reshape(B * S, *frames.shape[2:]).contiguous()
            out = torch.empty(B * S, c.t, c.embed_dim, dtype=torch.float32, device=self.dev)
            step = min(self.MAX_SEGMENTS, B * S)
            self._workspaces(step)
            for s0 in range(0, B * S, step):
                n = min(step, B * S - s0)
                L.check(self.lib.vaura_avclip_forward(C.byref(self.v), L.ptr(x[s0:]), n, L.ptr(out[s0:]), L.current_stream(self.dev)),
                        "vaura_avclip_forward")
            self._frames_keepalive = x
        if caller is not None:
            out.record_stream(caller)
        return out.view(B, S, c.t, c.embed_dim)
