"""Host-side description of the fp8 weight format (``VAURA_W_FP8``, include/vaura_hip.h).

The device quantiser lives in libvaura_hip.so (``vaura_pack_weight(..., VAURA_W_FP8)``); these few torch
lines state the same rule so that a caller (and the parity tests) can ask "which fp32 matrix does the fp8
model actually multiply by?".  The reference has no fp8 path (BASELINE.json configs[4] is ours), so the
contract is: the fp8 engine generates exactly the tokens the one-fp16-plane ("h1") engine generates for the
checkpoint whose four per-layer matrices are replaced by ``fp8_effective_weight(W)``.

Format: OCP e4m3 (``torch.float8_e4m3fn``), one scale per output row, the smallest power of two with
``max|W[n, :]| <= 448 * scale[n]``.  e4m3 widens to fp16 exactly (3 significand bits, exponents inside fp16's
range) and the power-of-two scale is applied to the fp32 sum, so the kernel computes exactly what the
one-plane path computes on the dequantised matrix.
"""
from __future__ import annotations

import torch

FP8_MAX = 448.0


def fp8_row_scales(w: torch.Tensor) -> torch.Tensor:
    """(N, K) fp32 -> (N) fp32 power-of-two scales."""
    amax = w.detach().float().abs().amax(dim=1)
    m, e = torch.frexp(amax)                       # amax = m * 2^e, m in [0.5, 1);  448 = 0.875 * 2^9
    exp = torch.where(m <= 0.875, e - 9, e - 8)
    scale = torch.ldexp(torch.ones_like(amax), exp)
    return torch.where(amax > 0, scale, torch.ones_like(amax))


def fp8_effective_weight(w: torch.Tensor) -> torch.Tensor:
    """The fp32 matrix the fp8 engine multiplies by: round-to-nearest-even e4m3 of W/scale, times scale."""
    w = w.detach().float()
    s = fp8_row_scales(w)[:, None]
    return (w / s).to(torch.float8_e4m3fn).float() * s


FP8_LAYER_KEYS = ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w2.weight",
                  "feed_forward.w3.weight")


def fp8_effective_state_dict(sd: dict) -> dict:
    """Sampler state dict with every matrix the fp8 engine stores in fp8 replaced by its dequantised value.
    w1/w3 rows are scaled independently per row, so quantising them separately equals quantising the
    interleaved (w1, w3) matrix the engine streams."""
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith("layers.") and k.endswith(FP8_LAYER_KEYS):
            out[k] = fp8_effective_weight(v)
    return out


def fp8_effective_codec_state_dict(sd: dict) -> dict:
    """DAC state dict whose decoder convolutions (every ``decoder.model.*`` conv except the final C -> 1 one) are replaced
    by their fp8-dequantised values: e4m3 with one power-of-two scale per OUTPUT channel over (input channels x taps).
    Weight-norm parametrisation is kept foldable: ``weight_v`` holds the dequantised weight and ``weight_g`` its own norm,
    so ``g * v / |v|`` returns it unchanged."""
    out = dict(sd)
    for k in list(sd):
        if not (k.startswith("decoder.model.") and k.endswith("weight_v")):
            continue
        p = k[: -len("weight_v")]
        v, g = sd[k].float(), sd[p + "weight_g"].float()
        w = v * (g / v.norm(2, dim=tuple(range(1, v.dim())), keepdim=True))      # folded weight
        transposed = g.shape[0] == w.shape[0] and w.dim() == 3 and (".block.1." in p and p.count(".block.") == 1)
        if transposed:       # ConvTranspose1d (Cin, Cout, k): output channels on dim 1
            cin, cout, kk = w.shape
            we = fp8_effective_weight(w.permute(1, 0, 2).reshape(cout, cin * kk)).reshape(cout, cin, kk).permute(1, 0, 2).contiguous()
        else:                # Conv1d (Cout, Cin, k)
            cout, cin, kk = w.shape
            if cout == 1:
                continue     # the last conv (C -> 1) stays as is: it runs in fp32
            we = fp8_effective_weight(w.reshape(cout, cin * kk)).reshape(cout, cin, kk)
        out[k] = we
        out[p + "weight_g"] = we.norm(2, dim=tuple(range(1, we.dim())), keepdim=True)
    return out


# ------------------------------------------------------------------------------------------------------------------
# Block-scaled fp8 ("mx8") codec convolutions — codec precision 3 (include/vaura_hip.h, csrc/dac.hip::conv_mx8_kernel)

def mx8_effective_activation(x: torch.Tensor) -> torch.Tensor:
    """What an mx8 consumer sees of an activation tensor (..., C), C % 32 == 0: every block of 32 channels of a row is
    rounded to e4m3 under its own power-of-two scale (the same smallest-power-of-two rule as the weights)."""
    shp = x.shape
    blk = x.detach().float().reshape(-1, 32)
    amax = blk.abs().amax(dim=1)
    m, e = torch.frexp(amax)
    exp = torch.where(m <= 0.875, e - 9, e - 8).clamp(min=-126, max=126)     # the kernel clamps the E8M0 byte to [1, 253]
    s = torch.ldexp(torch.ones_like(amax), exp)[:, None]
    return ((blk / s).to(torch.float8_e4m3fn).float() * s).reshape(shp)


def mx8_pack_conv_weight(wl: torch.Tensor):
    """(P, NT, Cout, Cin) fp32 taps of one conv (P output phases; 1 for a plain conv) -> (uint8 stream
    [P][steps][Cout][4][32], (Cout) fp32 power-of-two scales) in the k order conv_mx8_kernel walks: input channels in
    super-chunks of 128 (the last holds nch = Cin%128/32 blocks if that is not zero), k-blocks kb = tap*nch + ch, four
    per step; inside a step the 32 bytes of lane group G are [half G&1 of block G>>1 | half G&1 of block 2 + (G>>1)] —
    the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (bytes 0..15 of group G: k = 16G.., bytes 16..31: k = 64+16G..;
    measured by tools/microbench/mfma_mx8_probe.hip).  The scale is per output channel over (phases x taps x Cin) — fp8_row_scales' rule — so a weight that
    fp8_effective_weight already rounded is represented exactly."""
    P, NT, cout, cin = wl.shape
    assert cin % 32 == 0
    flat = wl.permute(2, 0, 1, 3).reshape(cout, -1)
    scale = fp8_row_scales(flat)
    q = (wl / scale[None, None, :, None]).to(torch.float8_e4m3fn).view(torch.uint8)      # (P, NT, Cout, Cin)
    nsc = (cin + 127) // 128
    steps = []
    for sc in range(nsc):
        nch = min(4, (cin - 128 * sc) // 32)
        for st in range((nch * NT + 3) // 4):
            blk = torch.zeros(P, cout, 4, 32, dtype=torch.uint8, device=wl.device)
            for kbl in range(4):
                kb = 4 * st + kbl
                if kb < nch * NT:
                    t, ch = divmod(kb, nch)
                    c0 = 128 * sc + 32 * ch
                    for h in range(2):           # 16-channel half h of the block -> lane group 2*(kbl%2)+h, quad kbl//2
                        blk[:, :, 2 * (kbl % 2) + h, 16 * (kbl // 2):16 * (kbl // 2) + 16] = q[:, t, :, c0 + 16 * h:c0 + 16 * h + 16]
            steps.append(blk)
    return torch.stack(steps, dim=1).contiguous(), scale
