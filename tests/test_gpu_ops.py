"""HIP kernels vs the CPU oracle, op by op, through the C ABI (run on the GPU box: -m gpu)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from vaura_amd import _lib as L
from vaura_amd import ops, synth

DEV = "cuda:0"


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_library_loaded_and_versioned():
    assert b"gfx950" in L.lib().vaura_version()


@pytest.mark.parametrize("rows,C", [(1, 512), (8, 1536), (16, 1536), (37, 768), (64, 4096)])
def test_pack_rows_round_trip(rows, C):
    g = torch.Generator().manual_seed(rows * 7 + C)
    x = torch.randn(rows, C, generator=g)
    xp = ops.pack_rows(x.to(DEV))
    assert xp.numel() == ((rows + 15) // 16 * 16) * C
    # layout contract of include/vaura_hip.h
    r, c = rows - 1, C - 3
    idx = (((r >> 4) * (C >> 2) + (c >> 2)) * 16 + (r & 15)) * 4 + (c & 3)
    assert float(xp[idx]) == float(x[r, c])
    assert torch.equal(ops.unpack_rows(xp, rows, C).cpu(), x)


def _ref_gemv(w, x, gain, eps, epi, res):
    """fp32 oracle of the fused op: F.linear(rmsnorm(x)) [+res | silu*mul | gelu]."""
    if gain is not None:
        x = (x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)) * gain
    return x, w


GEMV_CASES = [
    # K, N, epilogue, norm, rows
    (1536, 4608, L.EPI_STORE, True, 16),
    (1536, 4608, L.EPI_STORE, True, 3),
    (1536, 1536, L.EPI_RESID, False, 16),
    (1536, 8192, L.EPI_SWIGLU, True, 8),
    (4096, 1536, L.EPI_RESID, False, 16),
    (1536, 9216, L.EPI_LOGITS, True, 16),
    (1536, 9216, L.EPI_LOGITS, True, 5),
    (768, 512, L.EPI_GELU, False, 40),
    (512, 512, L.EPI_STORE, False, 40),
    (1536, 4608, L.EPI_STORE, True, 32),
]


@pytest.mark.parametrize("wdtype", [L.W_F32, L.W_BF16])
@pytest.mark.parametrize("K,N,epi,norm,rows", GEMV_CASES)
def test_gemv_variants(K, N, epi, norm, rows, wdtype):
    g = torch.Generator().manual_seed(K + N + epi + rows)
    w = torch.randn(N, K, generator=g) * 0.02
    if wdtype == L.W_BF16:
        w = synth.to_bf16_exact(w)  # same real numbers on both sides
    x = torch.randn(rows, K, generator=g)
    gain = torch.rand(K, generator=g) + 0.5 if norm else None
    eps = 1e-5
    xn = x if gain is None else (x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)) * gain
    n_out = N // 2 if epi == L.EPI_SWIGLU else N
    res = torch.randn(rows, n_out, generator=g) if epi == L.EPI_RESID else None
    if epi == L.EPI_SWIGLU:
        F_ = N // 2
        w1, w3 = w[:F_], w[F_:]
        ref = F.silu(F.linear(xn, w1)) * F.linear(xn, w3)
        w_dev = torch.stack([w1.view(F_ // 16, 16, K), w3.view(F_ // 16, 16, K)], dim=1).reshape(N, K)
    else:
        ref = F.linear(xn, w)
        w_dev = w
        if epi == L.EPI_RESID:
            ref = res + ref
        if epi == L.EPI_GELU:
            ref = F.gelu(ref, approximate="tanh")
    wp = ops.pack_weight(w_dev.to(DEV), wdtype)
    xp = ops.pack_rows(x.to(DEV))
    resp = ops.pack_rows(res.to(DEV)) if res is not None else None
    out = ops.gemv(wp, wdtype, xp, rows, N, K, epi, gain.to(DEV) if norm else None, resp, eps)
    got = out.cpu() if epi == L.EPI_LOGITS else ops.unpack_rows(out, rows, n_out).cpu()
    assert got.shape == ref.shape
    # fp32 products summed in a different order than the CPU BLAS: ~sqrt(K)*2^-24 relative
    assert rel_err(got, ref) < 2e-5, rel_err(got, ref)


SPLIT_GEMV_CASES = [
    # K, N, epilogue, norm, rows
    (1536, 4608, L.EPI_STORE, True, 16),
    (1536, 1536, L.EPI_RESID, False, 5),
    (1536, 8192, L.EPI_SWIGLU, True, 16),
    (4096, 1536, L.EPI_RESID, False, 16),
    (1536, 9216, L.EPI_LOGITS, True, 7),
    (1536, 4608, L.EPI_STORE, True, 40),
    # >= 16 row blocks: the prefill GEMM tiling; 17 / 19 blocks = ragged M tile
    (1536, 4608, L.EPI_STORE, True, 300),
    (1536, 1536, L.EPI_RESID, False, 256),
    (1536, 8192, L.EPI_SWIGLU, True, 260),
    (4096, 1536, L.EPI_RESID, False, 300),
    (1536, 9216, L.EPI_LOGITS, True, 270),
    (1536, 4608, L.EPI_STORE, True, 100),     # 7 row blocks: still the GEMV loop
    # enough 128-row x 256-column workgroups to fill the chip: the 128-row instances of the LDS-DMA GEMM (ragged last panel of
    # the XCD tile order: 17 and 21 row tiles)
    (1536, 8192, L.EPI_SWIGLU, True, 2100),
    (1536, 9216, L.EPI_LOGITS, True, 2100),
    (1536, 4608, L.EPI_STORE, True, 2656),
    (1536, 8192, L.EPI_SWIGLU, True, 2656),   # cut into two launches: 512 x 128 rows, then the last 38 row blocks in 96-row workgroups
]


@pytest.mark.parametrize("wdtype", [L.W_H1, L.W_H2, L.W_FP8, L.W_FP8H])
@pytest.mark.parametrize("K,N,epi,norm,rows", SPLIT_GEMV_CASES)
def test_split_row_gemv_variants(K, N, epi, norm, rows, wdtype):
    """vaura_gemv_pair: activations as (hi, lo) fp16 planes (22 significand bits), weights as one fp16 plane (a
    bf16-representable matrix: held exactly), two planes (an fp32 matrix: 22 bits) or fp8, fused RMSNorm via partial sums of
    squares, optional split / sum-of-squares outputs for the next kernel.  Checked against the fp64 product on the matrix the
    storage HOLDS (engine.h_effective_weight / quant.fp8_effective_weight) and, for two planes, also on the fp32 matrix itself:
    the tolerance there is what 22-bit operands cost (2^-22 per product, averaged over K).  VAURA_W_FP8H (round 6): the fp8 matrix
    against the HI activation plane only — the reference is the fp64 product of the fp16-ROUNDED activation (x * gain) with the
    dequantised matrix, at the same 3e-6 (the decode instances: fewer than 16 row blocks; a prompt-sized GEMM of that storage
    multiplies both planes)."""
    from vaura_amd import quant
    from vaura_amd.engine import h_effective_weight
    g = torch.Generator().manual_seed(K + N + epi + rows + 1)
    w = torch.randn(N, K, generator=g) * 0.02
    x = torch.randn(rows, K, generator=g)
    gain = torch.rand(K, generator=g) + 0.5 if norm else None
    gain_out = torch.rand(N // 2 if epi == L.EPI_SWIGLU else N, generator=g) + 0.5
    eps = 1e-5
    n_out = N // 2 if epi == L.EPI_SWIGLU else N
    res = torch.randn(rows, n_out, generator=g) if epi == L.EPI_RESID else None
    if epi == L.EPI_SWIGLU:
        F_ = N // 2
        w = torch.stack([w[:F_].view(F_ // 16, 16, K), w[F_:].view(F_ // 16, 16, K)], dim=1).reshape(N, K)
    if wdtype == L.W_H1:
        w = synth.to_bf16_exact(w)
    w_eff = quant.fp8_effective_weight(w) if wdtype in (L.W_FP8, L.W_FP8H) else h_effective_weight(w, 1 if wdtype == L.W_H1 else 2)
    hi_only = wdtype == L.W_FP8H and (rows + 15) // 16 < 16
    if wdtype == L.W_H1:
        assert torch.equal(w_eff, w)          # one plane holds a bf16-representable matrix exactly
    # the producer's side of the fused norm: x*gain travels as planes, sum(x^2) as per-tile partials
    xd = x.to(DEV)
    xs, ss = ops.split_rows(ops.pack_rows(xd), rows, K, gain.to(DEV) if norm else None, want_ss=norm)
    x64 = x.double()
    xn = x64 if gain is None else (x64 * gain.double()) * torch.rsqrt(torch.mean(x64 * x64, dim=-1, keepdim=True) + eps)
    if hi_only:      # the plane the consumer reads: fp16(x * gain) as the producer wrote it; rinv multiplies the sum afterwards
        xg = (x if gain is None else x * gain).half().double()
        xn = xg if gain is None else xg * torch.rsqrt(torch.mean(x64 * x64, dim=-1, keepdim=True) + eps)
    y = xn @ w_eff.double().t()
    if epi == L.EPI_SWIGLU:
        yv = y.view(rows, N // 32, 2, 16)
        ref = (F.silu(yv[:, :, 0]) * yv[:, :, 1]).reshape(rows, n_out)
    elif epi == L.EPI_RESID:
        ref = res.double() + y
    else:
        ref = y
    wp = ops.pack_weight(w.to(DEV), wdtype)
    resp = ops.pack_rows(res.to(DEV)) if res is not None else None
    want = epi != L.EPI_LOGITS
    if epi == L.EPI_SWIGLU:
        gain_out = torch.ones(n_out)     # the SwiGLU output feeds w2 directly: no norm gain in between
    out, osp, oss = ops.gemv_pair(wp, xs, rows, N, K, epi, ss_in=ss if norm else None, residual=resp,
                                  gain_out=gain_out.to(DEV) if want and epi != L.EPI_SWIGLU else None, want_split=want,
                                  want_ss=want and epi != L.EPI_SWIGLU, eps=eps, wdtype=wdtype)
    got = out.cpu() if epi == L.EPI_LOGITS else ops.unpack_rows(out, rows, n_out).cpu()
    assert rel_err(got.double(), ref) < 3e-6, rel_err(got.double(), ref)
    if want:
        planes = ops.unsplit_rows(osp, rows, n_out).cpu()
        # hi = fp16(v), lo = fp16(v - hi) of v = out * gain_out: the planes' own definition, bit for bit
        v = got * gain_out
        assert torch.equal(planes[0], v.half().float()) and torch.equal(planes[1], (v - v.half().float()).half().float())
        assert float(((planes[0] + planes[1]) - v).abs().max()) <= float(v.abs().max()) * 2.0 ** -22
        if oss is not None:
            ssum = oss.view(-1, n_out // 16, 16).sum(1).reshape(-1)[:rows].cpu()
            assert rel_err(ssum, (got * got).sum(-1)) < 1e-5


@pytest.mark.parametrize("K,N,epi,norm", [(1536, 8192, L.EPI_SWIGLU, True), (4096, 1536, L.EPI_RESID, False), (1536, 4608, L.EPI_STORE, True)])
def test_prefill_gemm_tilings_are_bit_identical(K, N, epi, norm):
    """The prefill GEMM picks a workgroup height per GEMM, cuts a GEMM into two launches, or uses 128 x 128 workgroups (gemv3.hip
    launch_gemm4): every output element is the same sum of the same products in the same k order whatever the tiling, so a
    prompt-sized GEMM must come out bit-identical with the choices forced off (debug flag bits 5, 22, 23)."""
    rows = 2656
    g = torch.Generator().manual_seed(K + N)
    w = torch.randn(N, K, generator=g) * 0.02
    x = torch.randn(rows, K, generator=g)
    gain = (torch.rand(K, generator=g) + 0.5).to(DEV)
    n_out = N // 2 if epi == L.EPI_SWIGLU else N
    res = ops.pack_rows(torch.randn(rows, n_out, generator=g).to(DEV)) if epi == L.EPI_RESID else None
    xs, ss = ops.split_rows(ops.pack_rows(x.to(DEV)), rows, K, gain if norm else None, want_ss=norm)
    wp = ops.pack_weight(w.to(DEV), L.W_H2)
    outs = []
    try:
        for flags in (0, 1 << 22, 1 << 23, 32):
            L.lib().vaura_set_debug_flags(flags)
            out, osp, oss = ops.gemv_pair(wp, xs, rows, N, K, epi, ss_in=ss if norm else None, residual=res, want_split=True,
                                          want_ss=epi == L.EPI_RESID, wdtype=L.W_H2)
            outs.append((out.clone(), osp.clone(), None if oss is None else oss.clone()))
    finally:
        L.lib().vaura_set_debug_flags(0)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
        assert (o[2] is None and outs[0][2] is None) or torch.equal(o[2], outs[0][2])


@pytest.mark.parametrize("rows", [16, 5, 40])
def test_k_split_gemv_partials_add_up(rows):
    """Decode qkv instance: two workgroup sets over the two halves of K write two partial outputs which the consumer
    (attention) adds on load — their sum is the fused-norm GEMV."""
    K, N = 1536, 4608
    g = torch.Generator().manual_seed(rows)
    w = synth.to_bf16_exact(torch.randn(N, K, generator=g) * 0.02)
    x = torch.randn(rows, K, generator=g)
    gain = torch.rand(K, generator=g) + 0.5
    xs, ss = ops.split_rows(ops.pack_rows(x.to(DEV)), rows, K, gain.to(DEV), want_ss=True)
    rp = (rows + 15) // 16 * 16
    out2 = torch.full((rp * N,), float("nan"), device=DEV)
    out, _, _ = ops.gemv_pair(ops.pack_weight(w.to(DEV), L.W_H1), xs, rows, N, K, L.EPI_STORE, ss_in=ss, out_khalf2=out2)
    x64 = x.double()
    ref = ((x64 * gain.double()) * torch.rsqrt(torch.mean(x64 * x64, dim=-1, keepdim=True) + 1e-5)) @ w.double().t()
    a, b = ops.unpack_rows(out, rows, N).cpu().double(), ops.unpack_rows(out2, rows, N).cpu().double()
    assert rel_err(a + b, ref) < 3e-6
    assert float(b.abs().max()) > 0.01 and rel_err(a, ref) > 0.1        # each half really is a partial
    with pytest.raises(L.VauraHipError):                                 # only that instance is compiled
        ops.gemv_pair(ops.pack_weight(w[:1536].to(DEV), L.W_H1), xs, rows, 1536, K, L.EPI_RESID,
                      residual=torch.zeros(rp * 1536, device=DEV), out_khalf2=out2)


def test_fp8_pack_matches_host_quantiser():
    """Device quantiser (vaura_pack_weight, VAURA_W_FP8) == the torch statement of the format: same
    power-of-two row scales, same round-to-nearest-even e4m3 codes, laid out as fp8 tile pairs."""
    from vaura_amd import quant
    g = torch.Generator().manual_seed(21)
    N, K = 48, 128
    w = torch.randn(N, K, generator=g) * torch.logspace(-4, 1, N)[:, None]
    w[5] = 0.0
    w[6, 3] = 448.0 * 2.0 ** -7      # exactly on a scale boundary
    w[6, 4:] *= 1e-3
    packed = ops.pack_weight(w.to(DEV), L.W_FP8).cpu()
    assert packed.numel() == N * K + 4 * N
    scales = packed[N * K:].view(torch.float32)
    ref_s = quant.fp8_row_scales(w)
    assert torch.equal(scales, ref_s)
    codes = (w / ref_s[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)          # (N, K)
    # tile pairs: [N/16][K/64][lane = n%16 + 16*q][h*8 + j]  <-  k = kg2*64 + 32*h + 8*q + j
    ref_tiles = codes.view(N // 16, 16, K // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).reshape(-1)
    assert torch.equal(packed[:N * K], ref_tiles)


def test_gemv_rejects_unsupported_shapes():
    w = torch.zeros(16 * 40 * 4, dtype=torch.uint8, device=DEV)
    x = torch.zeros(16 * 40, device=DEV)
    out = torch.zeros(16 * 16, device=DEV)
    rc = L.lib().vaura_gemv(L.ptr(w), L.W_F32, L.ptr(x), 0, 0, L.ptr(out), 16, 16, 40, 0, 1e-5, L.current_stream())
    assert rc == -1  # K % 32 != 0 -> VAURA_ERR_ARG
    rc = L.lib().vaura_gemv(L.ptr(w), L.W_F32, L.ptr(x), 0, 0, L.ptr(out), 16, 16, 64, 0, 1e-5, L.current_stream())
    assert rc == -2  # depth not compiled -> VAURA_ERR_SHAPE


@pytest.mark.parametrize("rows,steps", [(2, 5), (16, 40), (3, 130)])
def test_attention_step_matches_oracle(rows, steps):
    from oracle.decoder_oracle import apply_rope, rope_table
    H, hd, D = 16, 96, 1536
    max_len = 160
    g = torch.Generator().manual_seed(rows + steps)
    rope = rope_table(max_len, hd)
    kc = torch.zeros(rows, H, max_len, hd, device=DEV)
    vc = torch.zeros_like(kc)
    k_ref = torch.zeros(rows, H, max_len, hd)
    v_ref = torch.zeros_like(k_ref)
    rope_d = rope.to(DEV)
    for pos in range(steps):
        qkv = torch.randn(rows, 3 * D, generator=g)
        out = ops.attention_step(ops.pack_rows(qkv.to(DEV)), rope_d, kc, vc, rows, H, hd, pos)
        q, k, v = qkv.split([D, D, D], dim=-1)
        q = apply_rope(q.view(rows, 1, H, hd), rope[pos:pos + 1]).transpose(1, 2)
        k = apply_rope(k.view(rows, 1, H, hd), rope[pos:pos + 1]).transpose(1, 2)
        k_ref[:, :, pos] = k[:, :, 0]
        v_ref[:, :, pos] = v.view(rows, H, hd)
        s = torch.matmul(q, k_ref[:, :, :pos + 1].transpose(-1, -2)) / math.sqrt(hd)
        ref = torch.matmul(torch.softmax(s, -1), v_ref[:, :, :pos + 1]).transpose(1, 2).reshape(rows, D)
        got = ops.unpack_rows(out, rows, D).cpu()
        assert rel_err(got, ref) < 3e-6, (pos, rel_err(got, ref))
    assert torch.equal(kc[:, :, :steps].cpu(), k_ref[:, :, :steps]) or rel_err(kc[:, :, :steps].cpu(), k_ref[:, :, :steps]) < 1e-6
    assert torch.equal(vc[:, :, :steps].cpu(), v_ref[:, :, :steps])


@pytest.mark.parametrize("rows,n_split", [(4, 4), (2, 8), (5, 2)])
def test_range_split_attention_matches_oracle(rows, n_split):
    """Long-cache decode attention split over several workgroups per (row, head) (configs[3] regime): single steps
    at positions around every block / split edge against the fp32 restatement, on a cache filled with random
    (already rotated) keys and values; also the generic one-workgroup kernel on the same inputs."""
    from oracle.decoder_oracle import apply_rope, rope_table
    H, hd, D = 16, 96, 1536
    max_len = 1024
    g = torch.Generator().manual_seed(rows * 10 + n_split)
    rope = rope_table(max_len, hd)
    k_ref = torch.randn(rows, H, max_len, hd, generator=g)
    v_ref = torch.randn(rows, H, max_len, hd, generator=g)
    rope_d = rope.to(DEV)
    for pos in [0, 1, 63, 64, 65, 255, 256, 257, 511, 512, 700, 1023]:
        qkv = torch.randn(rows, 3 * D, generator=g)
        q, k, v = qkv.split([D, D, D], dim=-1)
        q = apply_rope(q.view(rows, 1, H, hd), rope[pos:pos + 1]).transpose(1, 2)
        k = apply_rope(k.view(rows, 1, H, hd), rope[pos:pos + 1]).transpose(1, 2)
        kk = torch.cat([k_ref[:, :, :pos], k], dim=2)
        vv = torch.cat([v_ref[:, :, :pos], v.view(rows, H, 1, hd)], dim=2)
        sc = torch.matmul(q, kk.transpose(-1, -2)) / math.sqrt(hd)
        ref = torch.matmul(torch.softmax(sc, -1), vv).transpose(1, 2).reshape(rows, D)
        for split in (n_split, 1):
            kc, vc = k_ref.to(DEV), v_ref.to(DEV)
            qp = ops.pack_rows(qkv.to(DEV))
            out = ops.attention_step_split(qp, rope_d, kc, vc, rows, H, hd, pos, split) if split > 1 else \
                ops.attention_step(qp, rope_d, kc, vc, rows, H, hd, pos)
            got = ops.unpack_rows(out, rows, D).cpu()
            assert rel_err(got, ref) < 3e-6, (pos, split, rel_err(got, ref))
            # the new position was appended (by exactly one workgroup), nothing else was touched
            assert rel_err(kc[:, :, pos].cpu(), k[:, :, 0]) < 1e-6 and torch.equal(vc[:, :, pos].cpu(), v.view(rows, H, hd))
            if pos > 0:
                assert torch.equal(kc[:, :, :pos].cpu(), k_ref[:, :, :pos])
    assert L.lib().vaura_attention_splits(4, 16, 1024) == 4 and L.lib().vaura_attention_splits(16, 16, 1024) == 1
    assert L.lib().vaura_attention_splits(4, 16, 256) == 1 and L.lib().vaura_attention_splits(1, 16, 1024) == 8


SAMPLE_CASES = [("topk1", dict(top_k=1, top_p=0.0)), ("topk128", dict(top_k=128, top_p=0.0)),
                ("topk250", dict(top_k=250, top_p=0.0)), ("topp90", dict(top_k=250, top_p=0.9)),
                ("topp30", dict(top_k=0, top_p=0.3)), ("plain", dict(top_k=0, top_p=0.0))]


@pytest.mark.parametrize("temp", [1.0, 0.7])
@pytest.mark.parametrize("name,kw", SAMPLE_CASES)
def test_sampler_matches_reference_goldens(golden, name, kw, temp):
    """Same vectors the reference's sample_top_k/top_p/multinomial produced (utils/utils.py:139-196)."""
    g = golden("sampling.npz")
    logits = torch.from_numpy(g["logits"])
    noise = synth.exp_noise(1, 27, 1024, int(g["noise_seed"]))
    tok = ops.sample(logits.to(DEV), 3, use_sampling=True, temp=temp, noise=noise.to(DEV), **kw)
    got, ref = tok.cpu().numpy(), g[f"{name}_t{temp}_tok"]
    if kw["top_p"] > 0:
        # rows (0,0) and (1,3) hold exact ties; the reference's torch.sort(descending=True) is not
        # stable, so the rank (and therefore the noise) a tied token gets is unspecified there.
        keep = np.ones((3, 9), dtype=bool)
        keep[0, 0] = keep[1, 3] = False
        assert np.array_equal(got[keep], ref[keep])
    else:
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("temp", [1.0, 0.7])
@pytest.mark.parametrize("name,kw", SAMPLE_CASES)
def test_reference_named_sampling_functions(golden, name, kw, temp):
    """``vaura_amd.utils.sample_top_k / sample_top_p / multinomial`` — the reference's names and argument (PROBABILITIES, as
    utils/utils.py:139-196 takes them) — on the probabilities the reference itself fed: softmax(logits / temp) computed the way
    vaura_model.py:816-817 does.  Same goldens as above."""
    from vaura_amd import utils as vu
    g = golden("sampling.npz")
    probs = torch.softmax(torch.from_numpy(g["logits"]) / temp, dim=-1).to(DEV)
    noise = synth.exp_noise(1, 27, 1024, int(g["noise_seed"]))[0].to(DEV)
    if kw["top_p"] > 0:
        tok = vu.sample_top_p(probs, kw["top_p"], noise=noise)
    elif kw["top_k"] > 0:
        tok = vu.sample_top_k(probs, kw["top_k"], noise=noise)
    else:
        tok = vu.multinomial(probs, num_samples=1, noise=noise)
    got, ref = tok.cpu().numpy(), g[f"{name}_t{temp}_tok"]
    assert got.shape == ref.shape == (3, 9, 1)
    keep = np.ones((3, 9), dtype=bool)
    if kw["top_p"] > 0:
        keep[0, 0] = keep[1, 3] = False          # exact ties: the reference's unstable sort leaves their ranks unspecified
    assert np.array_equal(got[keep], ref[keep])
    with pytest.raises(NotImplementedError):
        vu.multinomial(probs, num_samples=2)


def test_sampler_greedy_cfg_and_ties():
    from oracle import sampling_oracle as so
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(6, 9, 1024, generator=g)
    lg[0, 0, 100] = lg[0, 0, 7] = lg[0, 0].max() + 1.0  # exact tie: first index wins (torch.argmax)
    ref = so.next_token(so.cfg_mix(lg, 6.0), use_sampling=False, temp=1.0, top_k=0, top_p=0.0, noise=None)
    got = ops.sample(lg.to(DEV), 3, use_sampling=False, cfg_scale=6.0)
    assert torch.equal(got.cpu(), ref)
    ref1 = so.next_token(lg, use_sampling=False, temp=1.0, top_k=0, top_p=0.0, noise=None)
    got1 = ops.sample(lg.to(DEV), 6, use_sampling=False)
    assert torch.equal(got1.cpu(), ref1) and int(got1[0, 0, 0]) == 7


def test_sampler_philox_is_seeded_and_shard_invariant():
    g = torch.Generator().manual_seed(9)
    lg = torch.randn(4, 9, 1024, generator=g).to(DEV)
    a = ops.sample(lg, 4, use_sampling=True, top_k=250, seed=11, step=3)
    b = ops.sample(lg, 4, use_sampling=True, top_k=250, seed=11, step=3)
    c = ops.sample(lg, 4, use_sampling=True, top_k=250, seed=12, step=3)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # clips 2..3 sampled on "another rank" with clip_base=2 draw the same tokens
    d = ops.sample(lg[2:], 2, use_sampling=True, top_k=250, seed=11, step=3, clip_base=2)
    assert torch.equal(a[2:], d)
    # every sampled token is inside the top-k set
    p = torch.softmax(lg, -1)
    kth = torch.topk(p, 250, dim=-1).values[..., -1:]
    assert bool((torch.gather(p, -1, a) >= kth).all())


@pytest.mark.parametrize("T,Tp", [(4, 0), (55, 0), (220, 0), (221, 166), (20, 8)])
def test_pattern_kernels_match_reference_goldens(golden, T, Tp):
    g = golden("patterns.npz")
    k = f"T{T}_p{Tp}"
    seq = ops.pattern_build(torch.from_numpy(g[k + "_codes"].astype(np.int64)).to(DEV), 1024)
    assert np.array_equal(seq.cpu().numpy(), g[k + "_seq"])
    rev = ops.pattern_revert(torch.from_numpy(g[k + "_filled"].astype(np.int64)).to(DEV), T, -1)
    assert np.array_equal(rev.cpu().numpy(), g[k + "_rev"])


def test_pattern_round_trip_full_size():
    g = torch.Generator().manual_seed(0)
    codes = torch.randint(0, 1024, (64, 9, 220), generator=g).to(DEV)
    seq = ops.pattern_build(codes, 1024)
    assert seq.shape == (64, 9, 229)
    assert torch.equal(ops.pattern_revert(seq, 220, -1), codes)


def test_sampler_randomised_against_oracle():
    """40 random (logit scale, temperature, top-k / top-p / plain, CFG on/off, batch) settings with recorded Exp(1)
    noise: the kernel's tokens against the oracle's (which is pinned to the reference's sample_top_k / sample_top_p /
    multinomial by sampling.npz).  Random floats hold no exact ties, so top-p is compared in full.  The bar is EQUALITY; a
    differing draw must come with a proof (in fp64) that it sits on a near-tie."""
    from oracle import sampling_oracle as so
    rng = np.random.default_rng(7)
    bad = 0
    total = 0
    for trial in range(40):
        B = int(rng.integers(1, 6))
        cfg_on = bool(rng.integers(0, 2))
        rows = 2 * B if cfg_on else B
        g = torch.Generator().manual_seed(1000 + trial)
        logits = torch.randn(rows, 9, 1024, generator=g) * float(rng.choice([0.05, 1.0, 4.0]))
        temp = float(rng.choice([0.7, 1.0, 1.3]))
        mode = int(rng.integers(0, 4))
        top_k = int(rng.choice([1, 3, 64, 250, 1000, 1024])) if mode in (0, 1) else 0
        top_p = float(rng.choice([0.05, 0.5, 0.9, 0.999])) if mode == 2 else 0.0
        cfg_scale = float(rng.choice([2.0, 6.0])) if cfg_on else 1.0
        noise = torch.empty(B * 9, 1024).exponential_(1, generator=g)
        mixed = so.cfg_mix(logits, cfg_scale) if cfg_on else logits
        ref = so.next_token(mixed, use_sampling=True, temp=temp, top_k=top_k, top_p=top_p, noise=noise)
        got = ops.sample(logits.to(DEV), B, use_sampling=True, temp=temp, top_k=top_k, top_p=top_p, cfg_scale=cfg_scale,
                         noise=noise.to(DEV)).cpu()
        total += got.numel()
        assert got.shape == ref.shape == (B, 9, 1)
        # Equality is the bar.  A differing draw is admitted only with a PROOF that fp32 rounding alone decides it: in fp64, either
        # the two tokens' p/q ratios tie to 1e-5 relative (the argmax of the draw), or one of them sits within 1e-5 relative of
        # the top-k threshold / the top-p cut (membership of the kept set).  Anything else fails.
        for b, k, _ in torch.nonzero(got != ref).tolist():
            bad += 1
            p64 = torch.softmax(mixed[b, k].double() / temp, -1)
            q64 = noise.reshape(B, 9, 1024)[b, k].double()
            tg, tr = int(got[b, k, 0]), int(ref[b, k, 0])
            rg, rr = float(p64[tg] / q64[tg]), float(p64[tr] / q64[tr])
            near = abs(rg - rr) <= 1e-5 * max(rg, rr)
            if top_p > 0.0:
                ps, _ = torch.sort(p64, descending=True)
                cut = torch.cumsum(ps, 0) - ps
                near |= bool(((cut - top_p).abs() <= 1e-5).any())
            elif top_k > 0:
                kth = float(torch.topk(p64, min(top_k, 1024)).values[-1])
                near |= min(abs(float(p64[tg]) - kth), abs(float(p64[tr]) - kth)) <= 1e-5 * kth
            assert near, (trial, b, k, tg, tr, rg, rr)
    print(f"sampler: {bad} of {total} draws differ from the oracle, each on a proven near-tie")
    assert bad <= 3, f"{bad} of {total} draws differ from the oracle"


def test_pattern_kernels_randomised_against_oracle():
    """Random (B, T, prompt length): vaura_pattern_build / vaura_pattern_revert against the numpy restatement of
    Pattern.build_pattern_sequence / revert_pattern_sequence (pinned to the reference by patterns.npz), bit-exact."""
    from oracle import pattern_oracle as po
    rng = np.random.default_rng(11)
    for _ in range(12):
        B, T = int(rng.integers(1, 7)), int(rng.integers(1, 300))
        Tp = int(rng.integers(0, T))
        codes = rng.integers(0, 1024, size=(B, 9, T)).astype(np.int64)
        codes[:, :, Tp:] = -1
        seq_ref = po.build_sequence(codes, 1024)[0]
        seq = ops.pattern_build(torch.from_numpy(codes).to(DEV), 1024).cpu().numpy()
        assert np.array_equal(seq, seq_ref), (B, T, Tp)
        filled = rng.integers(0, 1025, size=seq_ref.shape).astype(np.int64)
        rev_ref = po.revert_sequence(filled, T, -1)[0]
        rev = ops.pattern_revert(torch.from_numpy(filled).to(DEV), T, -1).cpu().numpy()
        assert np.array_equal(rev, rev_ref), (B, T, Tp)


# ----------------------------------------------------------------------------- one codec convolution per precision
_CONV_SHAPES = [  # (Cin, Cout, k, dilation, stride)  — the decoder's layer geometries (DAC 44.1 kHz)
    (1536, 768, 16, 1, 8), (768, 768, 7, 9, 1), (768, 768, 1, 1, 1), (384, 384, 7, 3, 1), (384, 192, 8, 1, 4),
    (192, 192, 7, 1, 1), (192, 192, 1, 1, 1), (192, 96, 4, 1, 2), (96, 96, 7, 9, 1), (96, 96, 1, 1, 1)]


def test_snake_sine():
    """The codec's activation (vaura_snake = the snake_f of every conv epilogue; Snake1d of descript-audio-codec 1.0.0, restated in
    oracle/dac_oracle.py::snake) with its own sin^2 (csrc/dac.hip::snake_sin2) against fp64 and against the oracle's fp32 form:
    ordinary activations, the whole fp16-plane range (|x| up to 6e4, alpha up to 8), tiny alpha, exact zeros and multiples of pi/2.
    The bar is absolute on sin^2 in [0, 1]: 4e-7 (the fp32 torch.sin squared is itself 1.3e-7 from fp64), i.e. error / alpha-inverse."""
    from oracle import dac_oracle
    from vaura_amd.engine import snake
    g = torch.Generator().manual_seed(5)
    C = 96
    alpha = torch.cat([torch.rand(C - 6, generator=g) * 3 + 0.05, torch.tensor([1e-3, 8.0, 1.0, 0.5, 2.0, 1e-6])])
    rows = [torch.randn(4000, C, generator=g) * 3, torch.randn(2000, C, generator=g) * 300,
            (torch.rand(2000, C, generator=g) * 2 - 1) * 6.0e4,
            torch.zeros(1, C), (torch.arange(1, 65, dtype=torch.float64)[:, None] * (torch.pi / 2) / alpha.double()[None]).float()]
    x = torch.cat(rows)
    got = snake(x, alpha, DEV).cpu().double()
    ref = x.double() + (alpha.double() + 1e-9).reciprocal() * torch.sin(alpha.double() * x.double()) ** 2
    # the product alpha * x is rounded to fp32 before the sine by the reference too (torch fp32): compare on that argument
    arg = (alpha * x).double()
    ref32arg = x.double() + (alpha.double() + 1e-9).reciprocal() * torch.sin(arg) ** 2
    s2_err = ((got - ref32arg) * (alpha.double() + 1e-9)).abs()          # error of sin^2 itself
    s2_err = s2_err - (x.double().abs() * 2.0 ** -23 * (alpha.double() + 1e-9))   # minus the final add's own rounding (|y| ~ |x|)
    ora = dac_oracle.snake(x, alpha).double()
    ora_err = ((ora - ref32arg) * (alpha.double() + 1e-9)).abs() - (x.double().abs() * 2.0 ** -23 * (alpha.double() + 1e-9))
    print(f"snake: sin^2 max abs error {float(s2_err.max()):.2e} (the oracle's fp32 form: {float(ora_err.max()):.2e}); "
          f"max |got - oracle| / max(1, |x|) = {float(((got - ora).abs() / x.double().abs().clamp(min=1)).max()):.2e}")
    assert torch.isfinite(got).all()
    assert float(s2_err.max()) <= 4e-7, float(s2_err.max())
    assert float(((got - ref).abs() / (1 + ref.abs())).max()) <= 2e-2      # vs exact fp64 incl. the fp32 product alpha * x (ulp(6e4 * 8) = 0.03 rad)
    small = x.abs().amax(dim=1) < 50
    assert float((got[small] - ora[small]).abs().max()) <= 2e-6


@pytest.mark.parametrize("shape", _CONV_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("precision", ["f32", "f16pair", "f16", "f16pair_w8", "mx8"])
def test_codec_convolution_per_precision(shape, precision):
    """vaura_dac_conv against torch fp64 on the SAME numbers the kernel multiplies: for "mx8" the input is rounded by
    quant.mx8_effective_activation (what the producing kernel stores) and the weight by quant.fp8_effective_weight (what the
    packed e4m3 stream holds), so what is left is accumulation — every layer geometry of the decoder, ragged lengths, both
    ends of the sequence (halo rows outside [0, L) are zeros).  Tolerances, relative to max |ref|: 2e-6 for the fp16-pair paths (fp32
    accumulation order); 6e-6 for the exact-fp32 path (ONE fp32 chain over taps x Cin products — 5 376 for the 768-channel
    7-tap layer, measured 3.1e-6 — where the MFMA forms on 32-deep groups first); 5e-5 for mx8 — v_mfma_scale_f32_16x16x128_f8f6f4 does not accumulate its 128 products
    to fp32 accuracy (measured 1.5e-5 .. 2.5e-5 on every geometry, one-tap layers included; a wrong k order, scale or
    layout gives errors of order 1)."""
    import torch.nn.functional as F
    from vaura_amd import quant
    from vaura_amd.engine import CodecConvOp
    cin, cout, k, dil, stride = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + k + dil)
    B, Lin = 2, 150 if stride == 1 else 37
    x = torch.randn(B, Lin, cin, generator=g) * torch.rand(B, Lin, 1, generator=g) * 3
    bias = torch.randn(cout, generator=g) * 0.1
    if stride > 1:
        w = torch.randn(cin, cout, k, generator=g) / (cin * 2) ** 0.5
        flat = lambda t: t.permute(1, 0, 2).reshape(cout, -1)
        unflat = lambda t: t.reshape(cout, cin, k).permute(1, 0, 2)
    else:
        w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
        flat = lambda t: t.reshape(cout, -1)
        unflat = lambda t: t.reshape(cout, cin, k)
    xe = x
    if precision in ("mx8", "f16pair_w8"):       # both multiply by fp8-representable weights; mx8 also rounds the activations
        w = unflat(quant.fp8_effective_weight(flat(w))).contiguous()
    if precision == "mx8":
        xe = quant.mx8_effective_activation(x)
    if precision == "f16":                       # plain fp16 operands: the hi planes of the input and of the weight
        xe, w = x.half().float(), w.half().float()
    xd, wd = xe.double().transpose(1, 2), w.double()
    if stride > 1:
        ref = F.conv_transpose1d(xd, wd, bias.double(), stride=stride, padding=(stride + 1) // 2)
    else:
        ref = F.conv1d(xd, wd, bias.double(), dilation=dil, padding=(k - 1) // 2 * dil)
    ref = ref.transpose(1, 2)
    got = CodecConvOp(w, bias, dil, stride, precision, DEV)(x.to(DEV)).cpu().double()      # "f16": the op itself drops the lo planes of x
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"codec conv {shape} {precision}: max err / max |ref| = {err:.2e}")
    assert err <= {"mx8": 5e-5, "f32": 6e-6}.get(precision, 2e-6), err
