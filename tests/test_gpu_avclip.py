"""Row f2 on the GPU: vaura_avclip_forward (C ABI) through AvclipEngine / the MotionFormer plugin against vectors produced
by the reference's own classes (tests/golden/avclip*.npz) and against the oracle.

Tolerance: features within 1e-4 max-abs (values are O(0.4)).  Why not tighter: the 38 linear layers of a forward run on
(hi, lo) fp16 pairs — 22 significand bits per operand against fp32's 24 — with fp32 accumulation in a different order than
the reference's BLAS; LayerNorm, softmax and GELU are fp32.  The observed error is printed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vaura_amd import synth
from vaura_amd.engine import AvclipEngine

DEV = "cuda:0"
TOL = 1e-4


@pytest.fixture(scope="module")
def avclip_engine():
    return AvclipEngine(synth.FULL_AVCLIP, synth.avclip_state_dict(seed=0), DEV)


def test_one_segment_matches_reference_and_oracle(avclip_engine, golden):
    from oracle import avclip_oracle as ao
    g = golden("avclip.npz")
    frames = synth.video_frames(1, 1, seed=int(g["frame_seed"]))
    got = avclip_engine.forward(frames.to(DEV)).cpu()
    ref = torch.from_numpy(g["feats"])
    assert got.shape == ref.shape == (1, 1, 8, 768)
    err = float((got - ref).abs().max())
    with torch.no_grad():
        orc = ao.forward(synth.avclip_state_dict(seed=0), frames)
    err_o = float((got - orc).abs().max())
    print(f"avclip features: max-abs err vs reference {err:.3e}, vs oracle {err_o:.3e} (feature std {float(ref.std()):.3f})")
    assert err < TOL and err_o < TOL


def test_batched_segments_match_reference(avclip_engine, golden):
    g = golden("avclip_b2s2.npz")
    frames = synth.video_frames(2, 2, seed=int(g["frame_seed"]))
    got = avclip_engine.forward(frames.to(DEV)).cpu()
    ref = torch.from_numpy(g["feats"])
    assert got.shape == ref.shape == (2, 2, 8, 768)
    assert float((got - ref).abs().max()) < TOL
    # several passes (workspace smaller than the batch) give the same rows
    avclip_engine2 = AvclipEngine(synth.FULL_AVCLIP, synth.avclip_state_dict(seed=0), DEV)
    avclip_engine2.MAX_SEGMENTS = 3
    assert torch.equal(avclip_engine2.forward(frames.to(DEV)).cpu(), got)


def test_plugin_feeds_the_generate_path(golden):
    """MotionFormer plugin (reference keywords + state-dict keys) -> (B, S, 8, 768) -> flattened like
    VAURAModel._handle_visual_conditioning (vaura_model.py:199-204) -> the decoder's conditioning input."""
    from vaura_amd.feature_extractor import MotionFormer
    g = golden("avclip_b2s2.npz")
    fe = MotionFormer(extract_features=True, ckpt_path=None, factorize_space_time=True, agg_space_module="TransformerEncoderLayer",
                      agg_time_module="torch.nn.Identity", add_global_repr=False)
    fe.load_state_dict(synth.avclip_state_dict(seed=0), strict=True)
    fe = fe.to(DEV)
    frames = synth.video_frames(2, 2, seed=int(g["frame_seed"])).to(DEV)
    feats, glob = fe(frames)
    assert glob is None and feats.shape == (2, 2, 8, 768)
    assert float((feats.cpu() - torch.from_numpy(g["feats"])).abs().max()) < TOL
    same, _ = fe(feats)                      # pre-extracted features pass through
    assert same is feats


def test_lds_dma_linear_layers_are_bit_identical_to_the_register_staged_ones(avclip_engine):
    """csrc/dac.hip linear_dma_kernel (operands by LDS-DMA in whole cache lines, fragments read early, per-wave epilogue; the default
    since round 5) against round 4's register-staged linear_pair_kernel (second debug word, bit 5), its late-read instance (bit 6)
    and round 4's tile order (bits 12..15 = 15: no column panels): the same products summed in the same order, so the features must be
    BIT-identical — 2 clips x 2 segments, a ragged last row tile (1 x 1: 1569 rows = 12 tiles + 33) and 12 segments (147 row tiles)."""
    from vaura_amd import _lib as L
    lib = L.lib()
    try:
        for B, S in ((2, 2), (1, 1), (3, 4)):
            frames = synth.video_frames(B, S, seed=31 + B).to(DEV)
            outs = []
            for f2 in (0, 32, 64, 15 << 12):
                lib.vaura_set_debug_flags2(f2)
                outs.append(avclip_engine.forward(frames).clone())
            torch.cuda.synchronize()
            assert torch.isfinite(outs[0]).all()
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (B, S)
    finally:
        lib.vaura_set_debug_flags2(0)
