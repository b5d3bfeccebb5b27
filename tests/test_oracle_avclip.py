"""Row f2 oracle (oracle/avclip_oracle.py) against vectors produced by the reference's own MotionFormer classes
(tests/golden/make_golden.py avclip): features and intermediate token rows, one segment at full width; the (B, S)
batched path on 2 x 2 segments."""
import numpy as np
import torch

from oracle import avclip_oracle as ao
from vaura_amd import synth


def test_avclip_oracle_matches_reference_one_segment(golden):
    g = golden("avclip.npz")
    sd = synth.avclip_state_dict(seed=int(g["weight_seed"]))
    frames = synth.video_frames(1, 1, seed=int(g["frame_seed"]))
    trace = {}
    with torch.no_grad():
        out = ao.forward(sd, frames, trace=trace)
    rows = list(g["rows"])
    for k in ("tokens", "block0", "block11"):
        err = float((trace[k][0, rows] - torch.from_numpy(g[k])).abs().max())
        assert err < 5e-5, (k, err)
    ref = torch.from_numpy(g["feats"])
    assert out.shape == ref.shape == (1, 1, 8, 768)
    assert float((out - ref).abs().max()) < 5e-5 and float(ref.std()) > 0.1


def test_avclip_oracle_batched_segments(golden):
    g = golden("avclip_b2s2.npz")
    sd = synth.avclip_state_dict(seed=int(g["weight_seed"]))
    frames = synth.video_frames(2, 2, seed=int(g["frame_seed"]))
    with torch.no_grad():
        out = ao.forward(sd, frames)
    ref = torch.from_numpy(g["feats"])
    assert out.shape == ref.shape == (2, 2, 8, 768)
    assert float((out - ref).abs().max()) < 5e-5
    assert not np.allclose(ref[0, 0], ref[1, 1])
