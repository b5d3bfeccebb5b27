"""Oracle next-token selection vs the reference's sample_top_k / sample_top_p / multinomial
(utils/utils.py:139-196) run under a seeded global generator (make_golden.py::gold_sampling)."""
import numpy as np
import pytest
import torch

from oracle import sampling_oracle as so
from vaura_amd import synth

CASES = [("topk1", dict(top_k=1, top_p=0.0)), ("topk128", dict(top_k=128, top_p=0.0)),
         ("topk250", dict(top_k=250, top_p=0.0)), ("topp90", dict(top_k=250, top_p=0.9)),
         ("topp30", dict(top_k=0, top_p=0.3)), ("plain", dict(top_k=0, top_p=0.0))]


@pytest.mark.parametrize("temp", [1.0, 0.7])
@pytest.mark.parametrize("name,kw", CASES)
def test_tokens_match_reference(golden, name, kw, temp):
    g = golden("sampling.npz")
    logits = torch.from_numpy(g["logits"])
    noise = synth.exp_noise(1, 27, 1024, int(g["noise_seed"]))[0]
    tok = so.next_token(logits, use_sampling=True, temp=temp, noise=noise, **kw)
    assert np.array_equal(tok.numpy(), g[f"{name}_t{temp}_tok"])


def test_greedy_and_cfg():
    g = torch.Generator().manual_seed(1)
    lg = torch.randn(4, 9, 1024, generator=g)
    mixed = so.cfg_mix(lg, 6.0)
    assert mixed.shape == (2, 9, 1024)
    assert torch.equal(mixed, lg[2:] + (lg[:2] - lg[2:]) * 6.0)
    t = so.next_token(mixed, use_sampling=False, temp=1.0, top_k=250, top_p=0.0, noise=None)
    assert torch.equal(t[..., 0], mixed.argmax(-1))
    # temp <= 0 is greedy too (vaura_model.py:816)
    t2 = so.next_token(mixed, use_sampling=True, temp=0.0, top_k=250, top_p=0.0, noise=None)
    assert torch.equal(t, t2)
