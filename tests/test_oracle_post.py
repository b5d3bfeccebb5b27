"""Row f3 (post-codec scaling): the oracle against vectors produced by the reference's own normalize_audio."""
import numpy as np
import pytest
import torch

from oracle import post_oracle

CASES = [("clip", True, 6.0), ("clip", True, 3.0), ("peak", True, 6.0), ("peak", False, 6.0), ("rms", True, 6.0),
         ("rms", False, 6.0)]


@pytest.mark.parametrize("name", ["loud", "nominal", "quiet"])
@pytest.mark.parametrize("strategy,normalize,db", CASES)
def test_normalize_audio_matches_reference(golden, name, strategy, normalize, db):
    g = golden("post.npz")
    wav = torch.from_numpy(g[f"{name}_in"])
    ref = g[f"{name}_{strategy}_n{int(normalize)}_db{int(db)}"]
    got = post_oracle.normalize_audio(wav.clone(), normalize=normalize, strategy=strategy, peak_clip_headroom_db=db).numpy()
    assert np.array_equal(got, ref)


def test_scale_audio_shape_and_dtype():
    wav = torch.randn(1, 1, 64, dtype=torch.float16)
    out = post_oracle.scale_audio(wav, "clip", 44100)
    assert out.shape == (1, 64) and out.dtype == torch.float32 and out.device.type == "cpu"
    assert float(out.abs().max()) <= 10 ** (-6 / 20) + 1e-7
