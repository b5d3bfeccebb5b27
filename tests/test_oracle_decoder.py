"""Oracle decoder + generate loop vs the reference run on the same synthetic weights
(make_golden.py::gold_tiny / gold_full_greedy / gold_full_sample)."""
import numpy as np
import pytest
import torch

from oracle import generate_oracle as go
from oracle.decoder_oracle import DecoderOracle
from vaura_amd import synth


@pytest.fixture(scope="module")
def tiny(golden, tiny_sampler_sd):
    g = golden("tiny_model.npz")
    cfg = synth.tiny_sampler(2)
    dec = DecoderOracle(tiny_sampler_sd, cfg.num_layers, cfg.nhead)
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    return g, dec, feats


def test_full_forward_logits(tiny):
    g, dec, feats = tiny
    lg = dec.forward_full(torch.from_numpy(g["fwd_idx"].astype(np.int64)), feats)
    ref = torch.from_numpy(g["fwd_logits"])
    assert (lg[:, :, list(g["fwd_logits_pos"])] - ref).abs().max() < 2e-5


def test_positions_past_video_use_empty_embedding(tiny):
    g, dec, feats = tiny
    lg = dec.forward_full(torch.from_numpy(g["pad_idx"].astype(np.int64)), feats[:1, :4])
    assert (lg[:, :, [27, 28, 30]] - torch.from_numpy(g["pad_logits"])).abs().max() < 2e-5


def test_cached_equals_full(tiny):
    g, dec, feats = tiny
    from oracle.decoder_oracle import CachedDecoder
    idx = torch.from_numpy(g["fwd_idx"].astype(np.int64))
    full = dec.forward_full(idx, feats)
    c = CachedDecoder(dec, feats, 16)
    for p in range(idx.shape[-1]):
        lg = c.step(idx[:, :, p])
        assert (lg - full[:, :, p]).abs().max() < 2e-5


@pytest.mark.parametrize("mode", ["full", "cached"])
def test_generate_cases(tiny, mode):
    g, dec, feats = tiny
    S = 29
    ref = lambda k: torch.from_numpy(g[k].astype(np.int64))
    assert torch.equal(go.generate(dec, feats, 20, mode=mode), ref("greedy_T20"))
    assert torch.equal(go.generate(dec, feats, 20, mode=mode, cfg_scale=6.0), ref("greedy_cfg6_T20"))
    nz = synth.exp_noise(S - 1, 18, 1024, 99)
    assert torch.equal(go.generate(dec, feats, 20, mode=mode, cfg_scale=6.0, use_sampling=True, top_k=250, noise=nz),
                       ref("topk250_cfg6_seed99_T20"))
    nz = synth.exp_noise(S - 1, 18, 1024, 98)
    assert torch.equal(go.generate(dec, feats, 20, mode=mode, use_sampling=True, temp=0.9, top_k=250, top_p=0.8, noise=nz),
                       ref("topp80_t09_seed98_T20"))
    nz = synth.exp_noise(S - 1, 18, 1024, 97)
    assert torch.equal(go.generate(dec, feats, 20, mode=mode, use_sampling=True, noise=nz), ref("plain_seed97_T20"))
    prompt = ref("greedy_T20")[:, :, 5:13]
    assert torch.equal(go.generate(dec, feats, 20, mode=mode, prompt=prompt), ref("prompt8_greedy_T20"))


@pytest.mark.slow
def test_full_size_greedy_tokens(golden, full_sampler_sd):
    """24 layers / 694 M params, B=2, T=220: the KV-cached oracle reproduces the reference's
    (cache-less) greedy tokens exactly, and its logits at recorded steps."""
    g = golden("full_greedy_B2_T220.npz")
    cfg = synth.FULL_SAMPLER
    dec = DecoderOracle(full_sampler_sd, cfg.num_layers, cfg.nhead)
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    trace = {}
    tok = go.generate(dec, feats, 220, mode="cached", trace=trace)
    assert torch.equal(tok, torch.from_numpy(g["tokens"].astype(np.int64)))
    for L, ref in zip(g["logits_steps"], g["logits"]):
        assert (trace["logits"][int(L)] - torch.from_numpy(ref)).abs().max() < 5e-5
    assert float(g["margins"].min()) > 1e-4  # the fixture is not sitting on a near-tie


def test_dac_encode_oracle_matches_hf_cross_check(golden):
    """Row f4 structure check (the reference's own dependency is absent: parity unpinned): the encode restatement
    against transformers' independent DacModel on a reduced-width encoder — latent bit-for-bit, codes identical."""
    import numpy as np
    from oracle import dac_oracle
    g = golden("codec_enc_hf.npz")
    ccfg = synth.CodecCfg(latent_dim=int(g["latent_dim"]), encoder_dim=int(g["encoder_dim"]), encoder_rates=(2, 4, 8, 8),
                          decoder_dim=192)
    sd = dict(synth.codec_state_dict(ccfg, seed=int(g["codec_seed"])))
    sd.update(synth.codec_encoder_state_dict(ccfg, seed=int(g["codec_seed"])))
    wav = torch.from_numpy(g["wav"])
    z = dac_oracle.encode_latent(sd, wav, ccfg.encoder_rates)
    assert float((z - torch.from_numpy(g["z"])).abs().max()) < 1e-5
    codes = dac_oracle.quantize(sd, z, 9)
    assert torch.equal(codes, torch.from_numpy(g["codes"].astype(np.int64)))
    # preprocess pads on the right to a multiple of the hop
    assert dac_oracle.preprocess(torch.zeros(1, 1, 1000), 512).shape[-1] == 1024
    assert dac_oracle.encode(sd, wav[..., :-100], ccfg.encoder_rates).shape == (2, 9, 6)


@pytest.mark.slow
def test_unrounded_checkpoint_golden_first_frames(golden, full_sampler_sd_raw):
    """The reference's run on the UN-rounded checkpoint (make_golden.py full_greedy_raw): the cached oracle reproduces its
    recorded first-forward logits and the first 12 frames of tokens (bounded: the whole run is the GPU suite's job)."""
    g = golden("full_greedy_raw_B2_T220.npz")
    cfg = synth.FULL_SAMPLER
    dec = DecoderOracle(full_sampler_sd_raw, cfg.num_layers, cfg.nhead)
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    trace = {}
    tok = go.generate(dec, feats, 21, mode="cached", trace=trace)       # positions <= 12 do not depend on T
    assert torch.equal(tok[..., :12], torch.from_numpy(g["tokens"].astype(np.int64))[..., :12])
    for L, ref in zip(g["logits_steps"], g["logits"]):
        if int(L) in trace["logits"] and int(L) <= 10:
            assert (trace["logits"][int(L)] - torch.from_numpy(ref)).abs().max() < 5e-5


@pytest.mark.slow
def test_configs3_golden_first_frames(golden):
    """BASELINE configs[3] golden (reference built with block_size_audio=1024, Tv=128, B=1, T=880): the cached oracle with a
    1024-row rope table reproduces the first forwards' logits and the first 20 frames (bounded; GPU suite runs all 880)."""
    g = golden("full_c4_greedy_B1_T880.npz")
    cfg = synth.SamplerCfg(block_size_audio=int(g["block_size_audio"]))
    sd = synth.sampler_state_dict(cfg, seed=int(g["weight_seed"]), round_bf16=True)
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead, block_size=1024)
    feats = synth.video_features(1, tokens=128, seed=int(g["feat_seed"]))
    trace = {}
    tok = go.generate(dec, feats, 29, mode="cached", trace=trace)
    assert torch.equal(tok[..., :20], torch.from_numpy(g["tokens"].astype(np.int64))[..., :20])
    for L, ref in zip(g["logits_steps"], g["logits"]):
        if int(L) in trace["logits"] and int(L) <= 10:
            assert (trace["logits"][int(L)] - torch.from_numpy(ref)).abs().max() < 5e-5
    assert g["tokens"].shape == (1, 9, 880) and float(g["margins"].min()) > 0


@pytest.mark.slow
def test_headline_configuration_golden_first_frames(golden, full_sampler_sd_raw):
    """The reference's run of the HEADLINE configuration's arithmetic (make_golden.py full_sample_raw: un-rounded checkpoint, cfg 6,
    top-k 250 sampled with torch's CPU noise stream, B=2): the cached oracle reproduces the recorded first-forward logits (after the
    CFG mix) and the first 12 frames of tokens (bounded: the whole run at B=8 is the GPU suite's job)."""
    g = golden("full_topk250_cfg6_raw_B2_T220.npz")
    cfg = synth.FULL_SAMPLER
    dec = DecoderOracle(full_sampler_sd_raw, cfg.num_layers, cfg.nhead)
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    nz = synth.exp_noise(29, 18, 1024, int(g["noise_seed"]))            # the first 29 draws of the 228-step stream
    trace = {}
    tok = go.generate(dec, feats, 21, mode="cached", cfg_scale=float(g["cfg_scale"]), use_sampling=True, temp=1.0,
                      top_k=int(g["top_k"]), noise=nz, trace=trace)
    assert torch.equal(tok[..., :12], torch.from_numpy(g["tokens"].astype(np.int64))[..., :12])
    s = float(g["cfg_scale"])
    for L, ref in zip(g["logits_steps"], g["logits"]):
        if int(L) in trace["logits"] and int(L) <= 10:
            r = torch.from_numpy(ref)
            mixed = r[2:] + (r[:2] - r[2:]) * s
            assert (trace["logits"][int(L)] - mixed).abs().max() < 5e-4
    assert g["tokens"].shape == (2, 9, 220) and g["margins"].shape == (228, 2, 9) and float(g["margins"].min()) > 1e-5


@pytest.mark.slow
def test_reference_shipped_defaults_golden_first_frames(golden, full_sampler_sd_raw):
    """configs/generate_vgg.yaml:23-27 as shipped (top-k 128, cfg 6, temperature 1) on the un-rounded checkpoint (make_golden.py
    full_vgg_raw): the cached oracle reproduces the first 12 frames of the reference's tokens (bounded; the GPU suite runs all 220)."""
    g = golden("full_topk128_cfg6_raw_B2_T220.npz")
    cfg = synth.FULL_SAMPLER
    dec = DecoderOracle(full_sampler_sd_raw, cfg.num_layers, cfg.nhead)
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    nz = synth.exp_noise(29, 18, 1024, int(g["noise_seed"]))
    tok = go.generate(dec, feats, 21, mode="cached", cfg_scale=float(g["cfg_scale"]), use_sampling=True, temp=1.0, top_k=int(g["top_k"]), noise=nz)
    assert int(g["top_k"]) == 128 and float(g["cfg_scale"]) == 6.0
    assert torch.equal(tok[..., :12], torch.from_numpy(g["tokens"].astype(np.int64))[..., :12])
    assert g["tokens"].shape == (2, 9, 220) and float(g["margins"].min()) > 1e-5


@pytest.mark.slow
def test_later_longform_chunk_golden_first_steps(golden, full_sampler_sd_raw):
    """A later chunk of the sliding-window caller at full depth (make_golden.py full_chunk_raw: Tp = 166, T = 221, cfg 6): the cached
    oracle — 166 teacher-forced positions, then sampling — reproduces the reference's [cond; null] logits of its first pass (sequence
    length 167) after the CFG mix and the tokens of its first 10 sampled steps, for the top-k-128 run and the greedy one.  Bounded by a
    shorter T (the steps compared, 167..176, do not depend on it); the GPU suite runs all 63 passes."""
    gs, gg = golden("full_chunk_topk128_cfg6_raw_B2_Tp166_T221.npz"), golden("full_chunk_greedy_cfg6_raw_B2_Tp166_T221.npz")
    cfg = synth.FULL_SAMPLER
    dec = DecoderOracle(full_sampler_sd_raw, cfg.num_layers, cfg.nhead)
    feats = synth.video_features(2, seed=int(gs["feat_seed"]))
    prompt = torch.from_numpy(gs["prompt"].astype(np.int64))
    Tshort = 176
    steps = torch.arange(Tshort)[None, :] + 1 + torch.arange(9)[:, None]
    same_steps = (steps <= Tshort)[None].expand(2, -1, -1)               # steps whose validity mask is the one of the T = 221 run
    nz = synth.exp_noise(Tshort + 9 - 167, 18, 1024, int(gs["noise_seed"]))
    trace = {}
    tok = go.generate(dec, feats, Tshort, prompt=prompt, mode="cached", cfg_scale=6.0, use_sampling=True, temp=1.0, top_k=128, noise=nz, trace=trace)
    ref = torch.from_numpy(gs["tokens"].astype(np.int64))[..., :Tshort]
    assert torch.equal(tok[same_steps], ref[same_steps])
    r = torch.from_numpy(gs["logits"][list(gs["logits_steps"]).index(167)])
    assert (trace["logits"][167] - (r[2:] + (r[:2] - r[2:]) * 6.0)).abs().max() < 5e-4
    # greedy: the same prefix, so the same first-pass logits; tokens of clip 0 (clip 1 holds the run's literal tie at step 175)
    tokg = go.generate(dec, feats[:1], Tshort, prompt=prompt[:1], mode="cached", cfg_scale=6.0)
    refg = torch.from_numpy(gg["tokens"].astype(np.int64))[:1, :, :Tshort]
    assert torch.equal(tokg[same_steps[:1]], refg[same_steps[:1]])
    assert float(gg["margins"][175 - 167, 1, 8]) < 1e-5                  # the run's literal tie (clip 1, step 175, codebook 8)


def test_op_level_vectors_of_the_reference_modules(golden):
    """the oracle's ops, one by one, against outputs captured from the reference's OWN modules (make_golden.py ops:
    forward hooks on RMSNorm, Attention, FeedForward, TransformerBlock, AVCLIPEmbedder, DacEmbeddingProjection during one forward of
    the 2-layer model on a trained-like checkpoint; `_repeat_and_pad_video`, `precompute_freqs_cis` and `apply_rotary_emb` called
    directly)."""
    from oracle import decoder_oracle as do
    g = golden("ops.npz")
    cfg = synth.tiny_sampler(2)
    sd = synth.trained_like(synth.sampler_state_dict(cfg, seed=int(g["weight_seed"]), round_bf16=False), seed=int(g["trained_like_seed"]))
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    t = lambda k: torch.from_numpy(g[k])
    sub = lambda x: x[..., ::8]
    close = lambda a, b, tol=2e-5: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    h0 = t("block0_in")
    L = h0.shape[1]
    tab = dec.rope[:L]
    # RMSNorm (llama.py:147-158) with a non-trivial gain
    x = do.rmsnorm(h0, dec.sd["layers.0.attention_norm.weight"], dec.eps)
    assert float(dec.sd["layers.0.attention_norm.weight"].max()) > 5.0 and close(x, t("rmsnorm"))
    # rope table + rotation (llama.py:593-603, 633-650)
    assert close(do.rope_table(40, 96), t("freqs_cis"), 1e-6)
    xr = torch.randn(2, 5, 16, 96, generator=torch.Generator().manual_seed(int(g["rope_seed"])))
    assert close(do.apply_rope(xr, do.rope_table(40, 96)[:5]), t("rope_out"), 1e-6)
    # Attention.forward, FeedForward, TransformerBlock (llama.py:219-283) on the reference's own intermediate inputs
    assert close(sub(dec.attention(t("rmsnorm"), 0, tab)), t("attention"))
    assert close(sub(dec.feed_forward(t("ffn_in"), 0)), t("ffn"))
    b0 = dec.block(h0, 0, tab)
    assert close(sub(b0), t("block0")) and close(sub(dec.block(b0, 1, tab)), t("block1"))
    # AVCLIPEmbedder, _repeat_and_pad_video (positions 28.. read empty_video_emb with 4 video tokens), DacEmbeddingProjection
    feats = synth.video_features(2, tokens=4, seed=int(g["feat_seed"]))
    cp = dec.cond_projection(feats)
    assert close(cp, t("cond_proj"))
    assert close(dec.cond_for_positions(cp, torch.tensor([0, 6, 7, 27, 28, 30])), t("padded_video"))
    idx = torch.from_numpy(g["idx"].astype(np.int64))
    import torch.nn.functional as F
    for k in (0, 8):
        e = F.linear(F.embedding(idx[:, k], dec.sd[f"tok_embeddings.{k}.emb.weight"]), dec.tok_w[k], dec.tok_b[k])
        assert close(sub(e), t(f"tok_emb{k}"))
    # and the whole thing: last-position logits of that forward
    lg = dec.forward_full(idx, feats)
    assert close(lg[:, :, -1, ::16], t("logits_last"))
