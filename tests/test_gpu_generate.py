"""End-to-end parity of the HIP generate loop (C ABI: vaura_prefill_cond / vaura_pattern_* /
vaura_generate_loop) against goldens produced by the reference itself and against the live oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vaura_amd import synth
from vaura_amd.engine import CodecEngine, DecoderEngine
from parity_helpers import assert_tokens_equal, assert_tokens_or_recorded_near_tie

DEV = "cuda:0"
# storages of the streamed matrices: "h2" = (hi, lo) fp16 planes (what "auto" gives an fp32 checkpoint), "h1" = one fp16 plane
# (lossless for the bf16-representable synthetic checkpoint), "f32" = the exact-fp32-MFMA GEMVs kept as a cross-check
ENGINE_KINDS = ["h2", "f32", "h1"]


@pytest.fixture(scope="module", params=ENGINE_KINDS)
def tiny_engine(request, tiny_sampler_sd):
    return DecoderEngine(synth.tiny_sampler(2), tiny_sampler_sd, DEV, wdtype=request.param)


def _ref(g, k):
    return torch.from_numpy(g[k].astype(np.int64))


def test_cond_projection_matches_oracle(tiny_engine, tiny_sampler_sd, golden):
    from oracle.decoder_oracle import DecoderOracle
    g = golden("tiny_model.npz")
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    dec = DecoderOracle(tiny_sampler_sd, 2, 16)
    tiny_engine.prepare(2, 20, 32, True)
    tiny_engine.set_condition(feats.to(DEV))
    got = tiny_engine.cond_projection().cpu()
    ref = dec.cond_projection(torch.cat([feats, dec.null_condition(feats)], 0))
    assert (got - ref).abs().max() < 1e-5


def test_teacher_forced_logits_match_reference(tiny_engine, golden):
    """Transformer.forward semantics (llama.py:445-504): logits at every position."""
    g = golden("tiny_model.npz")
    feats = synth.video_features(2, seed=int(g["feat_seed"]))
    idx = _ref(g, "fwd_idx")
    lg = tiny_engine.logits_all_positions(idx.to(DEV), feats.to(DEV)).cpu()
    ref = torch.from_numpy(g["fwd_logits"])
    assert (lg[:, :, list(g["fwd_logits_pos"])] - ref).abs().max() < 3e-5
    # positions past Tv*7 read empty_video_emb (llama.py:569-572)
    lg2 = tiny_engine.logits_all_positions(_ref(g, "pad_idx").to(DEV), feats[:1, :4].to(DEV)).cpu()
    assert (lg2[:, :, [27, 28, 30]] - torch.from_numpy(g["pad_logits"])).abs().max() < 3e-5


@pytest.mark.parametrize("use_graph", [False, True])
def test_generate_cases_match_reference(tiny_engine, golden, use_graph):
    g = golden("tiny_model.npz")
    feats = synth.video_features(2, seed=int(g["feat_seed"])).to(DEV)
    S = 29
    e = tiny_engine
    kw = dict(use_graph=use_graph)
    assert torch.equal(e.generate_codes(feats, 20, **kw).cpu(), _ref(g, "greedy_T20"))
    assert torch.equal(e.generate_codes(feats, 20, cfg_scale=6.0, **kw).cpu(), _ref(g, "greedy_cfg6_T20"))
    nz = synth.exp_noise(S - 1, 18, 1024, 99)
    assert torch.equal(e.generate_codes(feats, 20, cfg_scale=6.0, use_sampling=True, top_k=250, noise=nz, **kw).cpu(),
                       _ref(g, "topk250_cfg6_seed99_T20"))
    nz = synth.exp_noise(S - 1, 18, 1024, 98)
    assert torch.equal(e.generate_codes(feats, 20, use_sampling=True, temp=0.9, top_k=250, top_p=0.8, noise=nz, **kw).cpu(),
                       _ref(g, "topp80_t09_seed98_T20"))
    nz = synth.exp_noise(S - 1, 18, 1024, 97)
    assert torch.equal(e.generate_codes(feats, 20, use_sampling=True, noise=nz, **kw).cpu(), _ref(g, "plain_seed97_T20"))


def test_prompt_continuation_matches_reference(tiny_engine, golden):
    """The sliding-window caller's shape (scripts/generate.py:327-365), scaled down: Tp=8 of T=20."""
    g = golden("tiny_model.npz")
    feats = synth.video_features(2, seed=int(g["feat_seed"])).to(DEV)
    prompt = _ref(g, "greedy_T20")[:, :, 5:13]
    out = tiny_engine.generate_codes(feats, 20, prompt=prompt.to(DEV)).cpu()
    assert torch.equal(out, _ref(g, "prompt8_greedy_T20"))
    assert torch.equal(out[:, :, :8], prompt)


def test_philox_generation_is_batch_shard_invariant(tiny_engine):
    feats = synth.video_features(4, seed=8).to(DEV)
    full = tiny_engine.generate_codes(feats, 12, use_sampling=True, top_k=250, cfg_scale=6.0, seed=5).cpu()
    part = tiny_engine.generate_codes(feats[2:], 12, use_sampling=True, top_k=250, cfg_scale=6.0, seed=5, clip_base=2).cpu()
    assert torch.equal(full[2:], part)
    assert int(full.min()) >= 0 and int(full.max()) < 1024


def test_random_depth_against_live_oracle():
    """3-layer model, fresh seeds, B=3 (ragged vs the 16-row tile), T=33 with 3 video tokens
    (positions >= 21 read empty_video_emb): HIP loop == oracle loop, token for token."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=77)
    feats = synth.video_features(3, tokens=3, seed=78)
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    ref = go.generate(dec, feats, 33, mode="cached")
    eng = DecoderEngine(cfg, sd, DEV, wdtype="h1")
    got = eng.generate_codes(feats.to(DEV), 33).cpu()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("wdtype", ENGINE_KINDS)
def test_full_size_greedy_tokens_match_reference(golden, full_sampler_sd, wdtype, parity_report):
    """configs[0]-shaped case at full depth (24 layers, 694 M params), B=2, T=220, greedy:
    tokens identical to what the reference's cache-less CPU generate() produced (bf16-representable checkpoint:
    every storage holds the same numbers)."""
    g = golden("full_greedy_B2_T220.npz")
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd, DEV, wdtype=wdtype)
    feats = synth.video_features(2, seed=int(g["feat_seed"])).to(DEV)
    tok = eng.generate_codes(feats, 220).cpu()
    eng.check_status()
    assert_tokens_equal(parity_report, "full_greedy_B2_T220", wdtype, "greedy cfg 1, B=2 (bf16-representable checkpoint)", tok, _ref(g, "tokens"),
                        g["margins"])
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("wdtype", ["auto", "f32"])
def test_unrounded_checkpoint_against_live_oracle(wdtype):
    """A checkpoint whose weights 16 bits cannot hold (no rounding at synthesis = a real fp32 checkpoint's situation):
    the default storage decision is two fp16 planes ("h2"), and both that engine and the exact-fp32-MFMA one ("f32") are
    token-exact against the oracle — greedy, ragged batch, CFG + top-k sampling with a recorded noise stream, and a
    teacher-forced prompt."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=171, round_bf16=False)
    assert not all(torch.equal(sd[k], sd[k].bfloat16().float()) for k in sd if synth.is_streamed_weight(k))
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    assert eng.wdtype == ("h2" if wdtype == "auto" else "f32") and eng.requested_wdtype == wdtype
    feats = synth.video_features(3, tokens=3, seed=172)
    ref = go.generate(dec, feats, 33, mode="cached")
    assert torch.equal(eng.generate_codes(feats.to(DEV), 33).cpu(), ref)
    feats = synth.video_features(9, seed=173)                        # 18 rows with CFG: two row blocks
    nz = synth.exp_noise(24 + 9 - 1, 9 * 9, 1024, 174)
    ref = go.generate(dec, feats, 24, mode="cached", cfg_scale=6.0, use_sampling=True, top_k=250, noise=nz)
    got = eng.generate_codes(feats.to(DEV), 24, cfg_scale=6.0, use_sampling=True, top_k=250, noise=nz).cpu()
    assert torch.equal(got, ref), float((got == ref).float().mean())
    prompt = ref[:2, :, :18]                                         # 18 positions >= 16 row blocks: the prefill GEMM
    ref_p = go.generate(dec, feats[:2], 24, prompt=prompt, mode="cached")
    got_p = eng.generate_codes(feats[:2].to(DEV), 24, prompt=prompt.to(DEV)).cpu()
    assert torch.equal(got_p, ref_p)


def test_full_size_unrounded_checkpoint_matches_reference(golden, full_sampler_sd_raw, parity_report):
    """Full depth, B=2, T=220, greedy, on the UN-rounded 694 M-parameter checkpoint, against tokens the reference's
    own cache-less CPU generate() produced for that checkpoint (make_golden.py full_greedy_raw; min top-1/top-2
    margin 6.6e-5).  (1) default storage ("auto" -> two fp16 planes, 22 bits): tokens identical, first-forward logits within
    3e-5; (2) ONE plane FORCED on this checkpoint rounds 694 M weights to 11 bits: not token-exact by construction — its
    agreement and logit error are REPORTED (the session's parity report, entry storage "h1 (forced)"), with a loose sanity bound only;
    (3) the exact-fp32-MFMA engine ("f32"): tokens identical too."""
    g = golden("full_greedy_raw_B2_T220.npz")
    ref = _ref(g, "tokens")
    feats = synth.video_features(2, seed=int(g["feat_seed"])).to(DEV)
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV)      # "auto"
    assert eng.wdtype == "h2"
    tok = eng.generate_codes(feats, 220).cpu()
    eng.check_status()
    assert_tokens_equal(parity_report, "full_greedy_raw_B2_T220", "h2", "greedy cfg 1, B=2 (un-rounded checkpoint, auto)", tok, ref, g["margins"])
    # first forward of the reference run = position 0 of every row (all special tokens): its recorded logits
    step1 = torch.from_numpy(g["logits"][list(g["logits_steps"]).index(1)])            # (B, K, V)
    idx0 = torch.full((2, 9, 1), 1024, dtype=torch.long)
    lg32 = eng.logits_all_positions(idx0.to(DEV), feats)[:, :, 0].cpu()
    err32 = float((lg32 - step1).abs().max())
    assert err32 < 3e-5, err32
    parity_report.note_logit_err("full_greedy_raw_B2_T220", "h2", err32, logit_err_step=1)
    del eng
    torch.cuda.empty_cache()
    ef = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV, wdtype="f32")
    tokf = ef.generate_codes(feats, 220).cpu()
    assert_tokens_equal(parity_report, "full_greedy_raw_B2_T220", "f32", "greedy cfg 1, B=2 (un-rounded checkpoint, exact fp32 MFMA)", tokf, ref,
                        g["margins"])
    del ef
    torch.cuda.empty_cache()
    e16 = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV, wdtype="h1")
    tok16 = e16.generate_codes(feats, 220).cpu()
    lg16 = e16.logits_all_positions(idx0.to(DEV), feats)[:, :, 0].cpu()
    del e16
    torch.cuda.empty_cache()
    agree = float((tok16 == ref).float().mean())
    first_bad = int((tok16 != ref).any(dim=1).float().argmax(-1).min()) if agree < 1.0 else -1
    err16 = float((lg16 - step1).abs().max())
    rel16 = float((lg16 - step1).pow(2).mean().sqrt() / step1.pow(2).mean().sqrt())
    parity_report.add("full_greedy_raw_B2_T220", "h1 (forced: rounds the checkpoint to 11 bits, NOT a parity claim)",
                      "greedy cfg 1, B=2 (un-rounded checkpoint)", tok16, ref, g["margins"], max_logit_err=err16, logits_rel_rms_step1=rel16,
                      first_frame_with_a_different_token=first_bad)
    assert err16 < 0.1 and 0.0 < agree <= 1.0      # a rounded model is close, and it IS a different model than f32


def test_configs1_batch8_full_row_block_matches_reference(golden, full_sampler_sd, parity_report):
    """configs[1] at its real batch (B=8): features are keyed per clip, so clips 0-1 of the 8-clip run must equal the
    reference's B=2 goldens — greedy cfg 1 (8 rows) and CFG 6 / top-k 250 sampled (16 rows = one FULL row block, the
    benchmark's shape) with the reference's noise stream in the rows of clips 0-1."""
    g = golden("full_greedy_B2_T220.npz")
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd, DEV, wdtype="h1")
    feats = synth.video_features(8, seed=int(g["feat_seed"])).to(DEV)
    tok = eng.generate_codes(feats, 220).cpu()
    assert_tokens_equal(parity_report, "full_greedy_B2_T220", "h1", "greedy cfg 1, clips 0-1 of B=8 (8 rows)", tok[:2], _ref(g, "tokens"), g["margins"])
    gs = golden("full_topk250_cfg6_B2_T220.npz")
    nz2 = synth.exp_noise(228, 18, 1024, int(gs["noise_seed"]))
    other = synth.exp_noise(228, 54, 1024, 4321)
    nz8 = torch.cat([nz2, other], dim=1)                 # noise rows are (clip, codebook): clips 0-1 first
    tok = eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, noise=nz8).cpu()
    eng.check_status()
    assert_tokens_equal(parity_report, "full_topk250_cfg6_B2_T220", "h1", "cfg 6 / top-k 250 sampled, clips 0-1 of B=8 (16 rows)", tok[:2],
                        _ref(gs, "tokens"))
    assert int(tok.min()) >= 0 and int(tok.max()) < 1024


@pytest.mark.parametrize("wdtype", ["h1", "h2", "f32"])
def test_configs3_long_context_matches_reference(golden, wdtype, parity_report):
    """BASELINE configs[3] at full depth against the reference itself (make_golden.py full_c4: the reference
    Transformer built with block_size_audio=1024, Tv=128, cfg 1.0, B=1, greedy, T=880 -> 888 cache-less passes):
    tokens identical — up to the ONE step of that run whose reference margin (5.5e-6 at step 578; the next smallest of the 888 is
    6.2e-5) is inside fp32 summation-order noise: the reference's own logits move by ~3e-6 with the prefix length it re-feeds
    (SURVEY.md §7).  Run on all three engines, INCLUDING the exact-fp32-MFMA one ("f32": bit-for-bit fp32 products, only the order
    of the sums differs from torch's): whether each flips there, and how many of the 880 frames match, is recorded per storage in
    the session's parity report — a flip on "f32" says the step is decided by summation order, not by the 22-bit operand format.
    B=1 -> 16 (row, head) pairs: the range-split attention + combine pass run at every length."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_c4_greedy_B1_T880.npz")
    if not os.path.exists(path):
        pytest.skip("full_c4 golden not generated (tests/golden/make_golden.py full_c4, ~30 min)")
    g = np.load(path)
    cfg = synth.SamplerCfg(block_size_audio=int(g["block_size_audio"]))
    sd = synth.sampler_state_dict(cfg, seed=int(g["weight_seed"]), round_bf16=True)
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    feats = synth.video_features(1, tokens=128, seed=int(g["feat_seed"])).to(DEV)
    tok = eng.generate_codes(feats, 880).cpu()
    eng.check_status()
    ref = _ref(g, "tokens")
    assert eng.max_len >= 896
    # the reference run holds ONE genuinely tight step (margin 5.5e-6 at step 578; the next smallest is 6.2e-5)
    e = assert_tokens_or_recorded_near_tie(parity_report, "full_c4_greedy_B1_T880", wdtype, "configs[3]: greedy cfg 1, B=1, T=880", tok, ref,
                                           g["margins"], 2e-5)
    print(f"configs[3] [{wdtype}]: tokens_equal={e['tokens_equal']} first_diff_step={e['first_diff_step']} "
          f"identical frames {e['identical_frames_before_first_diff']}/880")
    assert e["tokens_equal"] or e["first_diff_step"] == 578, e
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("wdtype", ["h2", "h1", "f32"])
def test_configs3_at_its_stated_batch_4_matches_reference(golden, wdtype, parity_report):
    """BASELINE configs[3] at the batch it STATES (B=4, T=880, cfg 1 -> 4 decoder rows, 64 (row, head) pairs: a different split
    count in `attention_split_kernel` than B=1's 16 pairs, and the 4-live-row shapes of the one-launch MLP) at full depth.  Features
    are keyed per clip, so clip 0 of the 4-clip run must equal the reference's own B=1 run (`full_c4_greedy_B1_T880.npz`:
    models/vaura_model.py:502-547 with the rope table of llama.py:593-603 beyond 256 rows) token for token — on two planes, one plane
    and the exact-fp32 engine.  The golden's ONE literal near-tie (step 578, margin 5.5e-6) keeps its recorded-near-tie form; clips
    1-3 have no reference run: they are checked for range and for being the same on all three storages' clip-0 decision only."""
    g = golden("full_c4_greedy_B1_T880.npz")
    cfg = synth.SamplerCfg(block_size_audio=int(g["block_size_audio"]))
    sd = synth.sampler_state_dict(cfg, seed=int(g["weight_seed"]), round_bf16=True)
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    feats = synth.video_features(4, tokens=128, seed=int(g["feat_seed"])).to(DEV)
    tok = eng.generate_codes(feats, 880).cpu()
    eng.check_status()
    assert tuple(tok.shape) == (4, 9, 880) and int(tok.min()) >= 0 and int(tok.max()) < 1024
    e = assert_tokens_or_recorded_near_tie(parity_report, "full_c4_greedy_B1_T880", wdtype, "configs[3] at its stated batch: greedy cfg 1, clip 0 of B=4, T=880",
                                           tok[:1], _ref(g, "tokens"), g["margins"], 2e-5)
    print(f"configs[3] B=4 [{wdtype}]: tokens_equal={e['tokens_equal']} first_diff_step={e['first_diff_step']} "
          f"identical frames {e['identical_frames_before_first_diff']}/880")
    assert e["tokens_equal"] or e["first_diff_step"] == 578, e
    # the other three clips are independent sequences: each must differ from clip 0 (different features) — a row mix-up would not
    assert all(not torch.equal(tok[b], tok[0]) for b in (1, 2, 3))
    del eng
    torch.cuda.empty_cache()


def test_full_size_sampled_tokens_match_reference(golden, full_sampler_sd, parity_report):
    """configs[1] sampling settings (top-k 250, cfg 6.0) at B=2 with the reference's own CPU noise
    stream (seed recorded in the fixture) -> identical tokens."""
    g = golden("full_topk250_cfg6_B2_T220.npz")
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd, DEV, wdtype="h1")
    feats = synth.video_features(2, seed=int(g["feat_seed"])).to(DEV)
    nz = synth.exp_noise(228, 18, 1024, int(g["noise_seed"]))
    tok = eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, noise=nz).cpu()
    assert_tokens_equal(parity_report, "full_topk250_cfg6_B2_T220", "h1", "cfg 6 / top-k 250 sampled, B=2 (4 rows)", tok, _ref(g, "tokens"))


@pytest.mark.parametrize("precision,tol", [("f32", 5e-6), ("f16pair", 1e-4)])
def test_dac_decode_matches_oracle(precision, tol):
    """DAC decode (full-width 1536-channel decoder, 12 frames -> 6144 samples) vs the fp32 CPU
    restatement; tolerance: RMS error <= 1e-4 (north_star) for the fp16-pair path, 5e-6 for exact fp32."""
    from oracle import dac_oracle
    ccfg = synth.FULL_CODEC
    sd = synth.codec_state_dict(ccfg, seed=1)
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(0, 1024, (2, 9, 12), generator=g)
    ref = dac_oracle.decode(sd, codes, ccfg.decoder_rates)
    eng = CodecEngine(ccfg, sd, DEV, precision=precision)
    got = eng.decode(codes.to(DEV)).cpu()
    assert got.shape == ref.shape == (2, 1, 12 * 512)
    rms = float(((got - ref) ** 2).mean().sqrt())
    print(f"codec {precision}: rms err {rms:.3e}, max err {float((got - ref).abs().max()):.3e}")
    assert rms <= tol, rms
    assert float(ref.abs().max()) > 0.05  # the fixture is not a silent waveform


def test_dac_decode_plain_fp16_against_its_emulation_and_the_fp32_oracle():
    """Codec precision "f16": plain fp16 operands with fp32 accumulate, ONE matrix instruction per product — the arithmetic
    class the reference itself runs DAC in (models/vaura_model.py:92 `.half()`).  The kernel's arithmetic is pinned per layer
    (test_gpu_ops.py::test_codec_convolution_per_precision[f16]: 1e-6 of fp64 on the fp16-rounded operands, every geometry).
    End to end the model is the oracle with every convolution's weights and inputs rounded to fp16 (the last, C -> 1, conv stays
    fp32 like in the kernels) — but with synthetic Gaussian weights that emulation is itself discontinuous: a relative
    perturbation of 1e-7 of its pre-rounding activations (what a different fp32 summation order does) moves ITS output by 1.1e-4
    RMS, because an fp16 rounding flip is a 2^-11 step the random network does not damp (measured on the CPU; the same effect as
    for mx8 below).  So the end-to-end bar is: no farther from the emulation than the emulation is from the fp32 oracle; both
    distances are printed — the second is what the reference's own fp16 codec costs.  The default precision stays "f16pair"."""
    from oracle import dac_oracle
    ccfg = synth.FULL_CODEC
    sd = synth.codec_state_dict(ccfg, seed=1)
    codes = torch.randint(0, 1024, (2, 9, 12), generator=torch.Generator().manual_seed(3))
    h = lambda t: t.half().float()
    emu = dac_oracle.decode(sd, codes, ccfg.decoder_rates, act_quant=h, weight_quant=h, quant_input=True, quant_last=False)
    ref = dac_oracle.decode(sd, codes, ccfg.decoder_rates)
    got = CodecEngine(ccfg, sd, DEV, precision="f16").decode(codes.to(DEV)).cpu()
    sig = float((ref ** 2).mean().sqrt())
    rms_emu = float(((got - emu) ** 2).mean().sqrt())
    rms_ref = float(((got - ref) ** 2).mean().sqrt())
    rms_model = float(((emu - ref) ** 2).mean().sqrt())
    print(f"codec f16: rms vs its fp16-operand emulation {rms_emu:.3e}; vs the fp32 oracle {rms_ref:.3e} (emulation vs fp32 "
          f"{rms_model:.3e}; signal rms {sig:.3e})")
    assert torch.isfinite(got).all()
    assert rms_emu <= rms_model, (rms_emu, rms_model)
    assert rms_ref <= 1.5 * rms_model and rms_ref <= 5e-3 * sig, (rms_ref, rms_model, sig)


@pytest.mark.parametrize("B", [10, 16, 20])
def test_two_row_blocks_with_cfg_against_live_oracle(B):
    """B=10 with CFG -> 20 decoder rows = two 16-row blocks (the second one ragged), B=16 -> 32 rows (two full blocks: the
    reference's default batch, configs/generate_vgg.yaml:41; both on the two-row-block GEMVs and the two-row-block one-launch MLP),
    B=20 -> 40 rows (three blocks: a full two-block pass + a half-empty one, no one-launch MLP): attention / sampler grids grow.
    Token-exact vs the oracle, greedy and top-k-128 sampled with a recorded noise stream."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.tiny_sampler(2)
    sd = synth.sampler_state_dict(cfg, seed=31)
    feats = synth.video_features(B, seed=32)
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    ref = go.generate(dec, feats, 14, mode="cached", cfg_scale=6.0)
    nz = synth.exp_noise(14 + 9 - 1, B * 9, 1024, 33)
    refs = go.generate(dec, feats, 14, mode="cached", cfg_scale=6.0, use_sampling=True, top_k=128, noise=nz)
    for wd in ("f32", "h1", "h2"):
        eng = DecoderEngine(cfg, sd, DEV, wdtype=wd)
        got = eng.generate_codes(feats.to(DEV), 14, cfg_scale=6.0).cpu()
        assert torch.equal(got, ref), wd
        gots = eng.generate_codes(feats.to(DEV), 14, cfg_scale=6.0, use_sampling=True, top_k=128, noise=nz).cpu()
        eng.check_status()
        assert torch.equal(gots, refs), wd


def test_long_context_single_pass_against_live_oracle():
    """BASELINE configs[3] shape at reduced depth: block_size_audio=1024 (rope table beyond 256 rows),
    Tv=128 video tokens, T=300 (> 256 cached positions: the generic attention loop), B=2, cfg 1.0
    (the reference's CFG null embedding is fixed at 32 tokens, vaura_model.py:790-793)."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.tiny_sampler(2, block_size_audio=1024)
    sd = synth.sampler_state_dict(cfg, seed=41)
    feats = synth.video_features(2, tokens=128, seed=42)
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead, block_size=1024)
    ref = go.generate(dec, feats, 300, mode="cached")
    eng = DecoderEngine(cfg, sd, DEV, wdtype="h1")
    got = eng.generate_codes(feats.to(DEV), 300).cpu()
    assert eng.max_len >= 1024
    assert torch.equal(got, ref), float((got == ref).float().mean())
    # the range-split attention merges its partials inside the launch (the last split of a (row, head) to arrive); debug flag bit 19 keeps
    # the merge as its own launch: same sums in the same order, so the very same tokens, eager and through the captured graph
    from vaura_amd import _lib as L
    try:
        L.lib().vaura_set_debug_flags(1 << 19)
        eng._free_graph()
        two = eng.generate_codes(feats.to(DEV), 300).cpu()
        two_eager = eng.generate_codes(feats.to(DEV), 300, use_graph=False).cpu()
    finally:
        L.lib().vaura_set_debug_flags(0)
        eng._free_graph()
    assert torch.equal(two, got) and torch.equal(two_eager, got)
    assert torch.equal(eng.generate_codes(feats.to(DEV), 300, use_graph=False).cpu(), got)


@pytest.mark.parametrize("B,cfg_scale,pass_positions", [(2, 1.0, 32), (3, 6.0, 32), (10, 6.0, 32), (3, 6.0, 192), (2, 1.0, 8)])
def test_batched_prompt_prefill_against_live_oracle(B, cfg_scale, pass_positions, monkeypatch):
    """Sliding-window shape (scripts/generate.py:327-365): a 40-token prompt of a 60-token chunk is teacher-forced
    in passes of `pass_positions` (two GEMM passes of <= 32, one pass, or 5 passes of 8 = fewer than 16 row blocks:
    the register-resident GEMV loop), then 28 positions are generated.  Rows = B or 2B (incl. two ragged row
    blocks).  Token-exact."""
    monkeypatch.setattr(DecoderEngine, "PREFILL_POSITIONS", pass_positions)
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.tiny_sampler(2)
    sd = synth.sampler_state_dict(cfg, seed=51)
    feats = synth.video_features(B, seed=52)
    prompt = torch.randint(0, 1024, (B, 9, 40), generator=torch.Generator().manual_seed(53))
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    ref = go.generate(dec, feats, 60, prompt=prompt, mode="cached", cfg_scale=cfg_scale)
    eng = DecoderEngine(cfg, sd, DEV, wdtype="h1")
    got = eng.generate_codes(feats.to(DEV), 60, prompt=prompt.to(DEV), cfg_scale=cfg_scale).cpu()
    assert torch.equal(got[:, :, :40], prompt)
    assert torch.equal(got, ref), float((got == ref).float().mean())


def test_fp8_weights_against_live_oracle_on_dequantised_checkpoint():
    """BASELINE configs[4] storage (no reference counterpart): the fp8 engine must generate exactly what the
    oracle generates for the checkpoint whose per-layer matrices are the dequantised fp8 values — greedy, CFG,
    two row blocks, plus a prompt (batched prefill uses the same fp8 kernels)."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    from vaura_amd import quant
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=61)   # heads / conditioning bf16-exact; layer matrices get quantised
    sd_eff = quant.fp8_effective_state_dict(sd)
    assert any(not torch.equal(sd[k], sd_eff[k]) for k in sd)
    dec = DecoderOracle(sd_eff, cfg.num_layers, cfg.nhead)
    eng = DecoderEngine(cfg, sd, DEV, wdtype="fp8")
    feats = synth.video_features(9, seed=62)
    ref = go.generate(dec, feats, 24, mode="cached", cfg_scale=6.0)
    got = eng.generate_codes(feats.to(DEV), 24, cfg_scale=6.0).cpu()
    assert torch.equal(got, ref), float((got == ref).float().mean())
    prompt = ref[:2, :, :18]      # 18 positions x 1 row block >= 16: the fp8 prefill GEMM
    ref_p = go.generate(dec, feats[:2], 24, prompt=prompt, mode="cached")
    got_p = eng.generate_codes(feats[:2].to(DEV), 24, prompt=prompt.to(DEV)).cpu()
    assert torch.equal(got_p, ref_p)


def test_fp8_full_size_agreement_with_bf16(full_sampler_sd):
    """Full-depth model: (1) the fp8 engine against the bf16 engine run on the dequantised checkpoint (same real-number
    weights, different storage and kernel instances, hence different fp32 summation order): teacher-forced logits
    within 2e-5, greedy tokens identical; (2) agreement with the unquantised model is REPORTED (fp8 changes the model;
    no reference number exists for it): logits error, top-1 agreement, sampled-token agreement with the same Philox
    noise at configs[1] sampling settings."""
    from vaura_amd import quant
    cfg = synth.FULL_SAMPLER
    feats = synth.video_features(2, seed=5).to(DEV)
    kw = dict(use_sampling=True, top_k=250, cfg_scale=6.0, seed=7)
    e8 = DecoderEngine(cfg, full_sampler_sd, DEV, wdtype="fp8")
    tok8 = e8.generate_codes(feats, 220, **kw).cpu()
    greedy8 = e8.generate_codes(feats, 220, cfg_scale=6.0).cpu()
    idx = tok8[:, :, :40].contiguous()
    lg8 = e8.logits_all_positions(idx.to(DEV), feats).cpu()
    del e8
    torch.cuda.empty_cache()
    e_eff = DecoderEngine(cfg, quant.fp8_effective_state_dict(full_sampler_sd), DEV, wdtype="h1")
    greedy_eff = e_eff.generate_codes(feats, 220, cfg_scale=6.0).cpu()
    lg_eff = e_eff.logits_all_positions(idx.to(DEV), feats).cpu()
    del e_eff
    torch.cuda.empty_cache()
    assert float((lg8 - lg_eff).abs().max()) < 2e-5, float((lg8 - lg_eff).abs().max())
    assert torch.equal(greedy8, greedy_eff), float((greedy8 == greedy_eff).float().mean())
    e16 = DecoderEngine(cfg, full_sampler_sd, DEV, wdtype="h1")
    tok16 = e16.generate_codes(feats, 220, **kw).cpu()
    lg16 = e16.logits_all_positions(idx.to(DEV), feats).cpu()
    del e16
    torch.cuda.empty_cache()
    rel = float((lg8 - lg16).pow(2).mean().sqrt() / lg16.pow(2).mean().sqrt())
    top1 = float((lg8.argmax(-1) == lg16.argmax(-1)).float().mean())
    agree = float((tok8 == tok16).float().mean())
    print(f"fp8 vs bf16 (synthetic checkpoint): logits rel-RMS {rel:.3e}, top-1 agreement {top1:.4f}, "
          f"sampled-token agreement over 220 frames {agree:.4f}")
    assert rel < 0.15 and top1 > 0.5


def test_dac_encode_matches_oracle():
    """Row f4: DacModelWrapper.encode (models/modules/dac/model.py:30-39) at full width (64 -> 1024 channels, strides
    2,4,8,8, 9-stage residual VQ) against the fp32 CPU restatement.  Codes are integers but come out of an argmax over
    fp32 distances: a code may differ only where the oracle's best and second-best distances are closer than the conv
    arithmetic's error (fp16 pairs vs fp32, different summation order) — and every later stage of that frame then
    sees a different residual, so frames are compared up to their first such near-tie."""
    from oracle import dac_oracle
    ccfg = synth.FULL_CODEC
    sd = dict(synth.codec_state_dict(ccfg, seed=1))
    sd.update(synth.codec_encoder_state_dict(ccfg, seed=1))
    g = torch.Generator().manual_seed(4)
    n = 512 * 10 - 100                                   # not a multiple of the hop: preprocess pads
    t = torch.arange(n) / 44100.0
    wav = (0.4 * torch.sin(2 * torch.pi * 330.0 * t) + 0.15 * torch.randn(2, 1, n, generator=g))
    z = dac_oracle.encode_latent(sd, dac_oracle.preprocess(wav, 512), ccfg.encoder_rates)
    ref, margin = dac_oracle.quantize(sd, z, ccfg.n_codebooks, return_margin=True)
    from vaura_amd.engine import CodecEncoderEngine
    eng = CodecEncoderEngine(ccfg, sd, DEV)
    got = eng.encode(wav.to(DEV)).cpu()
    assert got.shape == ref.shape == (2, 9, 10)
    assert int(got.min()) >= 0 and int(got.max()) < 1024
    agree = float((got == ref).float().mean())
    first_bad = (got != ref).float().cumsum(1) > 0            # stages at or after the first mismatch of a frame
    clean = ~first_bad
    print(f"dac encode: code agreement {agree:.4f}; min oracle margin {float(margin.min()):.2e}")
    for b, k, tt in torch.nonzero(got != ref).tolist():
        if k == 0 or bool(clean[b, k - 1, tt]):               # a first mismatch must sit on a near-tie
            assert float(margin[b, k, tt]) < 5e-4, (b, k, tt, float(margin[b, k, tt]))
    assert agree > 0.9
    # latent check through the first stage: stage-0 codes only depend on the encoder output
    assert float((got[:, 0] == ref[:, 0]).float().mean()) > 0.95


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("f16pair", 2e-5), ("f16", 2e-3)])
def test_dac_decode_directly_against_the_hf_fixture(golden, precision, tol, name="codec_hf_full.npz", full=True):
    """`vaura_dac_decode` against waveforms an INDEPENDENT implementation of DAC-44k produced (transformers' `DacModel`, weights
    folded from the same synthetic DAC-1.0.0-keyed state dict: tests/golden/make_golden.py codec / codec_full), with no hop through
    oracle/dac_oracle.py.  Not the reference's own dependency (descript-audio-codec 1.0.0 is absent offline: a16 stays
    parity-unpinned), but it removes the oracle from between the HIP codec and the only external vectors this image can make.
    The 44.1 kHz model's full width (the HIP kernels' tile shapes are built for it; the reduced-width fixture is the CPU suite's)."""
    g = golden(name)
    ccfg = synth.FULL_CODEC if full else synth.CodecCfg(decoder_dim=int(g["decoder_dim"]), decoder_rates=(8, 8, 4, 2))
    sd = synth.codec_state_dict(ccfg, seed=int(g["codec_seed"]))
    codes = torch.from_numpy(g["codes"].astype(np.int64))
    ref = torch.from_numpy(g["wav"])
    wav = CodecEngine(ccfg, sd, DEV, precision=precision).decode(codes.to(DEV)).cpu()
    assert wav.shape == ref.shape and float(ref.abs().max()) > 0.05
    err, rms = float((wav - ref).abs().max()), float(((wav - ref) ** 2).mean().sqrt())
    print(f"HIP DAC decode [{precision}] vs HF DacModel ({name}): max abs {err:.3e}, rms {rms:.3e}")
    assert err < tol and rms <= (1e-4 if precision != "f16" else 1e-3), (err, rms)


def test_dac_encode_directly_against_the_hf_fixture(golden):
    """`vaura_dac_encode` at full width (64 -> 1024 channels, 9-stage residual VQ) against the codes transformers' `DacModel`
    produced for the same waveform and weights (`codec_enc_hf_full.npz`).  Codes are an argmin over fp32 distances: a frame may
    leave HF's codes only at a stage where HF's OWN latent has its two nearest codewords closer than the conv arithmetic's error
    (margin computed on the fixture's `z`, not on anything the HIP path produced)."""
    from oracle import dac_oracle
    from vaura_amd.engine import CodecEncoderEngine
    g = golden("codec_enc_hf_full.npz")
    ccfg = synth.FULL_CODEC
    sd = dict(synth.codec_state_dict(ccfg, seed=int(g["codec_seed"])))
    sd.update(synth.codec_encoder_state_dict(ccfg, seed=int(g["codec_seed"])))
    wav = torch.from_numpy(g["wav"])
    ref = torch.from_numpy(g["codes"].astype(np.int64))
    got = CodecEncoderEngine(ccfg, sd, DEV).encode(wav.to(DEV)).cpu()
    assert got.shape == ref.shape
    _, margin = dac_oracle.quantize(sd, torch.from_numpy(g["z"]), ccfg.n_codebooks, return_margin=True)
    bad = got != ref
    clean = ~(bad.float().cumsum(1) > 0)
    nfirst = 0
    for b, k, t in torch.nonzero(bad).tolist():
        if k == 0 or bool(clean[b, k - 1, t]):
            assert float(margin[b, k, t]) < 1e-3, (b, k, t, float(margin[b, k, t]))
            nfirst += 1
    agree = float((~bad).float().mean())
    print(f"HIP DAC encode vs HF DacModel codes: agreement {agree:.4f}, {nfirst} frames leave at a near-tie of HF's own latent")
    assert agree > 0.95


def test_codec_round_trip_through_the_plugin():
    """encode(decode(codes)) through vaura_amd.codec.DacModelWrapper: shapes / dtypes / ranges of the reference's
    wrapper (models/modules/dac/model.py:30-48); with random weights the round trip is not the identity."""
    import warnings
    from vaura_amd.codec import DacModelWrapper
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DacModelWrapper(model_sr=44100, synthetic=True).to(DEV)
    codes = torch.randint(0, 1024, (2, 9, 7), generator=torch.Generator().manual_seed(2)).to(DEV)
    wav = m.decode([(codes, None)])
    assert wav.shape == (2, 1, 7 * 512)
    back = m.encode(wav)
    assert back.shape == (2, 9, 7) and back.dtype == torch.int64 and int(back.min()) >= 0 and int(back.max()) < 1024
    one = m(wav[0, 0])                                        # forward == encode; 1-D input is unsqueezed twice
    assert torch.equal(one, back[:1])


def test_two_engines_interleaved_keep_their_own_step_graphs(tiny_sampler_sd):
    """Each engine owns the hipGraph it captured (vaura_step_graph_build returns a handle): A, B, A again must
    not replay B's graph against A's buffers."""
    a = DecoderEngine(synth.tiny_sampler(2), tiny_sampler_sd, DEV, wdtype="h1")
    b = DecoderEngine(synth.tiny_sampler(2), synth.sampler_state_dict(synth.tiny_sampler(2), seed=9), DEV, wdtype="h1")
    fa, fb = synth.video_features(2, seed=1).to(DEV), synth.video_features(3, seed=2).to(DEV)
    a1 = a.generate_codes(fa, 16, cfg_scale=6.0).cpu()
    b1 = b.generate_codes(fb, 16).cpu()
    a2 = a.generate_codes(fa, 16, cfg_scale=6.0).cpu()
    b2 = b.generate_codes(fb, 16).cpu()
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a1[:, :, :8], b1[:2, :, :8])
    eager = a.generate_codes(fa, 16, cfg_scale=6.0, use_graph=False).cpu()
    assert torch.equal(a1, eager)


def test_dac_decode_fp8_weights_against_oracle_on_dequantised_checkpoint():
    """BASELINE configs[4], codec part: conv weights in fp8 (e4m3, power-of-two scale per output channel) are exact in one
    fp16 plane, so the kernel skips the lo-plane product.  Same bar as the decoder's fp8 storage: the result must be the
    oracle's decode of the DEQUANTISED checkpoint (RMS <= 1e-4); the distance to the unquantised codec is reported."""
    from oracle import dac_oracle
    from vaura_amd import quant
    ccfg = synth.FULL_CODEC
    sd = synth.codec_state_dict(ccfg, seed=1)
    codes = torch.randint(0, 1024, (2, 9, 12), generator=torch.Generator().manual_seed(3))
    ref_q = dac_oracle.decode(quant.fp8_effective_codec_state_dict(sd), codes, ccfg.decoder_rates)
    ref = dac_oracle.decode(sd, codes, ccfg.decoder_rates)
    got = CodecEngine(ccfg, sd, DEV, precision="f16pair_w8").decode(codes.to(DEV)).cpu()
    rms = float(((got - ref_q) ** 2).mean().sqrt())
    rms_model = float(((ref_q - ref) ** 2).mean().sqrt())
    print(f"codec fp8 weights: rms err vs oracle on the dequantised weights {rms:.3e}; fp8 vs unquantised codec rms {rms_model:.3e} "
          f"(signal rms {float((ref ** 2).mean().sqrt()):.3e})")
    assert rms <= 1e-4, rms
    assert rms_model > 1e-4        # it IS a different model


@pytest.mark.parametrize("precision", ["f16pair", "f16pair_w8", "f16"])
def test_dac_decode_256_row_workgroups_are_bit_identical(precision):
    """The codec's convs run in 256-row workgroups wherever enough of them remain (csrc/dac.hip, conv_pair_kernel<..., MJ = 8>):
    only at full length, which the oracle-checked cases above (12 frames) never reach.  Same products summed in the same order,
    so the waveform of a full-length decode must equal, bit for bit, the one from 128-row workgroups (debug flag bit 20) — the
    instances the oracle pins.  Also the encoder's 64-column instances (codes identical)."""
    from vaura_amd import _lib as L
    from vaura_amd.engine import CodecEncoderEngine
    ccfg = synth.FULL_CODEC
    sd = dict(synth.codec_state_dict(ccfg, seed=1))
    sd.update(synth.codec_encoder_state_dict(ccfg, seed=1))
    codes = torch.randint(0, 1024, (4, 9, 220), generator=torch.Generator().manual_seed(5)).to(DEV)
    dec = CodecEngine(ccfg, sd, DEV, precision=precision)
    enc = CodecEncoderEngine(ccfg, sd, DEV) if precision == "f16pair" else None
    out = {}
    try:
        for flags in (0, 1 << 20):
            L.lib().vaura_set_debug_flags(flags)
            wav = dec.decode(codes)
            out[flags] = (wav.clone(), enc.encode(wav).clone() if enc else None)
    finally:
        L.lib().vaura_set_debug_flags(0)
    torch.cuda.synchronize()
    assert torch.isfinite(out[0][0]).all()
    assert torch.equal(out[0][0], out[1 << 20][0])
    if enc:
        assert torch.equal(out[0][1], out[1 << 20][1])


def test_dac_decode_block_scaled_fp8_against_its_emulation():
    """BASELINE configs[4], codec part on the fp8 matrix instruction (codec precision 3, csrc/dac.hip::conv_mx8_kernel): e4m3
    weights (per-output-channel power-of-two scale) AND e4m3 activations (one power-of-two scale per 32 channels of a row,
    quantised by the producing kernel).  The arithmetic of the kernel is pinned layer by layer in
    test_gpu_ops.py::test_codec_convolution_per_precision (5e-5 of the exact result on the same quantised numbers).  End to end
    the model is the CPU emulation — the oracle's decode of the dequantised checkpoint with every Snake output passed through
    ``quant.mx8_effective_activation`` — but with synthetic Gaussian weights that emulation is itself discontinuous: a relative
    perturbation of 1e-7 of its activations (fp32 sum order) moves ITS output by 4e-3 RMS and one of 2e-5 (the fp8 MFMA's
    accumulation error) by 1.3e-2, because an e4m3 rounding flip is a 6 % step that the random network does not damp.  So the
    end-to-end bar is: finite, and no farther from the emulation than the emulation is from the unquantised codec; both
    distances are printed (the second is the "tol vs bf16" configs[4] asks for)."""
    from oracle import dac_oracle
    from vaura_amd import quant
    ccfg = synth.FULL_CODEC
    sd = synth.codec_state_dict(ccfg, seed=1)
    codes = torch.randint(0, 1024, (2, 9, 12), generator=torch.Generator().manual_seed(3))
    sd_q = quant.fp8_effective_codec_state_dict(sd)
    emu = dac_oracle.decode(sd_q, codes, ccfg.decoder_rates, act_quant=quant.mx8_effective_activation)
    ref = dac_oracle.decode(sd, codes, ccfg.decoder_rates)
    got = CodecEngine(ccfg, sd, DEV, precision="mx8").decode(codes.to(DEV)).cpu()
    rms = float(((got - emu) ** 2).mean().sqrt())
    rms_model = float(((emu - ref) ** 2).mean().sqrt())
    rms_ref = float(((got - ref) ** 2).mean().sqrt())
    sig = float((ref ** 2).mean().sqrt())
    print(f"codec mx8: rms vs its emulation {rms:.3e}; emulation vs unquantised codec {rms_model:.3e}; kernel vs unquantised codec "
          f"{rms_ref:.3e} (signal rms {sig:.3e})")
    assert torch.isfinite(got).all()
    assert rms <= rms_model, (rms, rms_model)
    assert rms_ref <= 1.5 * rms_model, (rms_ref, rms_model)      # the same model error, not a broken decode
    assert rms_model > 1e-4


def test_configs4_per_gpu_shape_fp8_weights_and_mx8_codec(full_sampler_sd):
    """BASELINE configs[4] at its per-GPU shape: 16 clips = 32 decoder rows (two row blocks), full depth, fp8 weights for
    QKV / MLP, block-scaled fp8 codec.  (1) Tokens: the fp8 engine IS the bf16 engine on the dequantised checkpoint (same real
    numbers, other kernel instances): greedy + CFG-6 tokens of all 16 clips must be identical; a clip may differ only from a
    step on where the bf16 engine's own CFG-mixed top-1 / top-2 margin is below 5e-4 (the two engines' logits differ by
    <= 2e-5 in summation order, CFG multiplies that by up to 11).  (2) configs[1] sampling on the same shape runs and stays in
    range.  (3) "tol vs bf16 reported": the mx8 codec's waveform on those tokens against the default (fp16-pair) codec —
    printed, and bounded by 0.25 x the signal's RMS (synthetic Gaussian weights: 2e-2 on a 0.12 RMS signal)."""
    from vaura_amd import quant
    cfg = synth.FULL_SAMPLER
    B = 16
    feats = synth.video_features(B, seed=15).to(DEV)
    e8 = DecoderEngine(cfg, full_sampler_sd, DEV, wdtype="fp8")
    g8 = e8.generate_codes(feats, 220, cfg_scale=6.0).cpu()
    assert e8.rows == 32
    s8 = e8.generate_codes(feats, 220, use_sampling=True, top_k=250, cfg_scale=6.0, seed=11).cpu()
    assert int(s8.min()) >= 0 and int(s8.max()) < 1024
    del e8
    torch.cuda.empty_cache()
    e_eff = DecoderEngine(cfg, quant.fp8_effective_state_dict(full_sampler_sd), DEV, wdtype="h1")
    g_eff = e_eff.generate_codes(feats, 220, cfg_scale=6.0).cpu()
    seq_eff = e_eff.seq.clone()
    K = g8.shape[1]
    n_diff = 0
    for b in range(B):
        if torch.equal(g8[b], g_eff[b]):
            continue
        n_diff += 1
        bad = g8[b] != g_eff[b]
        steps = torch.arange(220)[None, :] + 1 + torch.arange(K)[:, None]
        s = int(steps[bad].min())                          # first differing sequence position
        idx = seq_eff[b:b + 1, :, :s].to(torch.int64).repeat(2, 1, 1)
        f2 = torch.stack([feats[b], e_eff.uncond.view(feats.shape[1], -1)])
        lg = e_eff.logits_all_positions(idx, f2)[:, :, s - 1].cpu()      # (2, K, V): cond row, null row
        mixed = lg[1] + (lg[0] - lg[1]) * 6.0
        top2 = mixed.topk(2, dim=-1).values
        for k in range(K):
            if bool((bad & (steps == s))[k].any()):
                m = float(top2[k, 0] - top2[k, 1])
                assert m < 5e-4, f"clip {b}: first mismatch at step {s}, codebook {k}, bf16 engine's margin {m:.3e}"
        assert torch.equal(g8[b][steps < s], g_eff[b][steps < s])
    print(f"configs[4] shape: {B - n_diff}/{B} clips token-identical to the bf16 engine on the dequantised checkpoint "
          f"({n_diff} differ from a proven near-tie on)")
    assert n_diff <= 2
    del e_eff
    torch.cuda.empty_cache()
    ccfg = synth.FULL_CODEC
    csd = synth.codec_state_dict(ccfg, seed=0)
    wav_ref = CodecEngine(ccfg, csd, DEV, precision="f16pair").decode(s8.to(DEV)).cpu()
    wav_mx8 = CodecEngine(ccfg, csd, DEV, precision="mx8").decode(s8.to(DEV)).cpu()
    sig = float((wav_ref ** 2).mean().sqrt())
    rms = float(((wav_mx8 - wav_ref) ** 2).mean().sqrt())
    print(f"configs[4] codec: mx8 vs fp16-pair waveform rms {rms:.3e} on a {sig:.3e} rms signal ({rms / sig:.3f} of it)")
    assert torch.isfinite(wav_mx8).all() and rms <= 0.25 * sig, (rms, sig)


def test_fp8h_hi_plane_activations_tolerance_against_fp8_and_bf16(full_sampler_sd):
    """Round 6, configs[4]'s measured storage "fp8h": the fp8 matrices multiplied against the HI activation plane only (11-bit
    activations; csrc/gemv3_kernel.h WT = 3) — a narrower arithmetic than "fp8" (which is the one-plane engine on the dequantised
    checkpoint, token-exact against the oracle there), so its tolerance is REPORTED, as configs[4] demands ("tol vs bf16 reported"),
    and bounded loosely.  Full depth, configs[4]'s per-GPU shape (16 clips, cfg 6 -> 32 rows): (1) teacher-forced logits of "fp8h"
    against "fp8" (same weights: what the dropped lo plane costs) and against "h1" (the bf16 model: what fp8 weights cost — the
    dominant term by two orders of magnitude); (2) greedy cfg-6 token agreement with "fp8" over 220 frames; (3) in range, status clean."""
    cfg = synth.FULL_SAMPLER
    B = 16
    feats = synth.video_features(B, seed=15).to(DEV)
    out = {}
    for wd in ("fp8h", "fp8", "h1", "fp8h+kv16", "fp8h+kv8"):
        eng = DecoderEngine(cfg, full_sampler_sd, DEV, wdtype=wd.split("+")[0], kv_dtype={"kv16": "f16", "kv8": "f8"}.get(wd.split("+")[-1], "f32"))
        tok = eng.generate_codes(feats, 220, cfg_scale=6.0).cpu()
        eng.check_status()
        if wd == "fp8h":
            assert eng.rows == 32 and int(tok.min()) >= 0 and int(tok.max()) < 1024
            idx = tok[:2, :, :40].contiguous().to(DEV)
        lg = eng.logits_all_positions(idx, feats[:2]).cpu()
        out[wd] = (tok, lg)
        del eng
        torch.cuda.empty_cache()
    def rel(a, b):
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    r_h8 = rel(out["fp8h"][1], out["fp8"][1])
    r_h16 = rel(out["fp8h"][1], out["h1"][1])
    r_816 = rel(out["fp8"][1], out["h1"][1])
    mx = float((out["fp8h"][1] - out["fp8"][1]).abs().max())
    top1 = float((out["fp8h"][1].argmax(-1) == out["fp8"][1].argmax(-1)).float().mean())
    agree = float((out["fp8h"][0] == out["fp8"][0]).float().mean())
    steps = torch.arange(220)[None, :] + 1 + torch.arange(9)[:, None]
    first = [int(steps[out["fp8h"][0][b] != out["fp8"][0][b]].min()) if not torch.equal(out["fp8h"][0][b], out["fp8"][0][b]) else 229 for b in range(B)]
    print(f"fp8h vs fp8 (same weights, hi activation plane only): logits rel-RMS {r_h8:.3e}, max abs {mx:.3e}, top-1 agreement {top1:.4f}; "
          f"greedy cfg-6 tokens over 220 frames: agreement {agree:.4f}, clips identical {sum(f == 229 for f in first)}/{B}, earliest "
          f"first difference at step {min(first)}; vs the bf16 model: fp8h {r_h16:.3e}, fp8 {r_816:.3e}")
    assert r_h8 < 0.02 * r_816 + 2e-3 and r_h16 < 1.05 * r_816 + 1e-3      # the lo plane is noise under the fp8 weights' own error
    assert top1 > 0.97
    # configs[4]'s measured configuration at FULL depth: the fp16 K/V cache on top (and the e4m3 option), against fp8h with the fp32 cache
    r_kv16, r_kv8 = rel(out["fp8h+kv16"][1], out["fp8h"][1]), rel(out["fp8h+kv8"][1], out["fp8h"][1])
    print(f"fp8h + fp16 K/V vs fp8h (fp32 K/V), full depth: logits rel-RMS {r_kv16:.3e}; e4m3 K/V: {r_kv8:.3e} (the fp8 weights themselves: {r_816:.3e})")
    assert r_kv16 < 5e-3 and r_kv8 < 0.8 * r_816
    for k in ("fp8h+kv16", "fp8h+kv8"):
        assert int(out[k][0].min()) >= 0 and int(out[k][0].max()) < 1024


def test_configs4_full_depth_first_frames_against_the_oracle_on_the_dequantised_checkpoint(full_sampler_sd):
    """The full-depth fp8 claim against the ORACLE, not only HIP-vs-HIP: configs[4]'s per-GPU shape (16 clips, cfg 6 -> 32 rows,
    24 layers, e4m3 weights) greedy, against the CPU oracle run on the dequantised checkpoint (`quant.fp8_effective_state_dict`:
    the real numbers the fp8 storage holds) for the first 12 frames of all 16 clips (the oracle generates T=21: frames <= 12 do
    not depend on T, the form the CPU suite uses for the 24-layer goldens).  Strict; a difference is admitted only on a step where
    the ORACLE's own CFG-mixed top-1 / top-2 margin is below 5e-4 (none observed)."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    from vaura_amd import quant
    cfg = synth.FULL_SAMPLER
    B, F = 16, 12
    feats = synth.video_features(B, seed=15)
    e8 = DecoderEngine(cfg, full_sampler_sd, DEV, wdtype="fp8")
    g8 = e8.generate_codes(feats.to(DEV), 220, cfg_scale=6.0).cpu()
    e8.check_status()
    assert e8.rows == 32
    del e8
    torch.cuda.empty_cache()
    sd_eff = quant.fp8_effective_state_dict(full_sampler_sd)
    dec = DecoderOracle(sd_eff, cfg.num_layers, cfg.nhead)
    trace = {}
    ref = go.generate(dec, feats, F + 9, mode="cached", cfg_scale=6.0, trace=trace)
    K = ref.shape[1]
    steps = torch.arange(F)[None, :] + 1 + torch.arange(K)[:, None]
    n_diff = 0
    for b in range(B):
        bad = g8[b, :, :F] != ref[b, :, :F]
        if not bool(bad.any()):
            continue
        n_diff += 1
        s = int(steps[bad].min())
        top2 = trace["logits"][s][b].topk(2, dim=-1).values          # the oracle's CFG-mixed logits that decided step s
        for k in range(K):
            if bool((bad & (steps == s))[k].any()):
                m = float(top2[k, 0] - top2[k, 1])
                assert m < 5e-4, f"clip {b}: first mismatch at step {s}, codebook {k}, the oracle's margin there {m:.3e}"
    print(f"configs[4] shape at full depth vs the oracle on the dequantised checkpoint: {B - n_diff}/{B} clips identical over the "
          f"first {F} frames")
    assert n_diff == 0 or n_diff <= 1


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the exact configuration the headline `value` runs — un-rounded checkpoint -> "auto" -> two fp16 planes (h2), B=8,
# cfg 6 (16 rows: one full row block = the bench's launch shapes), top-k 250 sampled — against the reference itself.
def test_headline_configuration_h2_cfg6_topk250_B8_matches_reference(golden, full_sampler_sd_raw, parity_report):
    """BENCH `value`'s exact arithmetic and launch shapes at full depth: DecoderEngine(..., "auto") on the UN-rounded checkpoint
    resolves to h2 (two fp16 planes), B=8 with cfg 6 -> 16 decoder rows, top-k 250 sampled.  Clips 0-1 (features and noise rows
    are keyed per clip) must reproduce what the reference's own cache-less CPU generate() produced for that checkpoint, cfg 6 and
    noise stream (make_golden.py full_sample_raw, models/vaura_model.py:775-827, configs/generate_vgg.yaml:23-27) — token for
    token; and the same for greedy decoding under cfg 6 (full_greedy_cfg6_raw), where CFG multiplies the format's logit error."""
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV)
    assert eng.wdtype == "h2" and eng.requested_wdtype == "auto"
    gs = golden("full_topk250_cfg6_raw_B2_T220.npz")
    feats = synth.video_features(8, seed=int(gs["feat_seed"])).to(DEV)
    nz2 = synth.exp_noise(228, 18, 1024, int(gs["noise_seed"]))
    nz8 = torch.cat([nz2, synth.exp_noise(228, 54, 1024, 4321)], dim=1)          # noise rows are (clip, codebook): clips 0-1 first
    tok = eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=int(gs["top_k"]), cfg_scale=float(gs["cfg_scale"]),
                             noise=nz8).cpu()
    assert eng.rows == 16
    eng.check_status()
    assert int(tok.min()) >= 0 and int(tok.max()) < 1024
    assert_tokens_equal(parity_report, "full_topk250_cfg6_raw_B2_T220", "h2", "HEADLINE: cfg 6 / top-k 250 sampled, clips 0-1 of B=8 (16 rows, auto)",
                        tok[:2], _ref(gs, "tokens"), gs["margins"], gs["threshold_rel_gap"])
    # CFG-mixed logits of the first forward against the reference's (cond; null rows recorded): error of the mix itself
    lg_ref = torch.from_numpy(gs["logits"][list(gs["logits_steps"]).index(1)])                    # (2B, K, V): [cond; null]
    idx0 = torch.full((4, 9, 1), 1024, dtype=torch.long)
    f2 = torch.cat([feats[:2], (torch.zeros_like(feats[:2]) + eng.uncond)], 0)
    lg = eng.logits_all_positions(idx0.to(DEV), f2)[:, :, 0].cpu()
    err = float((lg - lg_ref).abs().max())
    mix = lambda x: x[2:] + (x[:2] - x[2:]) * 6.0
    err_mix = float((mix(lg) - mix(lg_ref)).abs().max())
    print(f"first-forward logits: max-abs error {err:.3e}, after the cfg-6 mix {err_mix:.3e}")
    assert err < 3e-5 and err_mix < 3e-4
    parity_report.note_logit_err("full_topk250_cfg6_raw_B2_T220", "h2", err, max_abs_logit_err_after_cfg6_mix=err_mix, logit_err_step=1)
    gg = golden("full_greedy_cfg6_raw_B2_T220.npz")
    tokg = eng.generate_codes(feats, 220, cfg_scale=float(gg["cfg_scale"])).cpu()
    eng.check_status()
    assert_tokens_equal(parity_report, "full_greedy_cfg6_raw_B2_T220", "h2", "greedy cfg 6, clips 0-1 of B=8 (16 rows, auto)", tokg[:2],
                        _ref(gg, "tokens"), gg["margins"])
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("wdtype", ["h2", "h1", "fp8", "fp8h"])
def test_two_row_blocks_per_weight_pass_are_bit_identical_to_the_walk(wdtype):
    """17..32 decoder rows (the reference's default batch 16 under CFG, configs/generate_vgg.yaml:41 + :27; BASELINE configs[4]): every
    GEMV takes BOTH row blocks per weight fragment (gemv3_kernel / gemv3h_kernel RBK = 2: second accumulator set, one reduction
    barrier, the blocks' epilogues on different waves).  Per row block the products and their order are those of round 4's walk over
    the blocks (second flag word, bit 0): logits bit for bit and tokens, at 20 rows (second block ragged), 32 rows (full) and 40 rows
    (three blocks: a full pass + a half-empty one), eager and through the captured graph."""
    from vaura_amd import _lib as L
    cfg = synth.tiny_sampler(2)
    sd = synth.sampler_state_dict(cfg, seed=91)
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    g = torch.Generator().manual_seed(92)
    out = {}
    try:
        for walk in (0, 1):
            L.lib().vaura_set_debug_flags2(walk)
            eng._free_graph()
            for rows in (20, 32, 40):
                feats = synth.video_features(rows, seed=93).to(DEV)
                idx = torch.randint(0, 1025, (rows, 9, 5), generator=torch.Generator().manual_seed(94 + rows))
                out[("lg", rows, walk)] = eng.logits_all_positions(idx.to(DEV), feats).cpu()
            feats = synth.video_features(16, seed=95).to(DEV)
            for ug in (True, False):
                out[("tok", ug, walk)] = eng.generate_codes(feats, 14, cfg_scale=6.0, use_sampling=True, top_k=128, seed=7, use_graph=ug).cpu()
                eng.check_status()
    finally:
        L.lib().vaura_set_debug_flags2(0)
        eng._free_graph()
    for rows in (20, 32, 40):
        assert torch.isfinite(out[("lg", rows, 0)]).all()
        assert torch.equal(out[("lg", rows, 0)], out[("lg", rows, 1)]), (wdtype, rows, float((out[("lg", rows, 0)] - out[("lg", rows, 1)]).abs().max()))
    assert torch.equal(out[("tok", True, 0)], out[("tok", True, 1)]) and torch.equal(out[("tok", False, 0)], out[("tok", True, 0)])
    assert torch.equal(out[("tok", False, 1)], out[("tok", True, 0)])


def test_reference_shipped_defaults_top_k128_cfg6_h2_B8_matches_reference(golden, full_sampler_sd_raw, parity_report):
    """configs/generate_vgg.yaml:23-27 as shipped — use_sampling, temperature 1.0, top_k 128, top_p 0, cfg_scale 6 — at full depth on the
    un-rounded checkpoint ("auto" -> h2), B=8 -> 16 decoder rows (the bench's launch shapes): clips 0-1 token-identical to the reference's
    own cache-less CPU generate() with its noise stream (make_golden.py full_vgg_raw).  STRICT."""
    g = golden("full_topk128_cfg6_raw_B2_T220.npz")
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV)
    assert eng.wdtype == "h2"
    feats = synth.video_features(8, seed=int(g["feat_seed"])).to(DEV)
    nz8 = torch.cat([synth.exp_noise(228, 18, 1024, int(g["noise_seed"])), synth.exp_noise(228, 54, 1024, 4322)], dim=1)
    tok = eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=int(g["top_k"]), cfg_scale=float(g["cfg_scale"]), noise=nz8).cpu()
    eng.check_status()
    assert int(g["top_k"]) == 128 and eng.rows == 16
    assert_tokens_equal(parity_report, "full_topk128_cfg6_raw_B2_T220", "h2", "generate_vgg.yaml defaults: cfg 6 / top-k 128 sampled, clips 0-1 of B=8 (16 rows, auto)",
                        tok[:2], _ref(g, "tokens"), g["margins"], g["threshold_rel_gap"])
    lg_ref = torch.from_numpy(g["logits"][list(g["logits_steps"]).index(1)])
    f2 = torch.cat([feats[:2], (torch.zeros_like(feats[:2]) + eng.uncond)], 0)
    lg = eng.logits_all_positions(torch.full((4, 9, 1), 1024, dtype=torch.long).to(DEV), f2)[:, :, 0].cpu()
    err = float((lg - lg_ref).abs().max())
    assert err < 3e-5, err
    parity_report.note_logit_err("full_topk128_cfg6_raw_B2_T220", "h2", err, logit_err_step=1)
    del eng
    torch.cuda.empty_cache()


def test_reference_default_batch16_32_rows_full_depth_matches_reference(golden, full_sampler_sd_raw, parity_report):
    """The reference's default BATCH too (configs/generate_vgg.yaml:41 batch_size 16 -> 32 decoder rows under cfg 6: two row blocks per
    weight pass in every GEMV and in the one-launch MLP, 512-workgroup attention), at full depth on the default storage: clips 0-1 of
    B=16 must equal, token for token, what the reference's own generate() produced for them — sampled with the shipped defaults
    (top-k 128), sampled with top-k 250 (the headline golden) and greedy under cfg 6.  STRICT."""
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV)
    assert eng.wdtype == "h2"
    for name, seed2 in (("full_topk128_cfg6_raw_B2_T220", 4322), ("full_topk250_cfg6_raw_B2_T220", 4321)):
        g = golden(name + ".npz")
        feats = synth.video_features(16, seed=int(g["feat_seed"])).to(DEV)
        nz = torch.cat([synth.exp_noise(228, 18, 1024, int(g["noise_seed"])), synth.exp_noise(228, 9 * 14, 1024, seed2)], dim=1)
        tok = eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=int(g["top_k"]), cfg_scale=float(g["cfg_scale"]), noise=nz).cpu()
        eng.check_status()
        assert eng.rows == 32
        assert_tokens_equal(parity_report, name, "h2", f"cfg 6 / top-k {int(g['top_k'])} sampled, clips 0-1 of B=16 (32 rows: the reference's default batch)",
                            tok[:2], _ref(g, "tokens"), g["margins"], g["threshold_rel_gap"])
    gg = golden("full_greedy_cfg6_raw_B2_T220.npz")
    feats = synth.video_features(16, seed=int(gg["feat_seed"])).to(DEV)
    tokg = eng.generate_codes(feats, 220, cfg_scale=float(gg["cfg_scale"])).cpu()
    eng.check_status()
    assert_tokens_equal(parity_report, "full_greedy_cfg6_raw_B2_T220", "h2", "greedy cfg 6, clips 0-1 of B=16 (32 rows)", tokg[:2],
                        _ref(gg, "tokens"), gg["margins"])
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("wdtype", ["auto", "f32"])
def test_later_chunk_of_the_sliding_window_caller_full_depth_matches_reference(golden, full_sampler_sd_raw, parity_report, wdtype):
    """Row f1 at its REAL shape and depth (scripts/generate.py:344-357, a later chunk): prompt Tp = 166 encoded frames, max_new_tokens 221
    (S = 230), remove_prompts=False, cfg 6 -> the batched prompt pass (166 positions x 16 rows through the prefill GEMM and the MFMA
    prefill attention) + 63 sampled steps on the decode kernels — against the reference's own generate() on the un-rounded checkpoint
    (make_golden.py full_chunk_raw): (a) generate_vgg.yaml's defaults (top-k 128 sampled, the reference's noise stream): STRICT;
    (b) greedy under cfg 6: clip 0 STRICT; clip 1 of that reference run holds ONE literal tie (step 175, codebook 8: CFG-mixed top-1 /
    top-2 gap 3.8e-6 = one fp32 ulp at the logits' magnitude; the next smallest margin of the run is 1.4e-4) — tokens must match up
    to there and what each engine does there is recorded ("f32" = the exact-fp32-MFMA engine: only the ORDER of its sums differs from
    torch's).  B=8 on the default storage (16 rows: the one-launch MLP, the 128-row prefill tiles); clips 2-7 carry seeded prompts."""
    gs = golden("full_chunk_topk128_cfg6_raw_B2_Tp166_T221.npz")
    gg = golden("full_chunk_greedy_cfg6_raw_B2_Tp166_T221.npz")
    B = 8 if wdtype == "auto" else 2
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV, wdtype=wdtype)
    st = eng.wdtype
    assert st == ("h2" if wdtype == "auto" else "f32")
    feats = synth.video_features(B, seed=int(gs["feat_seed"])).to(DEV)
    p2 = torch.from_numpy(gs["prompt"].astype(np.int64))
    assert p2.shape == (2, 9, 166) and np.array_equal(gs["prompt"], gg["prompt"])
    prompt = torch.cat([p2, torch.randint(0, 1024, (B - 2, 9, 166), generator=torch.Generator().manual_seed(5))], 0).to(DEV)
    n_pass = 230 - 167
    nz = torch.cat([synth.exp_noise(n_pass, 18, 1024, int(gs["noise_seed"])), synth.exp_noise(n_pass, 9 * (B - 2), 1024, 4323)], dim=1)
    tok = eng.generate_codes(feats, 221, prompt=prompt, use_sampling=True, temp=1.0, top_k=int(gs["top_k"]), cfg_scale=float(gs["cfg_scale"]),
                             noise=nz).cpu()
    eng.check_status()
    assert torch.equal(tok[:, :, :166], prompt.cpu())                          # remove_prompts=False: the prompt is part of the output
    assert_tokens_equal(parity_report, "full_chunk_topk128_cfg6_raw_B2_Tp166_T221", st,
                        f"later long-form chunk (Tp=166, T=221), cfg 6 / top-k 128 sampled, clips 0-1 of B={B}", tok[:2], _ref(gs, "tokens"),
                        gs["margins"], gs["threshold_rel_gap"], first_step=167)
    # the first sampled pass's [cond; null] logits (sequence length 167 = the whole prompt pass + one position) against the reference's
    tokg = eng.generate_codes(feats, 221, prompt=prompt, cfg_scale=float(gg["cfg_scale"])).cpu()
    eng.check_status()
    refg = _ref(gg, "tokens")
    assert_tokens_equal(parity_report, "full_chunk_greedy_cfg6_raw_B2_Tp166_T221", st, f"later long-form chunk, greedy cfg 6, clip 0 of B={B}",
                        tokg[:1], refg[:1], gg["margins"][:, :1], first_step=167)
    e = assert_tokens_or_recorded_near_tie(parity_report, "full_chunk_greedy_cfg6_raw_B2_Tp166_T221", st,
                                           f"later long-form chunk, greedy cfg 6, clip 1 of B={B} (holds the run's one literal tie: step 175)",
                                           tokg[1:2], refg[1:2], gg["margins"][:, 1:2], 2e-5, first_step=167)
    assert e["tokens_equal"] or e["first_diff_step"] == 175, e
    del eng
    torch.cuda.empty_cache()


def test_full_length_codec_decode_and_encode_match_the_oracle_on_the_256_row_instances():
    """What the bench's codec stage actually launches: T=220, B=8, default precision (f16pair) — every conv but conv_in runs
    conv_pair_kernel<..., 8> (256-row workgroups, csrc/dac.hip), asserted through the library's launch counter — against the fp32
    CPU restatement DIRECTLY (not through the 128-row instances): waveform RMS <= 1e-4; then the full-length encode (2.56 s, the
    64-column 256-row instances) against the oracle's codes (a code may differ only from a proven near-tie of the argmin on).
    DAC itself is parity-unpinned by the reference (oracle/__init__.py): the oracle is the restated published architecture."""
    from oracle import dac_oracle
    from vaura_amd import _lib as L
    from vaura_amd.engine import CodecEncoderEngine
    ccfg = synth.FULL_CODEC
    sd = dict(synth.codec_state_dict(ccfg, seed=1))
    sd.update(synth.codec_encoder_state_dict(ccfg, seed=1))
    B, T = 8, 220
    codes = torch.randint(0, 1024, (B, 9, T), generator=torch.Generator().manual_seed(6))
    dec = CodecEngine(ccfg, sd, DEV, precision="f16pair")
    L.lib().vaura_debug_counter(0)
    L.lib().vaura_debug_counter(1)
    got = dec.decode(codes.to(DEV))
    torch.cuda.synchronize()
    n256, nfused = int(L.lib().vaura_debug_counter(0)), int(L.lib().vaura_debug_counter(1))
    # 4 x (up + 3 x 2) = 28 of the 30 convs; a residual unit launched as ONE kernel (C = 96 and C = 192: the last two blocks' six) covers two of them
    assert n256 + 2 * nfused >= 28 and nfused == 6, f"only {n256} conv launches + {nfused} fused units took the 256-row instances"
    # ... and the one-launch units are bit-identical to the two-launch form (debug flag bit 21), which is what the oracle is compared with below too
    L.lib().vaura_set_debug_flags(1 << 21)
    try:
        two = dec.decode(codes.to(DEV))
        torch.cuda.synchronize()
        assert int(L.lib().vaura_debug_counter(1)) == 0 and int(L.lib().vaura_debug_counter(0)) >= 28
    finally:
        L.lib().vaura_set_debug_flags(0)
    assert torch.equal(two, got), float((two - got).abs().max())
    got = got.cpu()
    worst = 0.0
    for b in range(B):                                                                  # clip by clip: bounds the oracle's memory
        ref = dac_oracle.decode(sd, codes[b:b + 1], ccfg.decoder_rates)
        assert ref.shape == (1, 1, T * 512)
        rms = float(((got[b:b + 1] - ref) ** 2).mean().sqrt())
        worst = max(worst, rms)
        assert float(ref.abs().max()) > 0.05
    print(f"full-length decode (B=8, T=220, {n256} launches on 256-row instances): worst per-clip rms vs the oracle {worst:.3e}")
    assert worst <= 1e-4, worst
    # encode at 2.56 s
    g = torch.Generator().manual_seed(7)
    n = T * 512
    t = torch.arange(n) / 44100.0
    wav = 0.4 * torch.sin(2 * torch.pi * 330.0 * t) + 0.15 * torch.randn(B, 1, n, generator=g)
    enc = CodecEncoderEngine(ccfg, sd, DEV)
    L.lib().vaura_debug_counter(0)
    gotc = enc.encode(wav.to(DEV)).cpu()
    n256e = int(L.lib().vaura_debug_counter(0))
    assert n256e >= 20, n256e
    agree_all, bad_first = [], 0
    for b in range(0, B, 2):
        z = dac_oracle.encode_latent(sd, dac_oracle.preprocess(wav[b:b + 2], 512), ccfg.encoder_rates)
        ref, margin = dac_oracle.quantize(sd, z, ccfg.n_codebooks, return_margin=True)
        gc = gotc[b:b + 2]
        assert gc.shape == ref.shape == (2, 9, T)
        agree_all.append(float((gc == ref).float().mean()))
        clean = ~((gc != ref).float().cumsum(1) > 0)
        for bb, k, tt in torch.nonzero(gc != ref).tolist():
            if k == 0 or bool(clean[bb, k - 1, tt]):                                    # a first mismatch must sit on a near-tie
                assert float(margin[bb, k, tt]) < 5e-4, (b + bb, k, tt, float(margin[bb, k, tt]))
                bad_first += 1
    print(f"full-length encode (B=8, 2.56 s, {n256e} launches on 256-row instances): code agreement {min(agree_all):.4f} (min over "
          f"clip pairs), {bad_first} frames leave the oracle at a proven near-tie")
    assert min(agree_all) > 0.9


def test_range_guard_detects_overflow_and_the_checked_call_reruns_it_in_fp32():
    """Activations travel between the decode kernels as (hi, lo) fp16 planes: |x| > 65504 becomes inf / NaN.  A checkpoint whose
    residual stream x next-norm gain leaves that range (token embedding and norm gains scaled up: values ~1e5..1e6) must end in
    VauraHipError from check_status() — the sampler raises the sticky device status bit on non-finite logits — not in tokens;
    a checkpoint that stays in range (the same one unscaled) passes, and the bit does not stick after a raise."""
    from vaura_amd import _lib as L
    cfg = synth.tiny_sampler(2)
    sd = dict(synth.sampler_state_dict(cfg, seed=81))
    eng = DecoderEngine(cfg, sd, DEV, wdtype="h2")
    feats = synth.video_features(2, seed=82).to(DEV)
    eng.generate_codes(feats, 12, cfg_scale=6.0)
    eng.check_status()
    big = dict(sd)
    for k in sd:
        if k.endswith("attention_norm.weight") or k.endswith("ffn_norm.weight") or k == "norm.weight":
            big[k] = sd[k] * 3000.0
        if "tok_embeddings" in k and k.endswith("out_proj.weight_g"):
            big[k] = sd[k] * 3000.0
    e2 = DecoderEngine(cfg, big, DEV, wdtype="h2")
    e2.generate_codes(feats, 12, cfg_scale=6.0)
    with pytest.raises(L.VauraHipError, match="non-finite logits"):
        e2.check_status()
    e2.check_status()                                   # read-and-clear: the bit is gone
    # the exact-fp32 path keeps fp32 activations: the same checkpoint decodes (fp32 has the range), finite logits
    e3 = DecoderEngine(cfg, big, DEV, wdtype="f32")
    tok = e3.generate_codes(feats, 12, cfg_scale=6.0).cpu()
    e3.check_status()
    assert int(tok.min()) >= 0 and int(tok.max()) < 1024
    # RANGE SAFETY BY CONSTRUCTION (generate_codes_checked, what VAURAModel.generate calls): the overflowing call is re-run on the
    # exact-fp32 twin — the x3000 checkpoint decodes token-exact against the oracle, greedy and sampled, instead of raising
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    dec = DecoderOracle(big, cfg.num_layers, cfg.nhead)
    ref = go.generate(dec, feats.cpu(), 12, mode="cached", cfg_scale=6.0)
    got = e2.generate_codes_checked(feats, 12, cfg_scale=6.0).cpu()
    assert e2.range_fallbacks == 1 and torch.equal(got, ref) and torch.equal(got, tok)
    nz = synth.exp_noise(12 + 9 - 1, 18, 1024, 83)
    refs = go.generate(dec, feats.cpu(), 12, mode="cached", cfg_scale=6.0, use_sampling=True, top_k=128, noise=nz)
    gots = e2.generate_codes_checked(feats, 12, cfg_scale=6.0, use_sampling=True, top_k=128, noise=nz).cpu()
    assert e2.range_fallbacks == 2 and torch.equal(gots, refs)
    e2.check_status()                                   # nothing left behind
    # a checkpoint that stays in range never takes the detour
    assert torch.equal(eng.generate_codes_checked(feats, 12, cfg_scale=6.0), eng.generate_codes(feats, 12, cfg_scale=6.0)) and eng.range_fallbacks == 0
    # LARGE NORM GAINS ALONE do not leave the planes' range: the engine folds a power of two out of every RMSNorm gain into the matrix
    # that consumes it (engine.fold_gain: exact on both sides), so the planes hold (g 2^-E) * h.  attention_norm gains x 3000 (the residual
    # stream then grows to ~1e3..1e4 and h * g to ~1e7) decode on the DEFAULT arithmetic, token-exact, without the fp32 detour
    ga = dict(sd)
    for k in sd:
        if k.endswith("attention_norm.weight"):
            ga[k] = sd[k] * 3000.0
    e4 = DecoderEngine(cfg, ga, DEV, wdtype="h2")
    ref4 = go.generate(DecoderOracle(ga, cfg.num_layers, cfg.nhead), feats.cpu(), 12, mode="cached", cfg_scale=6.0)
    got4 = e4.generate_codes_checked(feats, 12, cfg_scale=6.0).cpu()
    assert e4.range_fallbacks == 0 and torch.equal(got4, ref4)


def test_range_fallback_of_a_lossy_storage_decodes_the_model_the_storage_holds():
    """the exact-fp32 twin of a LOSSY storage (fp8 / fp8h) must hold the dequantised matrices — the model every other call of
    the job is decoded by — not the original state dict.  A x3000 checkpoint on the fp8 engine overflows the planes, the checked call is
    re-run on the twin, and the result equals the oracle on `quant.fp8_effective_state_dict(checkpoint)` (and differs from the oracle on
    the unquantised one: the two models are different)."""
    import warnings
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    from vaura_amd import quant
    cfg = synth.tiny_sampler(2)
    sd = dict(synth.sampler_state_dict(cfg, seed=81))
    big = dict(sd)
    for k in sd:
        if k.endswith("attention_norm.weight") or k.endswith("ffn_norm.weight") or k == "norm.weight":
            big[k] = sd[k] * 3000.0
        if "tok_embeddings" in k and k.endswith("out_proj.weight_g"):
            big[k] = sd[k] * 3000.0
    feats = synth.video_features(2, seed=82)
    eng = DecoderEngine(cfg, big, DEV, wdtype="fp8")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = eng.generate_codes_checked(feats.to(DEV), 12, cfg_scale=6.0).cpu()
    assert eng.range_fallbacks == 1 and eng._range_twin is not None and eng._twin_lossy
    assert any("DEQUANTISED" in str(x.message) for x in w)
    ref_eff = go.generate(DecoderOracle(quant.fp8_effective_state_dict(big), cfg.num_layers, cfg.nhead), feats, 12, mode="cached", cfg_scale=6.0)
    ref_raw = go.generate(DecoderOracle(big, cfg.num_layers, cfg.nhead), feats, 12, mode="cached", cfg_scale=6.0)
    assert torch.equal(got, ref_eff)
    assert not torch.equal(ref_eff, ref_raw)


def test_plane_shift_moves_the_fp16_plane_range_up_at_the_same_speed(golden, full_sampler_sd_raw, parity_report):
    """DecoderEngine(plane_shift=S): every activation plane set is stored times 2^-S, its consuming matrix times 2^S (exact both ways
    while the lo plane stays out of fp16's subnormals), so the planes end at 65504 * 2^S.
      * the x3000 checkpoint of the range-guard test (SwiGLU outputs ~1e7), which S = 0 can only decode through the exact-fp32 twin (and
        S = 8 too: tools/plane_shift_probe.py), decodes on the h2 kernels with S = 12: no status bit, no fallback, tokens equal to the
        oracle's — greedy and top-k sampled;
      * a checkpoint in range decodes to the same tokens at S = 4 as at S = 0 in every plane storage, one and two row blocks, with a
        teacher-forced prompt (the GEMM-tiled prefill stores the same planes), logits within 2e-5 of S = 0's;
      * full depth: the headline golden (reference generate(), cfg 6 / top-k 250, 16 rows) token for token at S = 4;
      * S is refused on the exact-fp32 step and outside 0..24."""
    from vaura_amd import _lib as L
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.tiny_sampler(2)
    sd = dict(synth.sampler_state_dict(cfg, seed=81))
    feats = synth.video_features(2, seed=82).to(DEV)
    big = dict(sd)
    for k in sd:
        if k.endswith("attention_norm.weight") or k.endswith("ffn_norm.weight") or k == "norm.weight":
            big[k] = sd[k] * 3000.0
        if "tok_embeddings" in k and k.endswith("out_proj.weight_g"):
            big[k] = sd[k] * 3000.0
    dec = DecoderOracle(big, cfg.num_layers, cfg.nhead)
    e8 = DecoderEngine(cfg, big, DEV, wdtype="h2", plane_shift=12)
    ref = go.generate(dec, feats.cpu(), 12, mode="cached", cfg_scale=6.0)
    got = e8.generate_codes_checked(feats, 12, cfg_scale=6.0).cpu()
    assert e8.range_fallbacks == 0 and torch.equal(got, ref)
    nz = synth.exp_noise(12 + 9 - 1, 18, 1024, 83)
    refs = go.generate(dec, feats.cpu(), 12, mode="cached", cfg_scale=6.0, use_sampling=True, top_k=128, noise=nz)
    gots = e8.generate_codes_checked(feats, 12, cfg_scale=6.0, use_sampling=True, top_k=128, noise=nz).cpu()
    assert e8.range_fallbacks == 0 and torch.equal(gots, refs)
    e0 = DecoderEngine(cfg, big, DEV, wdtype="h2", range_fallback=False)     # ... and S = 0 does overflow on it
    e0.generate_codes(feats, 12, cfg_scale=6.0)
    with pytest.raises(L.VauraHipError, match="non-finite logits"):
        e0.check_status()
    del e8, e0
    prompt = torch.randint(0, 1024, (10, 9, 40), generator=torch.Generator().manual_seed(5)).to(DEV)
    idx = torch.randint(0, 1024, (3, 9, 24), generator=torch.Generator().manual_seed(6)).to(DEV)
    for wd in ("h2", "h1", "fp8"):
        a = DecoderEngine(cfg, sd, DEV, wdtype=wd)
        b = DecoderEngine(cfg, sd, DEV, wdtype=wd, plane_shift=4)
        for B in (2, 10):                                                   # 4 rows; 20 rows = two row blocks per weight pass
            f = synth.video_features(B, seed=90 + B).to(DEV)
            nzb = synth.exp_noise(30 + 9 - 1, 9 * B, 1024, 91)
            kw = dict(cfg_scale=6.0, use_sampling=True, top_k=250, noise=nzb)
            assert torch.equal(a.generate_codes(f, 30, **kw), b.generate_codes(f, 30, **kw)), (wd, B)
            assert torch.equal(a.generate_codes(f, 30, cfg_scale=6.0), b.generate_codes(f, 30, cfg_scale=6.0)), (wd, B)
        f = synth.video_features(10, seed=93).to(DEV)
        assert torch.equal(a.generate_codes(f, 60, prompt=prompt, cfg_scale=6.0), b.generate_codes(f, 60, prompt=prompt, cfg_scale=6.0)), wd
        la = a.logits_all_positions(idx, synth.video_features(3, seed=94).to(DEV))
        lb = b.logits_all_positions(idx, synth.video_features(3, seed=94).to(DEV))
        err = float((la - lb).abs().max())
        print(f"{wd}: logits at plane_shift 4 against 0, max-abs {err:.2e}")
        assert err < 2e-5
        a.check_status(), b.check_status()
        del a, b
    with pytest.raises(L.VauraHipError, match="plane_shift"):
        DecoderEngine(cfg, sd, DEV, wdtype="f32", plane_shift=4)
    with pytest.raises(L.VauraHipError, match="plane_shift"):
        DecoderEngine(cfg, sd, DEV, wdtype="h2", plane_shift=25)
    torch.cuda.empty_cache()
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV, plane_shift=4)
    assert eng.wdtype == "h2"
    gs = golden("full_topk250_cfg6_raw_B2_T220.npz")
    f8 = synth.video_features(8, seed=int(gs["feat_seed"])).to(DEV)
    nz8 = torch.cat([synth.exp_noise(228, 18, 1024, int(gs["noise_seed"])), synth.exp_noise(228, 54, 1024, 4321)], dim=1)
    tok = eng.generate_codes(f8, 220, use_sampling=True, temp=1.0, top_k=int(gs["top_k"]), cfg_scale=float(gs["cfg_scale"]), noise=nz8).cpu()
    eng.check_status()
    assert_tokens_equal(parity_report, "full_topk250_cfg6_raw_B2_T220", "h2 plane_shift 4",
                        "headline golden with every plane set stored x 2^-4 (16x the fp16 range)", tok[:2], _ref(gs, "tokens"),
                        gs["margins"], gs["threshold_rel_gap"])
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["f16pair", "f16", "f16pair_w8"])
def test_one_launch_residual_units_are_bit_identical_to_two_launches(precision):
    """csrc/dac.hip: a residual unit (7-tap dilated conv -> Snake -> 1 x 1 conv -> + residual) of the last two decoder blocks (C = 192:
    conv_unit_kernel, C = 96: conv_pair_kernel<..., FUSE>) runs as ONE launch wherever >= 384 workgroups remain; debug flag bit 21 keeps
    the two launches.  Same matrix instructions on the same fragments in the same order: the waveform must be BIT-identical, in every
    pair-arithmetic precision (each has its own kernel instance), at a batch / length other than the bench's (B = 3, T = 130: six
    one-launch units, asserted through the library's counter), both sequence ends and the clip boundaries included."""
    from vaura_amd import _lib as L
    ccfg = synth.FULL_CODEC
    sd = synth.codec_state_dict(ccfg, seed=4)
    codes = torch.randint(0, 1024, (3, 9, 130), generator=torch.Generator().manual_seed(8)).to(DEV)
    dec = CodecEngine(ccfg, sd, DEV, precision=precision)
    try:
        L.lib().vaura_debug_counter(1)
        one = dec.decode(codes).clone()
        torch.cuda.synchronize()
        assert int(L.lib().vaura_debug_counter(1)) == 6
        L.lib().vaura_set_debug_flags(1 << 21)
        two = dec.decode(codes).clone()
        torch.cuda.synchronize()
        assert int(L.lib().vaura_debug_counter(1)) == 0
    finally:
        L.lib().vaura_set_debug_flags(0)
    assert torch.isfinite(one).all() and float(one.abs().max()) > 0.01
    assert torch.equal(one, two), float((one - two).abs().max())


@pytest.mark.parametrize("wdtype", ["h2", "h1"])
def test_heavy_tailed_checkpoint_rows_against_live_oracle(wdtype):
    """The fp16-plane storage keeps ONE power-of-two scale per output row.  Rows that stress it: a 100-sigma outlier (ordinary
    weights of that row land 7 binades below the row maximum: their lo plane approaches fp16's subnormal range), an all-zero
    row (scale 1), and a row of tiny weights (max 1e-30: the scale itself is far below fp16's range, the planes are not).  h2: the
    packed matrix must hold every element to 2^-21 of the ROW maximum or better (what a 22-bit split of the scaled row gives), the
    engine must be token-exact against the oracle on the checkpoint the planes hold, and within 3e-5 logits of the oracle on the
    ORIGINAL checkpoint.  h1 forced on it (bf16-rounded variant incl. the outliers): resolve_weight_dtype says 'h1' and tokens are exact."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    from vaura_amd.engine import h_effective_weight, resolve_weight_dtype
    cfg = synth.tiny_sampler(2)
    sd = dict(synth.sampler_state_dict(cfg, seed=91, round_bf16=(wdtype == "h1")))
    g = torch.Generator().manual_seed(92)
    for k in [k for k in sd if synth.is_streamed_weight(k)]:
        w = sd[k].clone()
        rows = torch.randperm(w.shape[0], generator=g)[:6]
        cols = torch.randint(0, w.shape[1], (6,), generator=g)
        w[rows[0], cols[0]] = 2.0                       # 100 sigma (sigma = 0.02)
        w[rows[1], cols[1]] = -2.0
        w[rows[2], cols[2]] = 64.0                      # 3200 sigma: ordinary weights 12 binades down
        w[rows[3]] = 0.0                                # zero row
        w[rows[4]] = w[rows[4]] * 1e-28                 # tiny row: scale ~2^-113
        sd[k] = w.bfloat16().float() if wdtype == "h1" else w
    planes = 2 if wdtype == "h2" else 1
    assert resolve_weight_dtype(sd, "auto") == wdtype
    sd_eff = dict(sd)
    for k in [k for k in sd if synth.is_streamed_weight(k)]:
        eff = h_effective_weight(sd[k], planes)
        sd_eff[k] = eff
        amax = sd[k].abs().amax(dim=1, keepdim=True)
        err = (eff - sd[k]).abs()
        bound = amax * (2.0 ** -21 if planes == 2 else 2.0 ** -10)
        assert bool((err <= bound).all()), k
        if planes == 1:
            assert torch.equal(eff, sd[k]), k           # lossless: that is what "auto" promised
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    feats = synth.video_features(3, seed=93)
    dec_eff = DecoderOracle(sd_eff, cfg.num_layers, cfg.nhead)
    ref = go.generate(dec_eff, feats, 20, mode="cached", cfg_scale=6.0)
    got = eng.generate_codes(feats.to(DEV), 20, cfg_scale=6.0).cpu()
    eng.check_status()
    assert torch.equal(got, ref), float((got == ref).float().mean())
    idx = ref[:, :, :10].contiguous()
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    lg_ref = dec.forward_full(idx, feats)
    lg = eng.logits_all_positions(idx.to(DEV), feats.to(DEV)).cpu()
    err = float((lg - lg_ref).abs().max())
    print(f"heavy-tailed rows, {wdtype}: logits max-abs error vs the oracle on the original checkpoint {err:.3e}")
    assert err < 3e-5 * max(1.0, float(lg_ref.abs().max())), err


@pytest.mark.parametrize("clips", [6, 3, 16, 10])
@pytest.mark.parametrize("wdtype", ["h2", "h1", "fp8", "fp8h"])
def test_one_launch_mlp_is_bit_identical_to_two_launches(wdtype, clips):
    """csrc/mlp_engine.h, the default where eligible: the MLP of a layer (w1||w3 + SwiGLU -> w2 + residual) as ONE launch with an
    in-launch hand-off, w2's weights requested ahead of it (debug flag bit 2: every GEMV its own launch; bit 3: the experimental
    three-phase launch that takes wo in as well).  Same products in the same order as the separate launches: teacher-forced
    logits must be BIT-identical, tokens (greedy + CFG, and Philox-sampled) identical, through the eager path and through the
    captured step graph, and no consumer may have given up waiting (status word clean).  6 clips x cfg = 12 decoder rows (both row
    halves live) and 3 clips = 6 rows (configs[3]'s regime: one live half; the separate launches then use one workgroup per tile, the
    one-launch form keeps its (tile, row half) workgroups and the second half multiplies zeros).  Round 5: 16 clips x cfg = 32 rows (the
    reference's default batch, configs/generate_vgg.yaml:41) and 10 clips = 20 rows (second row block ragged): the two-row-block
    instances (mlp_engine_kernel<.., RBK = 2>) against the separate two-row-block launches; the attention + wo and tail experiments
    do not exist there (the flags then select the default).  fp8 tile pairs (configs[4]) take the one-launch form with 17..32 rows only
    (its qkv phase on four waves of six k-groups, like the separate fp8 K-split kernel): 16 and 10 clips test it, 6 and 3 are the
    separate launches either way."""
    from vaura_amd import _lib as L
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=101, round_bf16=(wdtype == "h1"))
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    feats = synth.video_features(clips, seed=102).to(DEV)
    idx = torch.randint(0, 1024, (2 * clips, 9, 20), generator=torch.Generator().manual_seed(103)).to(DEV)
    f12 = synth.video_features(2 * clips, seed=104).to(DEV)
    out = {}
    try:
        # the default (w1||w3 -> w2 -> next layer's qkv in one launch) | qkv separate | every GEMV and the attention their own launches.
        # (The three measured-negative engines — attention + wo, the layer tail, the attention as a fourth phase — left the product
        # library in round 6: experiment builds only, `tools/experiment.sh engines`; their bit-identity records are in DESIGN_HISTORY.md.)
        for flags in (0, 0x2, 4):
            f1, f2 = flags if isinstance(flags, tuple) else (flags, 0)
            L.lib().vaura_set_debug_flags(f1)
            L.lib().vaura_set_debug_flags2(f2)
            eng._free_graph()
            out[flags] = (eng.logits_all_positions(idx, f12).clone(),
                          eng.generate_codes(feats, 24, cfg_scale=6.0).clone(),
                          eng.generate_codes(feats, 24, cfg_scale=6.0, use_sampling=True, top_k=250, seed=9).clone(),
                          eng.generate_codes(feats, 24, cfg_scale=6.0, use_graph=False).clone())
            eng.check_status()
    finally:
        L.lib().vaura_set_debug_flags(0)
        L.lib().vaura_set_debug_flags2(0)
        eng._free_graph()
    torch.cuda.synchronize()
    assert torch.isfinite(out[4][0]).all()
    for flags in (0, 0x2):
        assert torch.equal(out[flags][0], out[4][0]), (flags, float((out[flags][0] - out[4][0]).abs().max()))
        for i in (1, 2, 3):
            assert torch.equal(out[flags][i], out[4][i]), (flags, i)
    assert torch.equal(out[4][1], out[4][3])


@pytest.mark.parametrize("clips", [8, 16])
@pytest.mark.parametrize("one_launch", [True, False])
def test_one_launch_mlp_under_concurrent_load_and_repeats(one_launch, clips):
    """The in-launch hand-offs of csrc/mlp_engine.h under conditions an idle chip hides (MI355X_MICROARCH.md: "test every hand-off under
    UNEVEN load"): (1) the decode loop replayed many times on the same inputs must give the same tokens every time (a stale plane or a
    flag seen too early would show as a difference sooner or later); (2) with a SECOND stream keeping the chip busy with full-chip
    codec launches — workgroups of the MLP launch then become resident late and unevenly, its consumers wait (bounded) for producers
    that have not started — the tokens must still equal the quiet run's and no consumer may have given up (status word clean)."""
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=111, round_bf16=False)
    eng = DecoderEngine(cfg, sd, DEV, one_launch_mlp=one_launch)     # False: the control (separate launches under the same load)
    assert eng.wdtype == "h2" and eng.one_launch_mlp == one_launch
    feats = synth.video_features(clips, seed=112).to(DEV)            # 16 clips: 32 rows, the two-row-block instances
    kw = dict(cfg_scale=6.0, use_sampling=True, top_k=250, seed=5)
    ref = eng.generate_codes(feats, 60, **kw).clone()
    eng.check_status()
    for _ in range(12):
        assert torch.equal(eng.generate_codes(feats, 60, **kw), ref)
    eng.check_status()
    ccfg = synth.FULL_CODEC
    codec = CodecEngine(ccfg, synth.codec_state_dict(ccfg, seed=0), DEV)
    codes = torch.randint(0, 1024, (8, 9, 220), device=DEV)
    side = torch.cuda.Stream(DEV)
    main = torch.cuda.Stream(DEV)
    codec.decode(codes)                     # workspaces allocated before the overlap
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(4):
                codec.decode(codes)         # ~45 ms of full-chip conv grids next to the loop
        with torch.cuda.stream(main):
            outs.append(eng.generate_codes(feats, 60, **kw).clone())
        torch.cuda.synchronize()
    eng.check_status()                       # a consumer that gave up waiting raises here
    for o in outs:
        assert torch.equal(o, ref)


def test_handoff_timeout_switches_the_engine_to_separate_launches_instead_of_failing():
    """the one-launch MLP needs all 256 workgroups resident; on a SHARED GPU (MPS, a second tenant, a CU mask) a
    consumer gives up waiting (bounded spin) and raises VAURA_STATUS_HANDOFF_TIMEOUT — after which every hand-off returns at once and
    the tokens are garbage.  `generate_codes_checked` (what VAURAModel.generate calls) must turn that into a slower, CORRECT run: switch
    the engine to the separate launches for good and run the call again.  The timeout is simulated by raising the status bit before the
    call (so the first pass really decodes with every wait disabled), the result must equal the quiet run's."""
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=121, round_bf16=False)
    eng = DecoderEngine(cfg, sd, DEV)
    feats = synth.video_features(8, seed=122).to(DEV)
    kw = dict(cfg_scale=6.0, use_sampling=True, top_k=250, seed=5)
    ref = eng.generate_codes_checked(feats, 40, **kw).clone()
    assert eng.one_launch_mlp and eng.dec.ws_sync
    eng.state[4:5].fill_(2)                              # VAURA_STATUS_HANDOFF_TIMEOUT
    with pytest.warns(UserWarning, match="hand-off timed out"):
        got = eng.generate_codes_checked(feats, 40, **kw)
    assert torch.equal(got, ref)
    assert not eng.one_launch_mlp and not eng.dec.ws_sync and eng.handoff_fallbacks == 1
    assert torch.equal(eng.generate_codes_checked(feats, 40, **kw), ref)       # and it stays on the separate launches, clean
    eng.check_status()


@pytest.mark.parametrize("massive", [0.0, 3000.0])
def test_full_depth_trained_like_statistics_against_live_oracle(full_sampler_sd_raw, parity_report, massive):
    """VERDICT r4 weak #5 / missing #3: all 24-layer evidence was on N(0, 0.02^2) weights with unit gains.  Here the full-depth
    checkpoint carries the statistics of a TRAINED transformer (synth.trained_like: heavy-tailed matrices, log-normal norm gains with
    outlier channels x 20, two token-embedding channels x 100 = "massive activations" in the residual stream) and the default path
    ("auto" -> h2, fp16-plane activations, the one-launch MLP) is compared with the live fp32 CPU oracle on THAT checkpoint: greedy
    under cfg 6 (4 decoder rows), tokens strict, CFG-mixed first-forward logits within 3e-5 of the logits' scale; the oracle's own
    top-1 / top-2 margins are recorded.  massive = 3000: two norm gains deep in the stack x 3000 on top — activations leave the
    fp16-plane range, the status bit rises, and `generate_codes_checked` must return the exact-fp32 twin's tokens = the oracle's."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    cfg = synth.FULL_SAMPLER
    sd = synth.trained_like(full_sampler_sd_raw, seed=7, massive=massive)
    feats = synth.video_features(2, seed=131)
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    trace = {}
    T = 12
    ref = go.generate(dec, feats, T, mode="cached", cfg_scale=6.0, trace=trace)
    steps = sorted(trace["logits"])
    margins = np.stack([(lambda t2: (t2[..., 0] - t2[..., 1]).numpy())(torch.topk(trace["logits"][s_], 2, dim=-1).values) for s_ in steps])
    scale = float(max(trace["logits"][s_].abs().max() for s_ in steps))
    eng = DecoderEngine(cfg, sd, DEV)
    assert eng.wdtype == "h2"
    got = eng.generate_codes_checked(feats.to(DEV), T, cfg_scale=6.0).cpu()
    what = f"trained-like statistics{' + massive activations (x3000 gains): exact-fp32 twin' if massive else ''}, greedy cfg 6, B=2, T={T} vs the LIVE oracle"
    e = assert_tokens_equal(parity_report, "live oracle (synth.trained_like, full depth)", "h2" if not massive else "h2 -> f32 twin", what, got, ref,
                            margins, logits_scale=scale, range_fallbacks=eng.range_fallbacks)
    assert eng.range_fallbacks == (1 if massive else 0), eng.range_fallbacks
    if not massive:
        f2 = torch.cat([feats, dec.null_condition(feats)], 0).to(DEV)
        lg = eng.logits_all_positions(torch.full((4, 9, 1), 1024, dtype=torch.long).to(DEV), f2)[:, :, 0].cpu()
        mixed = lg[2:] + (lg[:2] - lg[2:]) * 6.0
        err = float((mixed - trace["logits"][1]).abs().max())
        print(f"trained-like statistics: |logits| up to {scale:.1f}, CFG-mixed first-forward error {err:.3e}, oracle min margin {float(margins.min()):.3e}")
        parity_report.note_logit_err("live oracle (synth.trained_like, full depth)", "h2", err, logits_scale=scale)
        assert err < 3e-4 * max(1.0, scale), (err, scale)
    del eng
    torch.cuda.empty_cache()


_EDGE_CASES = [
    # B, T, Tp, Tv, kwargs                                       (edge of ...)
    (1, 1, 0, 32, dict()),                                       # a single frame: S = 10, nine of the ten steps only flush the delay pattern
    (1, 2, 1, 32, dict(cfg_scale=6.0)),                          # prompt = T - 1: ONE frame to generate, prompt pass of 1 position
    (3, 10, 9, 32, dict(use_sampling=True, top_k=1)),            # top-k 1 under sampling = argmax whatever the noise
    (2, 12, 0, 32, dict(use_sampling=True, temp=0.0)),           # temp <= 0 -> greedy (vaura_model.py:816, 825)
    (2, 12, 0, 32, dict(use_sampling=True, top_k=1024)),         # k = the whole codebook: every token kept
    (2, 12, 3, 32, dict(use_sampling=True, top_k=250, top_p=1.0, cfg_scale=3.5)),   # top-p wins over top-k (:818-823); p = 1 keeps all
    (2, 12, 0, 32, dict(use_sampling=True, top_p=0.05, temp=1.7)),                   # tiny nucleus, hot temperature
    (5, 9, 0, 1, dict(cfg_scale=1.0)),                           # ONE video token: positions >= 7 read empty_video_emb
    (17, 6, 2, 32, dict(cfg_scale=6.0, use_sampling=True, top_k=128)),               # 34 rows: three row blocks, ragged
    (9, 7, 0, 32, dict(use_sampling=True)),                      # plain multinomial, 9 rows (ragged single block)
]


@pytest.mark.parametrize("case", range(len(_EDGE_CASES)))
@pytest.mark.parametrize("wdtype", ["h2", "f32"])
def test_edge_cases_against_live_oracle(case, wdtype):
    """Corners of generate() (vaura_model.py:410-597, 775-827; utils/utils.py:139-196) against the live oracle, token for token, on
    the default arithmetic and on the exact-fp32 engine: a single frame, a prompt of T - 1 frames, top-k 1 / beyond the codebook,
    temperature 0, top-p with top-k set (top-p wins), a tiny nucleus, one video token, 34 rows, plain multinomial."""
    from oracle import generate_oracle as go
    from oracle.decoder_oracle import DecoderOracle
    B, T, Tp, Tv, kw = _EDGE_CASES[case]
    cfg = synth.tiny_sampler(2)
    sd = synth.sampler_state_dict(cfg, seed=141, round_bf16=False)
    if kw.get("cfg_scale", 1.0) > 1.0 and Tv != 32:
        pytest.skip("the CFG null embedding is fixed at 32 tokens")
    feats = synth.video_features(B, tokens=Tv, seed=142 + case)
    prompt = torch.randint(0, 1024, (B, 9, Tp), generator=torch.Generator().manual_seed(143 + case)) if Tp else None
    dec = DecoderOracle(sd, cfg.num_layers, cfg.nhead)
    n_pass = T + 9 - (Tp + 1)
    nz = synth.exp_noise(n_pass, B * 9, 1024, 144 + case) if kw.get("use_sampling") else None
    ref = go.generate(dec, feats, T, prompt=prompt, mode="cached", noise=nz, **kw)
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    got = eng.generate_codes_checked(feats.to(DEV), T, prompt=None if prompt is None else prompt.to(DEV), noise=nz, **kw).cpu()
    assert got.shape == (B, 9, T)
    assert torch.equal(got, ref), (case, wdtype, float((got == ref).float().mean()))
    assert eng.range_fallbacks == 0
    if case == 0:
        # k beyond the codebook: the reference's sample_top_k calls torch.topk(probs, k) and raises (utils/utils.py:172); so does this build
        from vaura_amd import _lib as L
        with pytest.raises(L.VauraHipError, match="top_k"):
            eng.generate_codes(feats.to(DEV), T, use_sampling=True, top_k=5000)
        eng.generate_codes(feats.to(DEV), T, use_sampling=True, top_k=5000, top_p=0.5)      # top-p wins: top_k is never looked at (:818-823)


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (3, 7), (2, 33)])
def test_dac_decode_edge_lengths_against_the_oracle(B, T):
    """Codec decode at the short end (a single frame = 512 samples: every conv's receptive field is mostly zero padding, every
    workgroup tile is ragged) and at odd lengths / batches: default precision within the north star's 1e-4 RMS of the fp32 CPU
    restatement, the exact-fp32 precision within 5e-6; and the plugin's EnCodec-style frame list input gives the same waveform."""
    from oracle import dac_oracle
    ccfg = synth.FULL_CODEC
    sd = synth.codec_state_dict(ccfg, seed=5)
    codes = torch.randint(0, 1024, (B, 9, T), generator=torch.Generator().manual_seed(100 * B + T))
    ref = dac_oracle.decode(sd, codes, ccfg.decoder_rates)
    assert ref.shape == (B, 1, T * 512)
    for precision, tol in (("f16pair", 1e-4), ("f32", 5e-6)):
        got = CodecEngine(ccfg, sd, DEV, precision=precision).decode(codes.to(DEV)).cpu()
        assert got.shape == ref.shape and bool(torch.isfinite(got).all())
        rms = float(((got - ref) ** 2).mean().sqrt())
        assert rms <= tol, (precision, B, T, rms)


@pytest.mark.parametrize("wdtype", ["h2", "f32"])
def test_reference_op_vectors_on_the_trained_like_tiny_checkpoint(golden, wdtype):
    """tests/golden/ops.npz (outputs of the reference's own modules on a 2-layer TRAINED-LIKE checkpoint: non-trivial norm gains,
    heavy-tailed matrices, massive token-embedding channels): the hoisted video MLP equals the reference's AVCLIPEmbedder output, and
    the last-position logits of the same forward (4 video tokens, 12 positions) are within 3e-5 of their scale."""
    g = golden("ops.npz")
    cfg = synth.tiny_sampler(2)
    sd = synth.trained_like(synth.sampler_state_dict(cfg, seed=int(g["weight_seed"]), round_bf16=False), seed=int(g["trained_like_seed"]))
    eng = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    feats = synth.video_features(2, tokens=4, seed=int(g["feat_seed"])).to(DEV)
    idx = torch.from_numpy(g["idx"].astype(np.int64)).to(DEV)
    lg = eng.logits_all_positions(idx, feats).cpu()
    ref = torch.from_numpy(g["logits_last"])
    scale = max(1.0, float(ref.abs().max()))
    err = float((lg[:, :, -1, ::16] - ref).abs().max())
    print(f"reference op vectors [{wdtype}]: last-position logits error {err:.3e} on a scale of {scale:.1f}")
    assert err < 3e-5 * scale, (err, scale)
    cp = eng.cond_projection().cpu()
    assert float((cp - torch.from_numpy(g["cond_proj"])).abs().max()) < 1e-5 * max(1.0, float(np.abs(g["cond_proj"]).max()))


# ----------------------------------------------------------------------------------------------------------------------
# Round 6: the near-tie detector (csrc/step.hip sample_kernel; vaura_sampling.tie_eps; DecoderEngine(near_tie=...)).
def test_near_tie_detector_flags_the_goldens_literal_ties_and_nothing_else(golden, full_sampler_sd_raw, parity_report):
    """The plane storages carry 22-bit operands; a decision whose own margin is inside twice the bound on a logit's error
    (engine.NEAR_TIE_EPS x the row's largest |logit| x (2 cfg - 1)) could have gone the other way in the reference's arithmetic.  The
    sampler COUNTS such used decisions (status bit 4, state[6..7]); it never changes a token.  On the reference's own goldens at full
    depth, default storage: (1) the headline arithmetic's golden (cfg 6, top-k 250 sampled, the reference's noise, B=2: 4 104
    decisions) and the greedy cfg-6 golden (min margin 9.9e-5): NOTHING flagged, tokens strict; (2) the later chunk's greedy run, which
    holds ONE literal tie (clip 1, sequence step 175 = pass 8 of the chunk: 3.8e-6): exactly that step is the first flagged; (3) with
    near_tie="rerun" the flagged call is run again on the exact-fp32 twin (same noise) and still equals the reference; (4) "off": the
    same tokens as "report", nothing counted."""
    g = golden("full_topk250_cfg6_raw_B2_T220.npz")
    eng = DecoderEngine(synth.FULL_SAMPLER, full_sampler_sd_raw, DEV)
    assert eng.wdtype == "h2" and eng.near_tie == "report"
    feats = synth.video_features(2, seed=int(g["feat_seed"])).to(DEV)
    nz = synth.exp_noise(228, 18, 1024, int(g["noise_seed"]))
    tok = eng.generate_codes_checked(feats, 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, noise=nz).cpu()
    assert eng.last_near_ties == (0, None), eng.last_near_ties
    assert_tokens_equal(parity_report, "full_topk250_cfg6_raw_B2_T220", "h2", "near-tie detector on (report): cfg 6 / top-k 250 sampled, B=2; 0 flagged", tok,
                        _ref(g, "tokens"), g["margins"], g["threshold_rel_gap"], near_tie_flagged=0)
    gg = golden("full_greedy_cfg6_raw_B2_T220.npz")
    tokg = eng.generate_codes_checked(synth.video_features(2, seed=int(gg["feat_seed"])).to(DEV), 220, cfg_scale=6.0).cpu()
    assert eng.last_near_ties == (0, None), eng.last_near_ties
    assert torch.equal(tokg, _ref(gg, "tokens"))
    gc = golden("full_chunk_greedy_cfg6_raw_B2_Tp166_T221.npz")
    fc = synth.video_features(2, seed=int(gc["feat_seed"])).to(DEV)
    prompt = torch.from_numpy(gc["prompt"].astype(np.int64)).to(DEV)
    tokc = eng.generate_codes_checked(fc, 221, prompt=prompt, cfg_scale=float(gc["cfg_scale"])).cpu()
    n, first = eng.last_near_ties
    assert n >= 1 and first == 175 - 167, (n, first)             # pass index: the chunk's first sampled pass fills sequence step 167
    assert eng.near_tie_reruns == 0 and eng.range_fallbacks == 0
    refc = _ref(gc, "tokens")
    e = assert_tokens_or_recorded_near_tie(parity_report, "full_chunk_greedy_cfg6_raw_B2_Tp166_T221", "h2",
                                           f"near-tie detector on (report): later chunk, greedy cfg 6, B=2; {n} flagged, first at pass {first} = step 175",
                                           tokc, refc, gc["margins"], 2e-5, first_step=167, near_tie_flagged=n)
    assert e["tokens_equal"] or e["first_diff_step"] == 175, e
    # policy "off": nothing counted, same tokens
    eng.near_tie = "off"
    eng._free_graph()
    tok_off = eng.generate_codes_checked(fc, 221, prompt=prompt, cfg_scale=float(gc["cfg_scale"])).cpu()
    assert eng.last_near_ties == (0, None) and torch.equal(tok_off, tokc)
    # policy "rerun": the flagged call again on the exact-fp32 twin
    eng.near_tie = "rerun"
    eng._free_graph()
    tok_rr = eng.generate_codes_checked(fc, 221, prompt=prompt, cfg_scale=float(gc["cfg_scale"])).cpu()
    assert eng.near_tie_reruns == 1 and eng._range_twin is not None and eng._range_twin.wdtype == "f32"
    e = assert_tokens_or_recorded_near_tie(parity_report, "full_chunk_greedy_cfg6_raw_B2_Tp166_T221", "h2 -> f32 (near_tie='rerun')",
                                           "later chunk, greedy cfg 6, B=2: flagged call re-run on the exact-fp32 twin", tok_rr, refc, gc["margins"], 2e-5,
                                           first_step=167)
    assert e["tokens_equal"] or e["first_diff_step"] == 175, e
    del eng
    torch.cuda.empty_cache()


def test_near_tie_detector_flags_configs3_step_578(golden, parity_report):
    """configs[3]'s golden (B=1, greedy, cfg 1, T=880) holds ONE step inside fp32 summation-order noise (step 578: reference margin
    5.5e-6; the next smallest of its 7 920 decisions is 6.2e-5): the detector flags exactly one decision, at step index 577."""
    g = golden("full_c4_greedy_B1_T880.npz")
    cfg = synth.SamplerCfg(block_size_audio=int(g["block_size_audio"]))
    sd = synth.sampler_state_dict(cfg, seed=int(g["weight_seed"]), round_bf16=True)
    eng = DecoderEngine(cfg, sd, DEV, wdtype="h2")
    feats = synth.video_features(1, tokens=128, seed=int(g["feat_seed"])).to(DEV)
    tok = eng.generate_codes_checked(feats, 880).cpu()
    assert eng.last_near_ties == (1, 577), eng.last_near_ties
    e = assert_tokens_or_recorded_near_tie(parity_report, "full_c4_greedy_B1_T880", "h2", "near-tie detector on (report): configs[3] B=1; 1 flagged, at step 578",
                                           tok, _ref(g, "tokens"), g["margins"], 2e-5, near_tie_flagged=1)
    assert e["tokens_equal"] or e["first_diff_step"] == 578, e
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kw", [dict(), dict(cfg_scale=6.0), dict(cfg_scale=6.0, use_sampling=True, top_k=250, seed=3),
                                dict(use_sampling=True, top_k=40, temp=0.8, seed=4), dict(use_sampling=True, top_p=0.7, seed=5),
                                dict(use_sampling=True, seed=6)])
def test_near_tie_detector_never_changes_a_token(tiny_sampler_sd, kw):
    """Whatever the bound — off, the default, or an absurdly wide one that flags nearly every decision — the tokens are the same."""
    cfg = synth.tiny_sampler(2)
    feats = synth.video_features(5, seed=21).to(DEV)
    out, flagged = [], []
    for mode, eps in (("off", None), ("report", None), ("report", 1e-2)):
        eng = DecoderEngine(cfg, tiny_sampler_sd, DEV, wdtype="h2", near_tie=mode, near_tie_eps=eps)
        out.append(eng.generate_codes_checked(feats, 24, **kw).cpu())
        flagged.append(eng.last_near_ties[0])
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
    assert flagged[0] == 0 and flagged[2] > flagged[1] and flagged[2] >= 20, flagged


# ----------------------------------------------------------------------------------------------------------------------
# Round 6: the fp16 K/V cache (vaura_decoder.kv_dtype = 1; DecoderEngine(kv_dtype="f16")) of the low-precision serving configuration.
@pytest.mark.parametrize("wdtype,kv", [("h1", "f16"), ("fp8h", "f16"), ("fp8h", "f8")])
def test_fp16_kv_cache_tolerance_and_prefill_consistency(wdtype, kv, monkeypatch):
    """kv_dtype="f16": the cache holds fp16(rotated k) / fp16(v); everything else of the attention stays fp32.  No reference counterpart
    (the reference has no cache at all): what is checked is (1) the tolerance against the fp32 cache of the same engine — teacher-forced
    logits move by ~1e-3 of their RMS (reported) and NOT by zero; (2) the two writers / readers of the cache agree: a 40-frame prompt
    teacher-forced through the batched prefill path (rope_append_kernel writes fp16, the MFMA prefill attention widens it) and through
    single decode steps (attention_step256_kernel appends and reads) must leave the same logits at the first sampled position up to
    fp32 summation order; (3) generation runs in range with a clean status word; (4) a cache longer than 256 positions is refused."""
    cfg = synth.tiny_sampler(3)
    sd = synth.sampler_state_dict(cfg, seed=131)
    B = 5
    feats = synth.video_features(2 * B, seed=132).to(DEV)
    idx = torch.randint(0, 1024, (2 * B, 9, 48), generator=torch.Generator().manual_seed(133)).to(DEV)
    e32 = DecoderEngine(cfg, sd, DEV, wdtype=wdtype)
    e16 = DecoderEngine(cfg, sd, DEV, wdtype=wdtype, kv_dtype=kv)
    lg32 = e32.logits_all_positions(idx, feats).cpu()
    lg16 = e16.logits_all_positions(idx, feats).cpu()
    assert e16.kcache.dtype == (torch.float16 if kv == "f16" else torch.uint8) and e16.dec.kv_dtype == (1 if kv == "f16" else 2)
    rel = float((lg16 - lg32).pow(2).mean().sqrt() / lg32.pow(2).mean().sqrt())
    print(f"{kv} K/V cache [{wdtype}]: teacher-forced logits rel-RMS vs the fp32 cache {rel:.3e}, max abs {float((lg16 - lg32).abs().max()):.3e}")
    assert 1e-7 < rel < (5e-3 if kv == "f16" else 8e-2), rel          # ("f8": e4m3 keys and values, 3 mantissa bits: a ~1e-2-class approximation)
    # (position 0 differs too: the new position's own k / v are the fp16 values later steps will read back)
    # (2) prefill writers / readers against the step kernel
    prompt = torch.randint(0, 1024, (B, 9, 40), generator=torch.Generator().manual_seed(134)).to(DEV)
    caches = {}
    for pp in (192, 1):                                             # one batched pass | position by position (decode steps without sampling)
        monkeypatch.setattr(DecoderEngine, "PREFILL_POSITIONS", pp)
        e = DecoderEngine(cfg, sd, DEV, wdtype=wdtype, kv_dtype=kv)
        tok = e.generate_codes(feats[:B], 41, prompt=prompt, cfg_scale=6.0, use_graph=False).cpu()
        e.check_status()
        assert torch.equal(tok[:, :, :40], prompt.cpu()) and int(tok.min()) >= 0 and int(tok.max()) <= 1024
        if kv == "f8":
            caches[pp] = (e.kcache[:, :, :, :40].view(torch.float8_e4m3fn).float().cpu(), e.vcache[:, :, :, :40].view(torch.float8_e4m3fn).float().cpu())
        else:
            caches[pp] = (e.kcache[:, :, :, :40].float().cpu(), e.vcache[:, :, :, :40].float().cpu())
        del e
    # layer 0's K / V depend on the writers only; deeper layers also on the readers (a reader that mis-read the fp16 rows would change
    # every deeper layer's K / V grossly).  The two paths sum in different orders (K-split GEMV + step attention vs GEMM + MFMA prefill
    # attention): fp16 values may differ by an ulp here and there, nothing more.
    for a, b in zip(caches[192], caches[1]):
        same = float((a == b).float().mean())
        worst = float((a - b).abs().max() / a.abs().max())              # in units of the cache's largest value (an fp16 ulp there is 1e-3)
        print(f"{kv} K/V cache [{wdtype}]: batched prefill vs single steps: {same:.5f} of the cached values identical, worst difference {worst:.2e} of the largest value")
        # ("fp8h": a prompt pass multiplies BOTH activation planes — the exact fp8 arithmetic —, a decode step the hi plane only: the values
        # differ in the last fp16 bit about half the time, by design; the bound on the size of a difference is the same)
        assert same > (0.97 if wdtype == "h1" else 0.3) and worst < (2e-3 if kv == "f16" else 7e-2), (same, worst)      # (an e4m3 ulp at the top of the range is 1/16)
    # (3) a sampled run, (4) the refusal
    monkeypatch.setattr(DecoderEngine, "PREFILL_POSITIONS", 192)
    tok = e16.generate_codes_checked(feats[:B], 30, cfg_scale=6.0, use_sampling=True, top_k=250, seed=5).cpu()
    assert int(tok.min()) >= 0 and int(tok.max()) < 1024
    with pytest.raises(L_VauraHipError()):
        DecoderEngine(synth.SamplerCfg(num_layers=2, block_size_audio=1024), synth.sampler_state_dict(synth.SamplerCfg(num_layers=2, block_size_audio=1024), seed=1),
                      DEV, wdtype="h1", kv_dtype=kv).generate_codes(synth.video_features(1, tokens=128, seed=2).to(DEV), 300)


def L_VauraHipError():
    from vaura_amd import _lib as L
    return L.VauraHipError
