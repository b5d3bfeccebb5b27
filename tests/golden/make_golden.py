"""Generate tests/golden/*.npz by running the REAL reference (build container only).

    python tests/golden/make_golden.py small         # patterns, sampler, tiny-model cases (seconds)
    python tests/golden/make_golden.py ops           # op-level outputs of the reference's own modules on a 2-layer trained-like checkpoint (seconds)
    python tests/golden/make_golden.py full_greedy   # 24-layer model, B=2, T=220, greedy   (~4 min)
    python tests/golden/make_golden.py full_sample   # 24-layer, B=2, cfg 6, top-k 250      (~8 min)
    python tests/golden/make_golden.py full_greedy_raw   # full_greedy on the UN-rounded checkpoint (~4 min)
    python tests/golden/make_golden.py full_sample_raw   # UN-rounded checkpoint, B=2, cfg 6, top-k 250 sampled (~8 min)
    python tests/golden/make_golden.py full_greedy_cfg6_raw   # UN-rounded checkpoint, B=2, cfg 6, greedy      (~8 min)
    python tests/golden/make_golden.py full_vgg_raw      # UN-rounded checkpoint, configs/generate_vgg.yaml:23-27 defaults (top-k 128, cfg 6) (~8 min)
    python tests/golden/make_golden.py full_chunk_raw    # a LATER chunk of the sliding-window caller (scripts/generate.py:344-357): Tp=166, T=221,
                                                         # cfg 6, top-k 128 sampled + greedy, UN-rounded checkpoint (~8 min)
    python tests/golden/make_golden.py avclip        # Segment-AVCLIP extractor (row f2), reference classes, 1 + 4 segments (~1 min)
    python tests/golden/make_golden.py full_c4       # configs[3]: block_size 1024, Tv=128, B=1, T=880 (~30 min)
    python tests/golden/make_golden.py codec         # DAC decode, transformers' DacModel   (seconds)
    python tests/golden/make_golden.py post          # post-codec audio scaling (row f3)    (seconds)
    python tests/golden/make_golden.py codec_enc     # DAC encode, transformers' DacModel   (seconds)
    python tests/golden/make_golden.py codec_full    # both at the 44.1 kHz model's FULL width (~1 min)

Inputs are never stored when they can be regenerated: weights and features come from
``vaura_amd.synth`` (name-keyed seeds), sampling noise from ``synth.exp_noise(seed)``.
Each fixture records the seeds/arguments needed to rebuild its inputs.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from vaura_amd import synth  # noqa: E402
import ref_harness as rh  # noqa: E402


def sha1(a: np.ndarray) -> str:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------- patterns
def gold_patterns():
    rh.install()
    from models.modules.misc.codebook_patterns import DelayedPatternProvider
    out = {}
    prov = DelayedPatternProvider(n_q=9)
    for T, Tp in [(4, 0), (55, 0), (220, 0), (221, 166), (20, 8)]:
        pat = prov.get_pattern(T)
        g = torch.Generator().manual_seed(100 + T)
        codes = torch.full((2, 9, T), -1, dtype=torch.long)
        if Tp:
            codes[..., :Tp] = torch.randint(0, 1024, (2, 9, Tp), generator=g)
        seq, idx, mask = pat.build_pattern_sequence(codes, 1024)
        filled = torch.where(seq == -1, torch.randint(0, 1024, seq.shape, generator=g), seq)
        filled = torch.where(mask[None], filled, torch.full_like(filled, 1024))
        rev, ridx, rmask = pat.revert_pattern_sequence(filled, special_token=-1)
        k = f"T{T}_p{Tp}"
        out[k + "_codes"] = codes.numpy().astype(np.int16)
        out[k + "_seq"] = seq.numpy().astype(np.int16)
        out[k + "_idx"] = idx.numpy().astype(np.int32)
        out[k + "_mask"] = mask.numpy()
        out[k + "_filled"] = filled.numpy().astype(np.int16)
        out[k + "_rev"] = rev.numpy().astype(np.int16)
        out[k + "_ridx"] = ridx.numpy().astype(np.int32)
        out[k + "_rmask"] = rmask.numpy()
        out[k + "_first"] = np.int64(pat.get_first_step_with_timesteps(Tp))
        # revert_pattern_logits (train-time view of the logits, codebook_patterns.py:287-313): S - 1 output positions, NaN fill
        S = seq.shape[-1]
        lg = torch.randn(2, 3, 9, S - 1, generator=g)
        lv, lidx, lmask = pat.revert_pattern_logits(lg, float("nan"))
        out[k + "_lg"] = lg.numpy()
        out[k + "_lg_rev"] = lv.numpy()
        out[k + "_lg_idx"] = lidx.numpy().astype(np.int32)
        out[k + "_lg_mask"] = lmask.numpy()
    save("patterns.npz", **out)


# ------------------------------------------------------------------------------------- sampling
def gold_sampling():
    rh.install()
    from utils.utils import multinomial, sample_top_k, sample_top_p
    out = {}
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(3, 9, 1024, generator=g) * 1.5
    # exact ties: duplicate a few logits so that the k-th largest value is shared
    logits[0, 0, 10:20] = logits[0, 0, 5]
    logits[1, 3, :] = logits[1, 3, :].round(decimals=1)
    out["logits"] = logits.numpy()
    seed = 4242
    cases = [("topk1", dict(k=1)), ("topk128", dict(k=128)), ("topk250", dict(k=250)),
             ("topp90", dict(p=0.9)), ("topp30", dict(p=0.3)), ("plain", dict())]
    for temp in (1.0, 0.7):
        for name, kw in cases:
            probs = torch.softmax(logits / temp, dim=-1)
            torch.manual_seed(seed)
            if "k" in kw:
                tok = sample_top_k(probs.clone(), kw["k"])
            elif "p" in kw:
                tok = sample_top_p(probs.clone(), kw["p"])
            else:
                tok = multinomial(probs.clone(), num_samples=1)
            # the noise the call consumed, re-drawn from the same seed (one draw of (rows, V))
            noise = synth.exp_noise(1, 27, 1024, seed)[0]
            chk = torch.argmax(probs.reshape(27, 1024) / noise, -1)
            if name == "plain":
                assert torch.equal(chk, tok.reshape(-1)), "multinomial != argmax(p/Exp(1))"
            out[f"{name}_t{temp}_tok"] = tok.numpy().astype(np.int16)
    out["noise_seed"] = np.int64(seed)
    save("sampling.npz", **out)


# ------------------------------------------------------------------------------------- tiny model
def _capture_logits(model, store, rows=None):
    def hook(_m, _inp, outp):
        lg = outp[0]  # (Bs, K, L, V)
        store.append((lg.shape[2], lg[:, :, -1, :].detach().clone()))
    return model.sampler.register_forward_hook(hook)


def gold_tiny():
    cfg = synth.tiny_sampler(2)
    sd = synth.sampler_state_dict(cfg, seed=3)
    model = rh.build_reference_model(cfg.yaml_params(), sd)
    out = {"layers": np.int64(2), "weight_seed": np.int64(3), "feat_seed": np.int64(5)}
    feats = synth.video_features(2, seed=5)
    frames = feats.reshape(2, 4, 8, 768)

    # (1) full forward logits on a random sequence (all-position check of the decoder itself)
    g = torch.Generator().manual_seed(11)
    idx = torch.randint(0, 1025, (2, 9, 12), generator=g)
    with torch.no_grad():
        lg, _, _ = model.sampler(tgt=idx, memory=feats)
    out["fwd_idx"] = idx.numpy().astype(np.int16)
    out["fwd_logits_pos"] = np.array([0, 6, 7, 11])
    out["fwd_logits"] = lg[:, :, [0, 6, 7, 11], :].numpy()

    # (1b) positions past Tv*7 read empty_video_emb: 4 video tokens -> positions >= 28
    feats4 = feats[:, :4]
    idx2 = torch.randint(0, 1025, (1, 9, 31), generator=g)
    with torch.no_grad():
        lg2, _, _ = model.sampler(tgt=idx2, memory=feats4[:1])
    out["pad_idx"] = idx2.numpy().astype(np.int16)
    out["pad_logits"] = lg2[:, :, [27, 28, 30], :].numpy()

    # (2) greedy generate, T=20
    r = model.generate(frames=frames, audio=None, max_new_tokens=20, return_sampled_indices=True,
                       use_sampling=False, prompt_is_encoded=True, cfg_scale=1.0)
    out["greedy_T20"] = r["sampled_indices"].numpy().astype(np.int16)

    # (3) greedy + CFG 6.0
    r = model.generate(frames=frames, audio=None, max_new_tokens=20, return_sampled_indices=True,
                       use_sampling=False, prompt_is_encoded=True, cfg_scale=6.0)
    out["greedy_cfg6_T20"] = r["sampled_indices"].numpy().astype(np.int16)

    # (4) sampling: top-k 250, cfg 6, temp 1.0, seed 99
    torch.manual_seed(99)
    r = model.generate(frames=frames, audio=None, max_new_tokens=20, return_sampled_indices=True,
                       use_sampling=True, temp=1.0, top_k=250, top_p=0.0, prompt_is_encoded=True, cfg_scale=6.0)
    out["topk250_cfg6_seed99_T20"] = r["sampled_indices"].numpy().astype(np.int16)

    # (5) sampling: top-p 0.8, temp 0.9, cfg 1, seed 98
    torch.manual_seed(98)
    r = model.generate(frames=frames, audio=None, max_new_tokens=20, return_sampled_indices=True,
                       use_sampling=True, temp=0.9, top_k=250, top_p=0.8, prompt_is_encoded=True, cfg_scale=1.0)
    out["topp80_t09_seed98_T20"] = r["sampled_indices"].numpy().astype(np.int16)

    # (6) plain multinomial, seed 97
    torch.manual_seed(97)
    r = model.generate(frames=frames, audio=None, max_new_tokens=20, return_sampled_indices=True,
                       use_sampling=True, temp=1.0, top_k=0, top_p=0.0, prompt_is_encoded=True, cfg_scale=1.0)
    out["plain_seed97_T20"] = r["sampled_indices"].numpy().astype(np.int16)

    # (7) prompt continuation (the sliding-window caller's shape, scaled down): Tp=8 of T=20
    prompt = torch.from_numpy(out["greedy_T20"].astype(np.int64))[:, :, 5:13]
    r = model.generate(frames=frames, audio=prompt, max_new_tokens=20, return_sampled_indices=True,
                       use_sampling=False, prompt_is_encoded=True, cfg_scale=1.0, remove_prompts=False)
    out["prompt8_greedy_T20"] = r["sampled_indices"].numpy().astype(np.int16)
    save("tiny_model.npz", **out)


def gold_ops():
    """op-level vectors from the reference's OWN modules — RMSNorm, precompute_freqs_cis / apply_rotary_emb,
    Attention.forward, FeedForward, TransformerBlock, AVCLIPEmbedder, DacEmbeddingProjection, _repeat_and_pad_video — captured by
    forward hooks during ONE Transformer.forward of the 2-layer model (full width) on the trained-like checkpoint (non-trivial norm
    gains, heavy tails: synth.trained_like), plus direct calls of the free functions.  Channels are sub-sampled (every 8th) to keep
    the fixture small; inputs are regenerated from seeds."""
    cfg = synth.tiny_sampler(2)
    sd = synth.trained_like(synth.sampler_state_dict(cfg, seed=3, round_bf16=False), seed=11)
    model = rh.build_reference_model(cfg.yaml_params(), sd)
    from models.modules.sampler.llama import apply_rotary_emb, precompute_freqs_cis
    smp = model.sampler
    feats = synth.video_features(2, tokens=4, seed=21)                    # 4 video tokens: positions >= 28 read empty_video_emb
    idx = torch.randint(0, 1025, (2, 9, 12), generator=torch.Generator().manual_seed(22))
    taps = {}

    def tap(name, with_input=False):
        def hook(_m, inp, out):
            taps[name] = out.detach().clone()
            if with_input:
                taps[name + "_in"] = inp[0].detach().clone()
        return hook
    hooks = [smp.layers[0].attention_norm.register_forward_hook(tap("rmsnorm", True)),
             smp.layers[0].attention.register_forward_hook(tap("attention", True)),
             smp.layers[0].feed_forward.register_forward_hook(tap("ffn", True)),
             smp.layers[0].register_forward_hook(tap("block0", True)),
             smp.layers[1].register_forward_hook(tap("block1")),
             smp.cls_embeddings.register_forward_hook(tap("cond_proj")),
             smp.tok_embeddings[0].register_forward_hook(tap("tok_emb0")),
             smp.tok_embeddings[8].register_forward_hook(tap("tok_emb8"))]
    with torch.no_grad():
        logits, _, _ = smp(tgt=idx, memory=feats)
        padded = smp._repeat_and_pad_video(taps["cond_proj"], 31)
        fc = precompute_freqs_cis(40, 96, 10000)
        x = torch.randn(2, 5, 16, 96, generator=torch.Generator().manual_seed(23))
        rot = apply_rotary_emb(x, fc[:5])
    for h in hooks:
        h.remove()
    assert torch.equal(taps["rmsnorm_in"], taps["block0_in"]) and torch.equal(taps["attention_in"], taps["rmsnorm"])
    sub = lambda t: t[..., ::8].numpy()
    save("ops.npz", layers=np.int64(2), weight_seed=np.int64(3), trained_like_seed=np.int64(11), feat_seed=np.int64(21), idx_seed=np.int64(22),
         rope_seed=np.int64(23), idx=idx.numpy().astype(np.int16),
         # block0_in is also RMSNorm's input; RMSNorm's output is also Attention.forward's input
         block0_in=taps["block0_in"].numpy(), rmsnorm=taps["rmsnorm"].numpy(), attention=sub(taps["attention"]),
         ffn_in=taps["ffn_in"].numpy(), ffn=sub(taps["ffn"]), block0=sub(taps["block0"]), block1=sub(taps["block1"]),
         cond_proj=taps["cond_proj"].numpy(), padded_video=padded[:, [0, 6, 7, 27, 28, 30]].numpy(),
         tok_emb0=sub(taps["tok_emb0"]), tok_emb8=sub(taps["tok_emb8"]),
         freqs_cis=torch.view_as_real(fc).numpy() if fc.is_complex() else fc.numpy(), rope_out=rot.numpy(),
         logits_last=logits[:, :, -1, ::16].numpy())


# ------------------------------------------------------------------------------------- full size
def _full_model(round_bf16: bool = True, cfg: synth.SamplerCfg = None):
    cfg = cfg or synth.FULL_SAMPLER
    t = time.time()
    sd = synth.sampler_state_dict(cfg, seed=0, round_bf16=round_bf16)
    print(f"weights in {time.time() - t:.1f}s")
    return rh.build_reference_model(cfg.yaml_params(), sd)


def _greedy_run(model, name, B, T, Tv, keep, **extra):
    """Greedy, cfg 1.0 generate() of the reference with the last-position logits captured every step."""
    feats = synth.video_features(B, tokens=Tv, seed=0)
    store = []
    h = _capture_logits(model, store)
    t = time.time()
    r = model.generate(frames=feats.reshape(B, Tv // 8, 8, 768), audio=None, max_new_tokens=T,
                       return_sampled_indices=True, use_sampling=False, prompt_is_encoded=True, cfg_scale=1.0)
    dt = time.time() - t
    h.remove()
    tok = r["sampled_indices"].numpy()
    logits = {L: lg for (L, lg) in store}
    margins = []
    for (L, lg) in store:
        top2 = torch.topk(lg, 2, dim=-1).values
        margins.append((top2[..., 0] - top2[..., 1]).numpy())
    margins = np.stack(margins)  # (steps, B, K)
    save(name,
         tokens=tok.astype(np.int16), sha1=np.array(sha1(tok.astype(np.int16))),
         logits_steps=np.array(keep), logits=np.stack([logits[L].numpy() for L in keep]),
         margins=margins.astype(np.float32), ref_seconds=np.float64(dt),
         ref_threads=np.int64(torch.get_num_threads()), weight_seed=np.int64(0), feat_seed=np.int64(0), **extra)
    print(f"reference generate(): {dt:.1f}s  min margin {margins.min():.3e}")


def gold_full_greedy():
    _greedy_run(_full_model(), "full_greedy_B2_T220.npz", 2, 220, 32, [1, 8, 9, 10, 100, 224, 225, 228])


def gold_full_greedy_raw():
    """Same run on the UN-rounded synthetic checkpoint (fp32 weights that bf16 cannot hold: what a real
    V-AURA checkpoint looks like).  The f32-storage HIP path must reproduce these tokens; the bf16-storage path
    rounds 694 M weights and is only required to report its agreement (SURVEY.md §7 'Hard parts')."""
    _greedy_run(_full_model(round_bf16=False), "full_greedy_raw_B2_T220.npz", 2, 220, 32,
                [1, 8, 9, 10, 100, 224, 225, 228], round_bf16=np.int64(0))


def gold_full_c4():
    """BASELINE configs[3] (10.24 s single pass): the reference Transformer built with block_size_audio=1024 (its
    RoPE table is a pure function of the position, llama.py:593-603), 128 video tokens, cfg 1.0 (the CFG null
    embedding is fixed at 32 tokens, vaura_model.py:790-793), B=1, greedy, T=880 -> 888 cache-less passes."""
    cfg = synth.SamplerCfg(block_size_audio=1024)
    _greedy_run(_full_model(cfg=cfg), "full_c4_greedy_B1_T880.npz", 1, 880, 128,
                [1, 9, 10, 228, 229, 257, 444, 600, 887, 888], block_size_audio=np.int64(1024))


def gold_full_sample():
    model = _full_model()
    B = 2
    feats = synth.video_features(B, seed=0)
    torch.manual_seed(2024)
    t = time.time()
    r = model.generate(frames=feats.reshape(B, 4, 8, 768), audio=None, max_new_tokens=220,
                       return_sampled_indices=True, use_sampling=True, temp=1.0, top_k=250, top_p=0.0,
                       prompt_is_encoded=True, cfg_scale=6.0)
    dt = time.time() - t
    tok = r["sampled_indices"].numpy()
    save("full_topk250_cfg6_B2_T220.npz", tokens=tok.astype(np.int16), sha1=np.array(sha1(tok.astype(np.int16))),
         noise_seed=np.int64(2024), ref_seconds=np.float64(dt), weight_seed=np.int64(0), feat_seed=np.int64(0))
    print(f"reference generate(): {dt:.1f}s")


def _cfg_run(name, B, cfg_scale, use_sampling, top_k, noise_seed, T=220, prompt=None, feat_seed=0, keep=(1, 9, 10, 100, 228), model=None):
    """CFG generate() of the reference on the UN-rounded checkpoint (what a real fp32 V-AURA checkpoint looks like to the
    storage decision: 'auto' -> two fp16 planes) with every step's [cond; null] last-position logits captured.  Recorded per
    step, clip and codebook: the CFG-mixed top-1 / top-2 logit margin (greedy) or, for top-k sampling, the relative margin of
    argmax(p / Exp(1)) over the kept set and the relative gap at the top-k threshold — the only places where a storage
    format's logit error can change a token."""
    model = model or _full_model(round_bf16=False)
    feats = synth.video_features(B, seed=feat_seed)
    store = []
    h = _capture_logits(model, store)
    if use_sampling:
        torch.manual_seed(noise_seed)
    t = time.time()
    r = model.generate(frames=feats.reshape(B, 4, 8, 768), audio=prompt, max_new_tokens=T,
                       return_sampled_indices=True, use_sampling=use_sampling, temp=1.0, top_k=top_k, top_p=0.0,
                       prompt_is_encoded=True, cfg_scale=cfg_scale, remove_prompts=False)
    dt = time.time() - t
    h.remove()
    tok = r["sampled_indices"].numpy()
    noise = synth.exp_noise(len(store), B * 9, 1024, noise_seed) if use_sampling else None
    margins, thr_gap = [], []
    for i, (L, lg) in enumerate(store):
        assert lg.shape[0] == 2 * B
        c, u = lg[:B], lg[B:]
        mixed = u + (c - u) * cfg_scale                                   # vaura_model.py:810-813
        if not use_sampling:
            top2 = torch.topk(mixed, 2, dim=-1).values
            margins.append((top2[..., 0] - top2[..., 1]).numpy())
            continue
        p = torch.softmax(mixed, -1)
        srt = torch.sort(p, dim=-1, descending=True).values
        thr = srt[..., top_k - 1:top_k]
        thr_gap.append(((srt[..., top_k - 1] - srt[..., top_k]) / srt[..., top_k - 1]).numpy())
        kept = torch.where(p >= thr, p, torch.zeros_like(p))
        ratio = kept / noise[i].reshape(B, 9, 1024)
        top2 = torch.topk(ratio, 2, dim=-1).values
        margins.append(((top2[..., 0] - top2[..., 1]) / top2[..., 0]).numpy())
    extra = {}
    if use_sampling:
        extra = dict(noise_seed=np.int64(noise_seed), threshold_rel_gap=np.stack(thr_gap).astype(np.float32))
    if prompt is not None:
        extra["prompt"] = prompt.numpy().astype(np.int16)
    keep = list(keep)
    logits = {L: lg for (L, lg) in store}
    assert [L for (L, _) in store] == list(range(1 if prompt is None else prompt.shape[-1] + 1, T + 9)), "one pass per sequence step"
    save(name, tokens=tok.astype(np.int16), sha1=np.array(sha1(tok.astype(np.int16))),
         margins=np.stack(margins).astype(np.float32), logits_steps=np.array(keep),
         logits=np.stack([logits[L].numpy() for L in keep]), cfg_scale=np.float64(cfg_scale), top_k=np.int64(top_k),
         ref_seconds=np.float64(dt), weight_seed=np.int64(0), feat_seed=np.int64(feat_seed), round_bf16=np.int64(0), **extra)
    print(f"reference generate(): {dt:.1f}s  min margin {np.stack(margins).min():.3e}")
    return model


def gold_full_sample_raw():
    """The headline configuration's arithmetic (configs/generate_vgg.yaml:23-27 sampling with configs[1]'s top-k): un-rounded
    checkpoint, cfg 6, top-k 250 sampled, B=2, the reference's own noise stream."""
    _cfg_run("full_topk250_cfg6_raw_B2_T220.npz", 2, 6.0, True, 250, 2025)


def gold_full_greedy_cfg6_raw():
    _cfg_run("full_greedy_cfg6_raw_B2_T220.npz", 2, 6.0, False, 0, 0)


def gold_full_vgg_raw():
    """The reference's shipped sampling defaults (configs/generate_vgg.yaml:23-27: use_sampling, temperature 1.0, top_k 128,
    top_p 0.0, cfg_scale 6.0) at full depth on the un-rounded checkpoint, B=2, the reference's own noise stream."""
    _cfg_run("full_topk128_cfg6_raw_B2_T220.npz", 2, 6.0, True, 128, 2027)


def gold_full_chunk_raw():
    """A LATER chunk of the sliding-window caller at full depth (scripts/generate.py:327-369): max_gen_len = ceil(2.56 * 44100 / 512)
    = 221, the prompt = the previous chunk's tokens from stride 55 on = 166 tokens (`prompt_tokens = gen_tokens[:, :, stride_tokens:]`,
    :365), remove_prompts=False, prompt_is_encoded=True -> start_offset_sequence 167, 63 cache-less passes of 167..229 positions.
    The prompt is the tail of the headline golden's own sampled tokens (frames 54..219 of full_topk250_cfg6_raw_B2_T220: 166 frames
    the same model produced), the features of the chunk are a fresh seed.  Two runs on ONE model build: generate_vgg.yaml's
    defaults (cfg 6, top-k 128 sampled, the reference's noise stream) and greedy under cfg 6."""
    prev = np.load(os.path.join(HERE, "full_topk250_cfg6_raw_B2_T220.npz"))["tokens"].astype(np.int64)
    prompt = torch.from_numpy(prev[:, :, 54:220]).contiguous()
    assert prompt.shape == (2, 9, 166)
    keep = (167, 168, 200, 229)
    m = _cfg_run("full_chunk_topk128_cfg6_raw_B2_Tp166_T221.npz", 2, 6.0, True, 128, 2026, T=221, prompt=prompt, feat_seed=1, keep=keep)
    _cfg_run("full_chunk_greedy_cfg6_raw_B2_Tp166_T221.npz", 2, 6.0, False, 0, 0, T=221, prompt=prompt, feat_seed=1, keep=keep, model=m)


# ------------------------------------------------------------------------------------- codec
def gold_codec(full: bool = False):
    """DAC decode golden from transformers' independent DacModel (NOT the reference's dependency:
    structure cross-check only — see oracle/__init__.py 'parity unpinned').  full=True: the 44.1 kHz model's real
    width (decoder 1536 -> 96 channels, 54 M parameters), where the HIP kernels take their full-size tile paths."""
    from transformers import DacConfig, DacModel
    from transformers.models.dac import modeling_dac  # noqa: F401
    ccfg = synth.FULL_CODEC if full else synth.CodecCfg(decoder_dim=192, decoder_rates=(8, 8, 4, 2))
    hf = DacConfig(sampling_rate=44100, decoder_hidden_size=ccfg.decoder_dim, upsampling_ratios=list(ccfg.decoder_rates),
                   n_codebooks=9, codebook_size=1024, codebook_dim=8, hidden_size=ccfg.latent_dim)
    m = DacModel(hf).eval()
    sd = synth.codec_state_dict(ccfg, seed=1)
    # load folded weights into HF's plain convs (HF names differ; map by module order)
    dec = m.decoder
    fold = synth.fold_weight_norm

    def put(conv, prefix):
        conv.weight.data.copy_(fold(sd[prefix + "weight_g"], sd[prefix + "weight_v"]))
        conv.bias.data.copy_(sd[prefix + "bias"])

    put(dec.conv1, "decoder.model.0.")
    for b, blk in enumerate(dec.block):
        p = f"decoder.model.{b + 1}.block."
        blk.snake1.alpha.data.copy_(sd[p + "0.alpha"])
        put(blk.conv_t1, p + "1.")
        for u, ru in enumerate([blk.res_unit1, blk.res_unit2, blk.res_unit3]):
            q = p + f"{u + 2}.block."
            ru.snake1.alpha.data.copy_(sd[q + "0.alpha"])
            put(ru.conv1, q + "1.")
            ru.snake2.alpha.data.copy_(sd[q + "2.alpha"])
            put(ru.conv2, q + "3.")
    n = len(ccfg.decoder_rates) + 1
    dec.snake1.alpha.data.copy_(sd[f"decoder.model.{n}.alpha"])
    put(dec.conv2, f"decoder.model.{n + 1}.")
    for i, q in enumerate(m.quantizer.quantizers):
        p = f"quantizer.quantizers.{i}."
        q.codebook.weight.data.copy_(sd[p + "codebook.weight"])
        put(q.out_proj, p + "out_proj.")
    g = torch.Generator().manual_seed(21)
    codes = torch.randint(0, 1024, (2, 9, 12 if full else 24), generator=g)
    with torch.no_grad():
        z = m.quantizer.from_codes(codes)[0]
        wav = m.decoder(z)
    if full:   # the latent is not stored at full width (regenerable; 100 KB of waveform is the evidence)
        save("codec_hf_full.npz", codes=codes.numpy().astype(np.int16), wav=wav.numpy(), decoder_dim=np.int64(ccfg.decoder_dim),
             codec_seed=np.int64(1))
    else:
        save("codec_hf.npz", codes=codes.numpy().astype(np.int16), z=z.numpy(), wav=wav.numpy(),
             decoder_dim=np.int64(ccfg.decoder_dim), codec_seed=np.int64(1))


def gold_codec_enc(full: bool = False):
    """DAC encode cross-check with transformers' independent DacModel (reduced width: encoder 8 -> 128 = latent; full=True:
    the real 64 -> 1024 encoder): encoder latent + codes.  Structure check only, like gold_codec (the reference's own
    dependency is absent)."""
    from transformers import DacConfig, DacModel
    ccfg = synth.FULL_CODEC if full else synth.CodecCfg(latent_dim=128, encoder_dim=8, encoder_rates=(2, 4, 8, 8), decoder_dim=192)
    hf = DacConfig(sampling_rate=44100, encoder_hidden_size=ccfg.encoder_dim, downsampling_ratios=list(ccfg.encoder_rates),
                   decoder_hidden_size=ccfg.decoder_dim, upsampling_ratios=list(ccfg.decoder_rates),
                   n_codebooks=9, codebook_size=1024, codebook_dim=8, hidden_size=ccfg.latent_dim)
    m = DacModel(hf).eval()
    sd = dict(synth.codec_state_dict(ccfg, seed=2))
    sd.update(synth.codec_encoder_state_dict(ccfg, seed=2))
    fold = synth.fold_weight_norm

    def put(conv, prefix):
        conv.weight.data.copy_(fold(sd[prefix + "weight_g"], sd[prefix + "weight_v"]))
        conv.bias.data.copy_(sd[prefix + "bias"])

    enc = m.encoder
    put(enc.conv1, "encoder.block.0.")
    for b, blk in enumerate(enc.block):
        p = f"encoder.block.{b + 1}.block."
        for u, ru in enumerate([blk.res_unit1, blk.res_unit2, blk.res_unit3]):
            q = p + f"{u}.block."
            ru.snake1.alpha.data.copy_(sd[q + "0.alpha"])
            put(ru.conv1, q + "1.")
            ru.snake2.alpha.data.copy_(sd[q + "2.alpha"])
            put(ru.conv2, q + "3.")
        blk.snake1.alpha.data.copy_(sd[p + "3.alpha"])
        put(blk.conv1, p + "4.")
    n = len(ccfg.encoder_rates) + 1
    enc.snake1.alpha.data.copy_(sd[f"encoder.block.{n}.alpha"])
    put(enc.conv2, f"encoder.block.{n + 1}.")
    for i, q in enumerate(m.quantizer.quantizers):
        p = f"quantizer.quantizers.{i}."
        q.codebook.weight.data.copy_(sd[p + "codebook.weight"])
        put(q.in_proj, p + "in_proj.")
        put(q.out_proj, p + "out_proj.")
    g = torch.Generator().manual_seed(22)
    wav = torch.randn(2, 1, 512 * 6, generator=g) * 0.3
    with torch.no_grad():
        z = enc(wav)
        codes = m.quantizer(z)[1]
    save("codec_enc_hf_full.npz" if full else "codec_enc_hf.npz", wav=wav.numpy(), z=z.numpy(), codes=codes.numpy().astype(np.int16),
         codec_seed=np.int64(2), encoder_dim=np.int64(ccfg.encoder_dim), latent_dim=np.int64(ccfg.latent_dim))


def gold_avclip():
    """Row f2: the reference's own MotionFormer (divided space-time ViT-B/16 + spatial aggregation layer) on ONE seeded
    16-frame segment at full width with seeded weights (synth.avclip_state_dict): the (1, 1, 8, 768) features plus three
    intermediate token tensors (sub-sampled) for stage-level checks.  Inputs are regenerated from seeds, not stored."""
    sd = synth.avclip_state_dict(seed=0)
    m = rh.build_reference_motionformer(sd)
    frames = synth.video_frames(1, 1, seed=0)
    taps = {}
    hooks = [m.blocks[0].register_forward_hook(lambda _m, _i, o: taps.__setitem__("block0", o.detach().clone())),
             m.blocks[-1].register_forward_hook(lambda _m, _i, o: taps.__setitem__("block11", o.detach().clone())),
             m.pos_drop.register_forward_hook(lambda _m, _i, o: taps.__setitem__("tokens", o.detach().clone()))]
    t = time.time()
    with torch.no_grad():
        feats, glob = m(frames)
    dt = time.time() - t
    for h in hooks:
        h.remove()
    assert glob is None and feats.shape == (1, 1, 8, 768)
    rows = np.array([0, 1, 2, 197, 198, 785, 1568])
    save("avclip.npz", feats=feats.numpy(), rows=rows, tokens=taps["tokens"][0, rows].numpy(), block0=taps["block0"][0, rows].numpy(),
         block11=taps["block11"][0, rows].numpy(), weight_seed=np.int64(0), frame_seed=np.int64(0), ref_seconds=np.float64(dt))
    # two segments of two clips through the batched path (for_loop=False) for the layout of (B, S)
    frames2 = synth.video_frames(2, 2, seed=1)
    with torch.no_grad():
        f2, _ = m(frames2)
    save("avclip_b2s2.npz", feats=f2.numpy(), frame_seed=np.int64(1), weight_seed=np.int64(0))
    print(f"reference MotionFormer: {dt:.1f}s per segment")


def gold_post():
    """Post-codec scaling: the reference's own normalize_audio (utils/data_utils.py:407-466)
    on seeded waveforms: loud (peaks > 1), nominal, quiet; 'clip' (the configs' default), 'peak' and 'rms'
    with normalize True / False."""
    rh.install()
    cwd = os.getcwd()
    os.chdir(rh.REFERENCE_ROOT)
    from utils.data_utils import normalize_audio
    os.chdir(cwd)
    out = {}
    gains = {"loud": 1.7, "nominal": 0.3, "quiet": 0.01}
    n = 4096
    for name, gain in gains.items():
        g = torch.Generator().manual_seed(zlib_seed(name))
        t = torch.arange(n) / 44100.0
        wav = (torch.sin(2 * torch.pi * 440.0 * t) * 0.6 + torch.randn(n, generator=g) * 0.25)[None] * gain
        out[f"{name}_in"] = wav.numpy()
        for strategy, normalize, db in (("clip", True, 6.0), ("clip", True, 3.0), ("peak", True, 6.0), ("peak", False, 6.0),
                                        ("rms", True, 6.0), ("rms", False, 6.0)):
            o = normalize_audio(wav.clone(), normalize=normalize, strategy=strategy, peak_clip_headroom_db=db)
            out[f"{name}_{strategy}_n{int(normalize)}_db{int(db)}"] = o.numpy()
    save("post.npz", **out)


def zlib_seed(name: str) -> int:
    import zlib
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


if __name__ == "__main__":
    torch.set_float32_matmul_precision("highest")  # NOT main.py:34's "medium" 
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what == "small":
        gold_patterns(); gold_sampling(); gold_tiny()
    elif what == "ops":
        gold_ops()
    elif what == "full_greedy":
        gold_full_greedy()
    elif what == "full_greedy_raw":
        gold_full_greedy_raw()
    elif what == "full_c4":
        gold_full_c4()
    elif what == "avclip":
        gold_avclip()
    elif what == "full_sample":
        gold_full_sample()
    elif what == "full_sample_raw":
        gold_full_sample_raw()
    elif what == "full_greedy_cfg6_raw":
        gold_full_greedy_cfg6_raw()
    elif what == "full_vgg_raw":
        gold_full_vgg_raw()
    elif what == "full_chunk_raw":
        gold_full_chunk_raw()
    elif what == "codec":
        gold_codec()
    elif what == "codec_full":
        gold_codec(full=True); gold_codec_enc(full=True)
    elif what == "post":
        gold_post()
    elif what == "codec_enc":
        gold_codec_enc()
    else:
        raise SystemExit(f"unknown target {what}")
