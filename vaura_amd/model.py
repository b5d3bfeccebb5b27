"""Generation-path mirror of the reference's ``VAURAModel`` (/root/reference/models/vaura_model.py).

Same plugin slots and constructor keywords (:28-120), same ``generate()`` keywords and result dict
(:410-597) and same ``_sample_next_token()`` signature (:775-827), so ``scripts/generate.py``-style
callers and ``configs/generate_*.yaml`` work once the ``target:`` strings point at ``vaura_amd``.
It is a plain ``nn.Module`` (inference only — the Lightning training half is out of scope) and it
does no arithmetic itself: conditioning, the 228-step decode loop, sampling, pattern bookkeeping and
codec decode all run in libvaura_hip.so.

What differs from the reference, on purpose:
  * the hot loop uses a K/V cache and runs entirely on the device (the reference re-feeds the whole
    prefix every step, :504-506); results are identical under causal masking;
  * ``noise_mode``: "philox" (default; device RNG keyed by (seed, clip index): invariant to batch
    sharding) or "torch_cpu" (Exp(1) draws taken from torch's global CPU generator in the
    reference's own order, which reproduces the reference CPU path token for token);
  * the codec decodes in fp32 (the reference casts DAC to fp16, :92).
"""
from __future__ import annotations

from typing import Any, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib as L
from .engine import off_null_stream
from .patterns import DelayedPatternProvider
from .utils import instantiate_from_config, sample_from_logits


def _disabled_train(self, mode: bool = True):
    return self


def _plain(o):
    """OmegaConf / Lightning AttributeDict containers -> plain dicts and lists (hyper-parameters out of a checkpoint)."""
    if hasattr(o, "items"):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)) or (hasattr(o, "__iter__") and not isinstance(o, (str, bytes)) and hasattr(o, "__len__")
                                        and not torch.is_tensor(o)):
        return [_plain(v) for v in o]
    return o


class VAURAModel(nn.Module):
    def __init__(self, learning_rate: float = 5e-6, lr_scheduler: dict = None, weight_decay: float = 0.01,
                 betas: tuple = (0.9, 0.95), batch_size: int = 1, use_visual_conditioning: bool = True,
                 feature_extractor_config: dict = None, audio_encoder_config: dict = None, sampler_config: dict = None,
                 visual_bridge_config: dict = None, pattern_provider_config: dict = None,
                 predict_at_val_start: bool = False, return_attention_weights: bool = False,
                 plot_distr_of_pred_indices: bool = False, freeze_feature_extractor: bool = False,
                 files_to_track_during_training: List[str] = None, flatten_vis_feats: bool = False,
                 apply_per_video_frame_mask: bool = False, noise_mode: str = "philox", seed: int = 0):
        super().__init__()
        self.use_visual_conditioning = use_visual_conditioning
        self.visual_feature_extractor = instantiate_from_config(feature_extractor_config) if use_visual_conditioning else None
        if freeze_feature_extractor and self.visual_feature_extractor is not None:
            self.visual_feature_extractor.eval().requires_grad_(False)
        self.using_avclip = self.visual_feature_extractor.__class__.__name__ == "MotionFormer"
        self.flatten_vis_feats = self.using_avclip and flatten_vis_feats
        sampler_config = dict(sampler_config)
        sampler_config["params"] = dict(sampler_config.get("params", {}), use_visual_conditioning=use_visual_conditioning)
        self.sampler = instantiate_from_config(sampler_config)
        self.visual_bridge = instantiate_from_config(visual_bridge_config) if use_visual_conditioning else None
        self.audio_encoder = instantiate_from_config(audio_encoder_config)
        if hasattr(self.sampler, "initialize_embeddings") and self.audio_encoder.__class__.__name__ == "DacModelWrapper":
            self.sampler.initialize_embeddings(self.audio_encoder.model)
        self.audio_encoder.eval().requires_grad_(False)
        self.num_codebooks = self.sampler.num_codebooks
        if pattern_provider_config is not None:
            cfg = dict(pattern_provider_config)
            cfg["params"] = dict(cfg.get("params", {}), n_q=self.num_codebooks)  # :699-714
            self.pattern_provider = instantiate_from_config(cfg)
        else:
            self.pattern_provider = DelayedPatternProvider(n_q=self.num_codebooks)
        if hasattr(self.sampler, "codebook_pattern"):
            self.sampler.codebook_pattern = self.pattern_provider.__class__.__name__
        self.apply_per_video_frame_mask = apply_per_video_frame_mask
        self.return_attention_weights = return_attention_weights
        self.noise_mode = noise_mode
        self.seed = seed
        self.clip_base = 0  # global index of this rank's first clip (vaura_amd.dist)
        self.eval()

    # ------------------------------------------------------------------ checkpoint ingress (scripts/generate.py:208-212)
    # reference plugin classes -> their MI355X counterparts: a hparams.yaml written by the reference's training run names the
    # reference's classes; with `remap_targets` the reference's driver needs no config edit at all
    TARGET_MAP = {
        "models.modules.sampler.llama.Transformer": "vaura_amd.sampler.Transformer",
        "models.modules.dac.model.DacModelWrapper": "vaura_amd.codec.DacModelWrapper",
        "models.modules.feature_extractors.avclip.motionformer.MotionFormer": "vaura_amd.feature_extractor.MotionFormer",
        "models.modules.misc.codebook_patterns.DelayedPatternProvider": "vaura_amd.patterns.DelayedPatternProvider",
    }
    # tensors of a reference checkpoint that nothing on the generation path reads: the extractor's 2-D patch embedding
    # (video_model_builder.py:246-248 builds it, forward_features uses patch_embed_3d)
    UNUSED_CHECKPOINT_KEYS = ("visual_feature_extractor.patch_embed.proj.weight", "visual_feature_extractor.patch_embed.proj.bias")

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict: bool = True,
                             remap_targets: bool = True, **kwargs) -> "VAURAModel":
        """``LightningModule.load_from_checkpoint`` as ``scripts/generate.py:208-212`` calls it —
        ``VAURAModel.load_from_checkpoint(ckpt, hparams_file=hparams.yaml, map_location=device)`` — without Lightning:
        constructor arguments from ``hparams_file`` (the YAML ``save_hyperparameters()`` wrote, vaura_model.py:50; plain
        ``yaml.safe_load``) or, without one, from the checkpoint's own ``hyper_parameters``; then ``state_dict`` (keys
        ``sampler.*``, ``audio_encoder.model.*``, ``visual_feature_extractor.*``) loaded strictly; then ``.to(map_location)``.
        Keyword arguments override hyper-parameters, like Lightning's.  Plugin weights that the checkpoint itself carries
        need no file of their own: a codec without ``ckpt_path`` and an extractor whose ``ckpt_path`` does not exist on this
        machine (scripts/generate.py:31-35 points it at ./segment_avclip/...) are built empty and filled from ``state_dict``
        — if the checkpoint lacks them, the strict load fails loudly."""
        import inspect
        import os
        import warnings
        import yaml
        if callable(map_location) or isinstance(map_location, dict):
            raise L.VauraHipError("load_from_checkpoint: map_location must be a device (or None): tensors are read on the CPU and the model "
                                  "is moved afterwards; a callable / dict remapping is not supported")
        # With an hparams_file only `state_dict` is needed from the checkpoint: tensors only, no pickled objects executed.  Without one
        # the constructor arguments are the checkpoint's own pickled `hyper_parameters` (Lightning's AttributeDict / OmegaConf nodes):
        # that needs the full unpickler — the caller is trusting the file exactly as Lightning's own loader would.
        try:
            blob = torch.load(os.fspath(checkpoint_path), map_location="cpu", weights_only=True)
        except Exception:
            if hparams_file is not None:
                warnings.warn(f"{checkpoint_path}: not loadable with weights_only=True (pickled objects beside the tensors); falling back to "
                              "the full unpickler — only do this with checkpoints you trust", stacklevel=2)
            blob = torch.load(os.fspath(checkpoint_path), map_location="cpu", weights_only=False)
        if "state_dict" not in blob:
            raise L.VauraHipError(f"{checkpoint_path}: not a Lightning checkpoint (no 'state_dict')")
        sd = dict(blob["state_dict"])
        if hparams_file is not None:
            if not str(hparams_file).endswith((".yaml", ".yml")):
                raise L.VauraHipError("hparams_file must be the hparams.yaml of the run (csv is not read)")
            with open(os.fspath(hparams_file)) as f:
                hp = yaml.safe_load(f) or {}
        else:
            hp = blob.get("hyper_parameters") or {}
        hp = _plain(hp)
        hp.update(kwargs)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        dropped = sorted(k for k in hp if k not in accepted)
        if dropped:      # training-side hyper-parameters (optimizer, logging, ...) have no meaning here: say which ones were ignored
            warnings.warn(f"load_from_checkpoint: hyper-parameters without a counterpart on the generation path ignored: {dropped}", stacklevel=2)
        hp = {k: v for k, v in hp.items() if k in accepted}
        for key in ("feature_extractor_config", "audio_encoder_config", "sampler_config", "pattern_provider_config"):
            c = hp.get(key)
            if remap_targets and isinstance(c, dict) and c.get("target") in cls.TARGET_MAP:
                c["target"] = cls.TARGET_MAP[c["target"]]
        ae = hp.get("audio_encoder_config")
        if isinstance(ae, dict) and ae.get("target", "").startswith("vaura_amd.") and any(k.startswith("audio_encoder.model.") for k in sd):
            p = ae.setdefault("params", {})
            if not p.get("ckpt_path") and not p.get("synthetic"):
                p["weights_from_state_dict"] = True
        fe = hp.get("feature_extractor_config")
        if isinstance(fe, dict) and fe.get("target", "").startswith("vaura_amd.") and any(k.startswith("visual_feature_extractor.") for k in sd):
            p = fe.setdefault("params", {})
            if p.get("ckpt_path") and not os.path.exists(p["ckpt_path"]):
                warnings.warn(f"feature extractor ckpt_path {p['ckpt_path']!r} does not exist here: using the extractor tensors the "
                              "checkpoint itself carries (visual_feature_extractor.*)", stacklevel=2)
                p["ckpt_path"] = None
        model = cls(**hp)
        for k in cls.UNUSED_CHECKPOINT_KEYS:
            sd.pop(k, None)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if strict and (missing or unexpected):
            raise L.VauraHipError(f"{checkpoint_path}: state_dict does not fit the plugin modules: {len(missing)} missing "
                                  f"(e.g. {list(missing)[:3]}), {len(unexpected)} unexpected (e.g. {list(unexpected)[:3]})")
        fx = model.visual_feature_extractor
        if fx is not None and hasattr(fx, "_loaded") and not any(k.startswith("visual_feature_extractor.") for k in missing):
            fx._loaded = fx._loaded or any(k.startswith("visual_feature_extractor.") for k in sd)
        model.eval()
        if map_location is not None and not callable(map_location) and not isinstance(map_location, dict):
            model = model.to(map_location)
        return model

    # ------------------------------------------------------------------ small surface
    @property
    def special_token_id(self) -> int:
        return self.sampler.d_codebook

    @property
    def device(self):
        return next(self.sampler.parameters()).device

    def _handle_visual_conditioning(self, frames: torch.Tensor, clip_indices=None, B: int = None):
        if not self.use_visual_conditioning:
            return None
        assert frames is not None
        if self.using_avclip:
            vis_feats, _ = self.visual_feature_extractor(frames)
            if self.flatten_vis_feats:
                Bf, S, Tv, D = vis_feats.shape
                vis_feats = vis_feats.reshape(Bf, S * Tv, D)
        else:
            vis_feats = self.visual_feature_extractor(frames)
        return self.visual_bridge(vis_feats.detach())

    def _exp_noise(self, steps: int, rows: int, vocab: int) -> Optional[torch.Tensor]:
        if self.noise_mode == "philox":
            return None
        if self.noise_mode != "torch_cpu":
            raise ValueError(f"unknown noise_mode {self.noise_mode!r}")
        # one (rows, vocab) exponential draw per step from the global CPU generator: the stream the
        # reference's utils.multinomial -> torch.multinomial consumes on its CPU path
        return torch.stack([torch.empty(rows, vocab).exponential_(1) for _ in range(steps)])

    # ------------------------------------------------------------------ generate
    @torch.no_grad()
    def generate_tokens(self, frames=None, audio: Union[torch.Tensor, None] = None, clip_indices=None,
                        max_new_tokens: int = 512, return_attention_weights: bool = False,
                        return_sampled_indices: bool = True, check: bool = False, use_sampling: bool = True,
                        temp: float = 1.0, top_k: int = 256, top_p: float = 0.0, remove_prompts: bool = False,
                        prompt_is_encoded: bool = False, cfg_scale: float = 1.0) -> torch.Tensor:
        """generate() up to and including revert_pattern_sequence (vaura_model.py:410-572): (B, K, T') int64 tokens on
        the device, no codec decode.  The sliding-window caller (vaura_amd.longform) uses this for every chunk and
        decodes the concatenated tokens once, as the reference's script does (scripts/generate.py:366-369)."""
        assert not self.training, "do not use generation in training mode"
        if return_attention_weights:
            # the reference's own llama sampler returns (logits, None, None) (llama.py:520-539), so its generate() fails on
            # `sa_w[-1, -1, :]` (vaura_model.py:529-531) with this flag: there is no behaviour to reproduce
            raise NotImplementedError("attention-weight dumps are not produced by the fused decode path (nor by the reference's "
                                      "llama sampler, which returns None for them)")
        if audio is not None and not prompt_is_encoded:
            # vaura_model.py:463-469 encodes the prompt here.  (Its unpacking `cat([encoded[0] for encoded in audio])`
            # expects EnCodec's frame list and breaks on DacModelWrapper's (B, 9, T) tensor; the tensor is used as is.)
            audio = self.audio_encoder.encode(audio)
        vis = self._handle_visual_conditioning(frames, clip_indices)
        if vis is None:
            # the reference's llama sampler refuses a missing condition itself: `raise Exception("Not implemented")` under
            # "we should always have audio and video" (llama.py:474-476) — channel-concat conditioning has no unconditional form
            raise NotImplementedError("unconditional generation: the llama sampler always needs video features "
                                      "(the reference raises here too, llama.py:474-476)")
        B = vis.shape[0]
        K = self.num_codebooks
        Tp = 0 if audio is None else int(audio.shape[-1])
        assert Tp < max_new_tokens, "gt audio prompt can not be longer than max_new_tokens"
        use_cfg = cfg_scale > 1.0 and self.sampler.__class__.__name__ == "Transformer"
        eng = self.sampler.engine()
        if self.sampler.audio_tokens_per_video_frame is None:
            raise L.VauraHipError("sampler.audio_tokens_per_video_frame must be set (scripts/generate.py:216 sets 7)")
        S = max_new_tokens + K
        start = Tp + 1  # Pattern.get_first_step_with_timesteps(Tp), delayed pattern
        greedy = not (use_sampling and temp > 0.0)
        noise = None if greedy else self._exp_noise(S - start, B * K, self.sampler.d_codebook)
        # decode loop + its status word in one synchronisation (the reference's own post-conditions, :550-572, synchronise too); an
        # activation beyond the fp16-plane range is re-run on the exact-fp32 engine instead of raising (engine.generate_codes_checked)
        codes = eng.generate_codes_checked(
            vis.float(), max_new_tokens, prompt=audio if Tp else None, use_sampling=use_sampling, temp=temp,
            top_k=top_k, top_p=top_p, cfg_scale=cfg_scale if use_cfg else 1.0, noise=noise, seed=self.seed,
            clip_base=self.clip_base, tokens_per_frame=self.sampler.audio_tokens_per_video_frame)
        bad = (codes < 0) | (codes > self.sampler.d_codebook)
        assert not bool(bad.any()), "generated sequence is incomplete or out of range"
        if check:
            # vaura_model.py:508-515 checks, every step, that the prefix is coherent with the pattern mask and holds no unknown
            # token; the device loop fills the sequence in place, so the same two properties are checked on the finished one
            # (they are monotone: a violation at any step is still there at the end).  :550-558 are these asserts, always on.
            seq = eng.seq[:B].to(torch.int64)
            _, mask = self.pattern_provider.get_pattern(max_new_tokens)._build_indexes(max_new_tokens, seq.device)
            special = torch.full_like(seq, self.special_token_id)
            assert not bool((seq == -1).any()), "unknown tokens left in the generated sequence"
            assert bool((seq == torch.where(mask[None].expand_as(seq), seq, special)).all()), "sequence and pattern mask disagree"
        return codes[..., (Tp if remove_prompts else 0):max_new_tokens]

    @torch.no_grad()
    def generate(self, frames=None, audio: Union[torch.Tensor, None] = None, clip_indices=None, max_new_tokens: int = 512,
                 return_attention_weights: bool = False, return_sampled_indices: bool = False, check: bool = False,
                 use_sampling: bool = True, temp: float = 1.0, top_k: int = 256, top_p: float = 0.0,
                 remove_prompts: bool = False, prompt_is_encoded: bool = False, cfg_scale: float = 1.0) -> dict:
        K = self.num_codebooks
        with off_null_stream(self.sampler.engine().dev) as caller:   # decode loop + codec leave HIP's null stream together
            out_codes = self.generate_tokens(
                frames=frames, audio=audio, clip_indices=clip_indices, max_new_tokens=max_new_tokens,
                return_attention_weights=return_attention_weights, check=check, use_sampling=use_sampling, temp=temp,
                top_k=top_k, top_p=top_p, remove_prompts=remove_prompts, prompt_is_encoded=prompt_is_encoded,
                cfg_scale=cfg_scale)
            generated_audio = self.audio_encoder.decode([(out_codes[..., :K, :], None)])
        if caller is not None:
            out_codes.record_stream(caller)
            generated_audio.record_stream(caller)
        return {"generated_audio": generated_audio, "s_attn_weights": None, "mha_attn_weights": None,
                "sampled_indices": out_codes if return_sampled_indices else None}

    # ------------------------------------------------------------------ one step, reference signature
    @torch.no_grad()
    def _sample_next_token(self, sequence: torch.Tensor, condition: torch.Tensor, use_sampling: bool = False,
                           temp: float = 1.0, top_k: int = 0, top_p: float = 0.0, return_attention_weights: bool = False,
                           cfg_scale: float = 1.0) -> Tuple[torch.Tensor, Any, Any]:
        """sequence (B, K, L) int64, condition (B, Tv, 768) -> (next_token (B, K, 1), None, None).
        Stateless like the reference (the whole prefix is given), so it teacher-forces the prefix through
        the decode kernels; ``generate()`` does not go through here."""
        use_cfg = cfg_scale > 1.0 and self.sampler.__class__.__name__ == "Transformer"
        if use_cfg:
            null = torch.zeros_like(condition) + self.sampler.cls_embeddings.uncond_embedding.to(condition.device)
            condition = torch.cat([condition, null], dim=0)
            sequence = sequence.repeat(2, 1, 1)
        logits, _, _ = self.sampler(tgt=sequence, memory=condition, tgt_is_causal=True)
        last = logits[:, :, -1, :].contiguous()
        tok = sample_from_logits(last, use_sampling=use_sampling, temp=temp, top_k=top_k, top_p=top_p,
                                 cfg_scale=cfg_scale if use_cfg else 1.0)
        return tok, None, None
