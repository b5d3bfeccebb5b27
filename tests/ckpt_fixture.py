"""A Lightning-shaped V-AURA checkpoint built from seeded synthetic weights (tests only): ``state_dict`` with the reference's
key prefixes (``sampler.*``, ``audio_encoder.model.*``, ``visual_feature_extractor.*`` incl. the extractor's unused 2-D patch
embedding) + an ``hparams.yaml`` as ``save_hyperparameters()`` writes it (models/vaura_model.py:50), naming the REFERENCE's
plugin classes (configs/modules/*) — what ``scripts/generate.py:198-212`` hands to ``VAURAModel.load_from_checkpoint``."""
import os

import torch
import yaml

from vaura_amd import synth


def reference_hparams(sampler_cfg: synth.SamplerCfg) -> dict:
    return {
        "learning_rate": 1e-3, "weight_decay": 0, "betas": [0.9, 0.95], "batch_size": 2,
        "lr_scheduler": {"target": "models.modules.misc.lr_schedulers.InverseSquareRootLRScheduler",
                         "params": {"warmup_steps": 3000, "warmup_init_lr": 1e-6}},
        "use_visual_conditioning": True, "freeze_feature_extractor": True,
        "feature_extractor_config": {"target": "models.modules.feature_extractors.avclip.motionformer.MotionFormer",
                                     "params": {"ckpt_path": "./segment_avclip/vggsound/best.pt", "extract_features": True,
                                                "factorize_space_time": True, "agg_space_module": "TransformerEncoderLayer",
                                                "agg_time_module": "torch.nn.Identity", "add_global_repr": False}},
        "audio_encoder_config": {"target": "models.modules.dac.model.DacModelWrapper", "params": {"model_sr": 44100}},
        "sampler_config": {"target": "models.modules.sampler.llama.Transformer", "params": sampler_cfg.yaml_params()},
        "visual_bridge_config": {"target": "torch.nn.Identity"},
        "pattern_provider_config": {"target": "models.modules.misc.codebook_patterns.DelayedPatternProvider", "params": {"n_q": 9}},
        "predict_at_val_start": False, "return_attention_weights": False, "plot_distr_of_pred_indices": False,
        "files_to_track_during_training": ["a", "b"], "flatten_vis_feats": True, "apply_per_video_frame_mask": False,
    }


def state_dicts(sampler_cfg: synth.SamplerCfg, seed: int = 5):
    sd_s = synth.sampler_state_dict(sampler_cfg, seed=seed, round_bf16=False)
    sd_c = dict(synth.codec_state_dict(synth.FULL_CODEC, seed=seed + 1))
    sd_c.update(synth.codec_encoder_state_dict(synth.FULL_CODEC, seed=seed + 1))
    sd_v = synth.avclip_state_dict(synth.FULL_AVCLIP, seed=seed + 2)
    return sd_s, sd_c, sd_v


def write_checkpoint(dirpath, sampler_cfg: synth.SamplerCfg, seed: int = 5):
    """-> (ckpt path, hparams path, (sampler sd, codec sd, extractor sd))"""
    sd_s, sd_c, sd_v = state_dicts(sampler_cfg, seed)
    state = {f"sampler.{k}": v for k, v in sd_s.items()}
    state.update({f"audio_encoder.model.{k}": v for k, v in sd_c.items()})
    state.update({f"visual_feature_extractor.{k}": v for k, v in sd_v.items()})
    state["visual_feature_extractor.patch_embed.proj.weight"] = torch.zeros(768, 3, 16, 16)     # built, never used (video_model_builder.py:246)
    state["visual_feature_extractor.patch_embed.proj.bias"] = torch.zeros(768)
    hp = reference_hparams(sampler_cfg)
    ckpt = os.path.join(dirpath, "epoch=9-step=1000.ckpt")
    torch.save({"epoch": 9, "global_step": 1000, "pytorch-lightning_version": "2.1.0", "state_dict": state,
                "hyper_parameters": hp}, ckpt)
    hpath = os.path.join(dirpath, "hparams.yaml")
    with open(hpath, "w") as f:
        yaml.safe_dump(hp, f)
    return ckpt, hpath, (sd_s, sd_c, sd_v)
