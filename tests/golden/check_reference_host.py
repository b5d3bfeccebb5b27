"""Drop-in check on the REFERENCE host (build container only; CPU, no GPU call): the reference's own ``VAURAModel``
(/root/reference/models/vaura_model.py:28-120) is constructed with all plugin slots pointing at ``vaura_amd`` — the params
read from the reference's own YAML files with only the ``target:`` strings swapped — and driven through
everything its constructor and generate() preamble touch on the plugins:

  instantiate_from_config of the three slots                      vaura_model.py:64-85
  sampler.initialize_embeddings(audio_encoder.model)              :86-88  (class-name gate "DacModelWrapper")
  audio_encoder.eval() / requires_grad_(False) / .train / .model.half()   :89-92
  sampler.num_codebooks, sampler.codebook_pattern = ...           :93-101
  sampler.d_codebook (special_token_id), sampler.cls_embeddings.uncond_embedding     :127, 790-793
  sampler.config.block_size, sampler.audio_tokens_per_video_frame (scripts/generate.py:214-224)
  load_state_dict of a reference-keyed checkpoint through the HOST (sampler.*, audio_encoder.model.*, visual_feature_extractor.*)
  _handle_visual_conditioning with the MotionFormer plugin (pre-extracted features)  :194-214
  the MIXED case: the reference's own sampler initialising its embeddings from the vaura_amd codec (llama.py:387-412)

Fails with the attribute / key the host touches and a plugin lacks.     python tests/golden/check_reference_host.py
"""
import os
import sys

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
os.environ["VAURA_SYNTHETIC_CODEC"] = "1"      # no DAC checkpoint in this container: the codec plugin would otherwise refuse

import ref_harness as rh  # noqa: E402
from vaura_amd import synth  # noqa: E402


def numbers(o):
    """yaml.safe_load leaves `1e-5` a string; OmegaConf (what the reference uses) reads a float."""
    if isinstance(o, dict):
        return {k: numbers(v) for k, v in o.items()}
    if isinstance(o, str):
        try:
            return float(o)
        except ValueError:
            return o
    return o


def module_cfg(rel, target):
    cfg = numbers(yaml.safe_load(open(os.path.join(rh.REFERENCE_ROOT, "configs", "modules", rel))))
    cfg["target"] = target
    cfg.setdefault("params", {})
    return cfg


def main():
    rh.install()
    from models.vaura_model import VAURAModel  # the reference host
    fe = module_cfg("feature_extractors/avclip_vggsound.yaml", "vaura_amd.feature_extractor.MotionFormer")
    fe["params"]["ckpt_path"] = None           # /path/to/vggsound/epoch_best.pt in the YAML
    host = VAURAModel(
        feature_extractor_config=fe,
        audio_encoder_config=module_cfg("audio_codecs/dac_8kbps_wrapper.yaml", "vaura_amd.codec.DacModelWrapper"),
        sampler_config=module_cfg("samplers/llama_9cbs.yaml", "vaura_amd.sampler.Transformer"),
        visual_bridge_config=module_cfg("bridges/dummy_bridge.yaml", "torch.nn.Identity"),
        pattern_provider_config=module_cfg("codebook_patterns/delayed_9cbs.yaml", "vaura_amd.patterns.DelayedPatternProvider"),
        flatten_vis_feats=True, freeze_feature_extractor=True)
    host.eval()
    s = host.sampler
    assert type(s).__name__ == "Transformer" and type(host.audio_encoder).__name__ == "DacModelWrapper" and host.using_avclip
    assert host.num_codebooks == 9 and host.special_token_id == 1024 and s.codebook_pattern == "DelayedPatternProvider"
    assert s.config.block_size == 256 and s.cls_embeddings.uncond_embedding.shape == (32, 768)
    assert next(host.audio_encoder.model.parameters()).dtype == torch.float16          # the host's .half()
    # initialize_embeddings copied the codec's codebooks / projections into the sampler (llama.py:387-412)
    q0 = host.audio_encoder.model.quantizer.quantizers[0]
    assert torch.equal(s.tok_embeddings[0].emb.weight[:1024].half(), q0.codebook.weight.detach().cpu())   # codec halved AFTER the copy
    s.audio_tokens_per_video_frame = 7          # scripts/generate.py:216
    # a reference-keyed checkpoint through the HOST's load_state_dict (keys as SURVEY.md §5 lists them)
    cfg = synth.FULL_SAMPLER
    ckpt = {"sampler." + k: v for k, v in synth.sampler_state_dict(synth.tiny_sampler(24), seed=0, round_bf16=False).items()}
    ckpt.update({"audio_encoder.model." + k: v for k, v in host.audio_encoder.model.state_dict().items()})
    ckpt.update({"visual_feature_extractor." + k: v for k, v in synth.avclip_state_dict(seed=0).items()})
    res = host.load_state_dict(ckpt, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(s.state_dict()["layers.7.feed_forward.w2.weight"], ckpt["sampler.layers.7.feed_forward.w2.weight"])
    from vaura_amd.engine import resolve_weight_dtype
    assert s.weight_dtype == "auto" and resolve_weight_dtype(s.state_dict(), "auto") == "f32"     # un-rounded weights -> f32 storage
    # conditioning path of generate() with pre-extracted features (vaura_model.py:194-214)
    feats = synth.video_features(2).reshape(2, 4, 8, 768)
    vis = host._handle_visual_conditioning(feats, None, 2)
    assert vis.shape == (2, 32, 768)
    # the pattern plugin the host calls at the top of generate() (:480-496)
    pat = host.pattern_provider.get_pattern(220)
    assert pat.get_first_step_with_timesteps(0) == 1
    # generate() itself needs the HIP device: the plugin must say so loudly, not fall back
    try:
        host.generate(frames=feats, audio=None, max_new_tokens=4, use_sampling=False, prompt_is_encoded=True)
        raise SystemExit("generate() on CPU did not fail")
    except Exception as e:  # noqa: BLE001
        assert "HIP device" in str(e), repr(e)
    # MIXED case: the reference's own sampler next to the vaura_amd codec plugin
    mixed = VAURAModel(
        feature_extractor_config={"target": "vaura_amd.feature_extractor.MotionFormer", "params": {"extract_features": True}},
        audio_encoder_config=module_cfg("audio_codecs/dac_8kbps_wrapper.yaml", "vaura_amd.codec.DacModelWrapper"),
        sampler_config=numbers(yaml.safe_load(open(os.path.join(rh.REFERENCE_ROOT, "configs/modules/samplers/llama_9cbs.yaml")))),
        visual_bridge_config={"target": "torch.nn.Identity"},
        pattern_provider_config=module_cfg("codebook_patterns/delayed_9cbs.yaml", "models.modules.misc.codebook_patterns.DelayedPatternProvider"),
        flatten_vis_feats=True, freeze_feature_extractor=True)
    w = mixed.sampler.tok_embeddings[3].out_proj
    assert w.weight_v.shape == (1024, 8, 1)
    print("reference host accepts the vaura_amd plugins: constructor, initialize_embeddings, .half(), load_state_dict, "
          "conditioning, pattern; generate() refuses to run without a HIP device; mixed reference-sampler / vaura_amd-codec ok")


if __name__ == "__main__":
    main()
