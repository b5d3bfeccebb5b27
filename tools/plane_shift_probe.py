import sys, torch
sys.path.insert(0, '.')
from vaura_amd import synth
from vaura_amd.engine import DecoderEngine
from oracle import generate_oracle as go
from oracle.decoder_oracle import DecoderOracle
DEV='cuda:0'
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
ref = go.generate(dec, feats.cpu(), 12, mode="cached", cfg_scale=6.0)
for S in (4, 8, 10, 12, 16):
    e = DecoderEngine(cfg, big, DEV, wdtype="h2", plane_shift=S)
    got = e.generate_codes_checked(feats, 12, cfg_scale=6.0).cpu()
    print(S, "fallbacks", e.range_fallbacks, "equal", torch.equal(got, ref), "ndiff", int((got != ref).sum()))
