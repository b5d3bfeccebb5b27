"""Calibration of the near-tie detector (csrc/step.hip, vaura_sampling.tie_eps) on the reference's own goldens: for every bound, how
many USED decisions of each golden run are flagged, and at which step first.  The detector never changes a token; what is wanted is
the bound that flags the two literal near-ties the goldens hold (configs[3] step 578: reference margin 5.5e-6; the later chunk's greedy
run, clip 1, step 175: 3.8e-6) and nothing in the headline golden.  GPU box:  python tools/near_tie_sweep.py [eps ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import DecoderEngine  # noqa: E402

DEV = "cuda:0"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
EPS = [float(x) for x in sys.argv[1:]] or [0.5e-6, 1e-6, 1.2e-6, 1.5e-6, 2e-6, 4e-6, 1e-5]


def load(n):
    return np.load(os.path.join(G, n))


def run(eng, what, fn):
    for eps in EPS:
        eng.near_tie_eps = eps
        eng._free_graph()
        fn()
        eng.check_status()
        print(f"{what:72s} eps {eps:8.1e}: flagged {eng.last_near_ties[0]:4d}  first step {eng.last_near_ties[1]}", flush=True)


sd_raw = synth.sampler_state_dict(synth.FULL_SAMPLER, seed=0, round_bf16=False)
eng = DecoderEngine(synth.FULL_SAMPLER, sd_raw, DEV)
g = load("full_topk250_cfg6_raw_B2_T220.npz")
feats = synth.video_features(8, seed=int(g["feat_seed"])).to(DEV)
nz = torch.cat([synth.exp_noise(228, 18, 1024, int(g["noise_seed"])), synth.exp_noise(228, 54, 1024, 4321)], dim=1)
run(eng, "headline golden arithmetic (h2, B=8, cfg 6, top-k 250 sampled): 16 416 decisions",
    lambda: eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, noise=nz))
run(eng, "same, B=2 (the golden's own two clips): 4 104 decisions",
    lambda: eng.generate_codes(feats[:2], 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, noise=nz[:, :18].contiguous()))
gg = load("full_greedy_cfg6_raw_B2_T220.npz")
fg = synth.video_features(2, seed=int(gg["feat_seed"])).to(DEV)
run(eng, f"greedy cfg 6 golden (h2, B=2; reference min margin {float(gg['margins'].min()):.1e})", lambda: eng.generate_codes(fg, 220, cfg_scale=6.0))
gc = load("full_chunk_greedy_cfg6_raw_B2_Tp166_T221.npz")
fc = synth.video_features(2, seed=int(gc["feat_seed"])).to(DEV)
prompt = torch.from_numpy(gc["prompt"].astype(np.int64)).to(DEV)
run(eng, "later chunk, greedy cfg 6 (h2, B=2; holds the literal tie at step 175: 3.8e-6)", lambda: eng.generate_codes(fc, 221, prompt=prompt, cfg_scale=6.0))
# how often a plain production call is flagged at the product's bound: 24 different 8-clip cfg-6 top-k-250 calls (Philox seeds), 15 840 decisions each
from vaura_amd.engine import NEAR_TIE_EPS  # noqa: E402
eng.near_tie_eps = NEAR_TIE_EPS
eng._free_graph()
counts = []
for seed in range(24):
    eng.generate_codes(synth.video_features(8, seed=100 + seed).to(DEV), 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, seed=seed)
    eng.check_status()
    counts.append(eng.last_near_ties[0])
print(f"24 production-shaped calls (B=8, cfg 6, top-k 250, 15 840 decisions each) at eps {NEAR_TIE_EPS:.1e}: flagged per call {counts}; "
      f"mean {sum(counts) / len(counts):.2f}, calls with none {sum(c == 0 for c in counts)}/24", flush=True)
del eng
torch.cuda.empty_cache()
g4 = load("full_c4_greedy_B1_T880.npz")
cfg4 = synth.SamplerCfg(block_size_audio=int(g4["block_size_audio"]))
sd4 = synth.sampler_state_dict(cfg4, seed=int(g4["weight_seed"]), round_bf16=True)
for wd in ("h1", "h2"):
    e4 = DecoderEngine(cfg4, sd4, DEV, wdtype=wd)
    f4 = synth.video_features(1, tokens=128, seed=int(g4["feat_seed"])).to(DEV)
    run(e4, f"configs[3] golden ({wd}, B=1, greedy cfg 1, T=880; literal tie at step 578: 5.5e-6)", lambda: e4.generate_codes(f4, 880))
    del e4
    torch.cuda.empty_cache()
