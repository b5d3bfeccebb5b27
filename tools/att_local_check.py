"""scratch: the ATT-local experiment build (libvaura_hip_attl.so): tokens with flags2 = 4 (attention as the engine's fourth phase, XCD-local
hand-off) must equal flags2 = 0 (separate attention launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvaura_hip_attl.so")
from vaura_amd import synth
from vaura_amd.engine import DecoderEngine
dev = "cuda:0"
cfg = synth.FULL_SAMPLER
sd = synth.sampler_state_dict(cfg, seed=0, round_bf16=False)
feats = synth.video_features(8, seed=0).to(dev)
out = {}
for f2 in (0, 4, 0, 4):
    L.lib().vaura_set_debug_flags2(f2)
    eng = DecoderEngine(cfg, sd, dev, wdtype="h2")
    t = eng.generate_codes(feats, 220, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, seed=7)
    eng.check_status()
    out.setdefault(f2, []).append(t.clone())
    del eng
    torch.cuda.empty_cache()
print("ATT-local identical to separate attention:", all(torch.equal(out[0][0], x) for x in out[0] + out[4]))
