"""Time one later chunk of the sliding-window caller (scripts/generate.py:327-365): Tp = 166 prompt tokens,
T = 221, i.e. 166 teacher-forced positions + 63 generated ones, B = 8, cfg 6.0 — batched prefill vs
per-position teacher forcing."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth
from vaura_amd.engine import DecoderEngine

dev = "cuda:0"
cfg = synth.FULL_SAMPLER
sd = synth.sampler_state_dict(cfg, seed=0)
feats = synth.video_features(8, seed=0).to(dev)
prompt = torch.randint(0, 1024, (8, 9, 166), generator=torch.Generator().manual_seed(1)).to(dev)
res = {}
PASSES = tuple(int(x) for x in os.environ.get("VAURA_PREFILL_PASSES", "32,64,192,0").split(","))
from vaura_amd import _lib as L
FLAGS = int(os.environ.get("VAURA_DEBUG_FLAGS", "0"))    # A/B: 16 = per-position prefill attention, 32 = 64-row prefill GEMM only
L.lib().vaura_set_debug_flags(FLAGS)
print("debug flags:", FLAGS)
WD = os.environ.get("VAURA_WEIGHTS", "h1")     # h1: one fp16 plane (the checkpoint here is made bf16-representable) | h2: two planes
if WD == "h1":
    sd = {k: (synth.to_bf16_exact(v) if v.dtype == torch.float32 and v.dim() == 2 else v) for k, v in sd.items()}
print("weights:", WD)
for pp in PASSES:
    DecoderEngine.PREFILL_POSITIONS = pp if pp else 1
    eng = DecoderEngine(cfg, sd, dev, wdtype=WD)
    kw = dict(prompt=prompt, use_sampling=True, top_k=250, cfg_scale=6.0, seed=3)
    out = eng.generate_codes(feats, 221, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = eng.generate_codes(feats, 221, **kw)
    torch.cuda.synchronize()
    res[pp] = (time.perf_counter() - t0) / 3
    eng.check_status()        # a broken hand-off / non-finite logits would make this time meaningless: fail instead
    print(f"prefill_positions={pp}: {res[pp] * 1e3:.1f} ms per chunk (166 prompt + 63 generated positions)")
    if pp == PASSES[0]:
        ref = out.clone()
    elif pp:
        print("  tokens identical to the 32-position passes:", bool(torch.equal(ref, out)))
    else:
        print("tokens identical to per-position teacher forcing:", bool(torch.equal(ref, out)))
    del eng
    torch.cuda.empty_cache()
