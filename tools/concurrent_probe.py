"""Probe: decode-loop tokens under a concurrently loaded chip (second stream running codec decodes) against the quiet run, per
debug-flag set (vaura_set_debug_flags).   python tools/concurrent_probe.py 0,2,4 [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import _lib as L, synth  # noqa: E402
from vaura_amd.engine import CodecEngine, DecoderEngine  # noqa: E402

flagsets = [int(x, 0) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
DEV = "cuda:0"
cfg = synth.tiny_sampler(3)
sd = synth.sampler_state_dict(cfg, seed=111, round_bf16=False)
feats = synth.video_features(8, seed=112).to(DEV)
kw = dict(cfg_scale=6.0, use_sampling=True, top_k=250, seed=5)
ccfg = synth.FULL_CODEC
codec = CodecEngine(ccfg, synth.codec_state_dict(ccfg, seed=0), DEV)
codes = torch.randint(0, 1024, (8, 9, 220), device=DEV)
codec.decode(codes)
side, main = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
for f in flagsets:
    L.lib().vaura_set_debug_flags(f)
    eng = DecoderEngine(cfg, sd, DEV)
    with torch.cuda.stream(main):
        ref = eng.generate_codes(feats, 60, **kw).clone()
    torch.cuda.synchronize()
    bad, first = 0, None
    for r in range(rounds):
        with torch.cuda.stream(side):
            for _ in range(4):
                codec.decode(codes)
        with torch.cuda.stream(main):
            out = eng.generate_codes(feats, 60, **kw).clone()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            if first is None:
                d = (out != ref)
                steps = torch.arange(60, device=DEV)[None, None, :] + 1 + torch.arange(9, device=DEV)[None, :, None]
                first = int(steps.expand_as(d)[d].min())
    st = int(eng.state[4].item())
    print(f"flags {f:#x}: {bad}/{rounds} loaded runs differ from the quiet run (first differing step {first}); status word {st}")
    del eng
L.lib().vaura_set_debug_flags(0)
