"""A/B of codec kernel variants on one box: decode time of the default codec per debug-flag set (vaura_set_debug_flags), alternating,
and the waveform difference between the sets (same arithmetic -> expected 0).
    python tools/ab_codec.py 0,1048576 [clips] [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import _lib as L, synth  # noqa: E402
from vaura_amd.engine import CodecEngine  # noqa: E402

flagsets = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prec = sys.argv[3] if len(sys.argv) > 3 else None
dev = "cuda:0"
cfg = synth.FULL_CODEC
sd = dict(synth.codec_state_dict(cfg, seed=0))
dec = CodecEngine(cfg, sd, dev, **({"precision": prec} if prec else {}))
codes = torch.randint(0, 1024, (B, 9, 220), device=dev)
s = torch.cuda.Stream()
wavs = {}
for rep in range(3):
    for f in flagsets:
        L.lib().vaura_set_debug_flags(f)
        with torch.cuda.stream(s):
            wavs[f] = dec.decode(codes).clone()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record()
            for _ in range(5):
                dec.decode(codes)
            e[1].record()
        torch.cuda.synchronize()
        print(f"flags {f}: decode {e[0].elapsed_time(e[1]) / 5:.3f} ms")
L.lib().vaura_set_debug_flags(0)
for f in flagsets[1:]:
    d = (wavs[f] - wavs[flagsets[0]]).float()
    print(f"flags {f} vs {flagsets[0]}: max |diff| {float(d.abs().max()):.3e}, identical {bool(torch.equal(wavs[f], wavs[flagsets[0]]))}")
