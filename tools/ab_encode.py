"""Encode of one waveform with 256-row and with 128-row conv workgroups (debug flag bit 20): the codes must be identical."""
import os, torch, sys
sys.path.insert(0, ".")
from vaura_amd import synth, _lib as L
from vaura_amd.engine import CodecEncoderEngine, CodecEngine
cfg = synth.FULL_CODEC
sd = dict(synth.codec_state_dict(cfg, seed=0)); sd.update(synth.codec_encoder_state_dict(cfg, seed=0))
dec, enc = CodecEngine(cfg, sd, "cuda:0"), CodecEncoderEngine(cfg, sd, "cuda:0")
codes = torch.randint(0, 1024, (8, 9, 220), device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(1))
wav = dec.decode(codes).clone()
out = {}
for f in (0, 1048576):
    L.lib().vaura_set_debug_flags(f)
    out[f] = enc.encode(wav).clone()
torch.cuda.synchronize()
print("encode codes identical across workgroup heights:", bool(torch.equal(out[0], out[1048576])), int(out[0].sum()))
