"""Time DAC encode / decode of the full-width codec on the GPU box (DESIGN.md §3.5 numbers).
    python tools/time_codec.py [clips]"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import CodecEncoderEngine, CodecEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
cfg = synth.FULL_CODEC
sd = dict(synth.codec_state_dict(cfg, seed=0))
sd.update(synth.codec_encoder_state_dict(cfg, seed=0))
dec, enc = CodecEngine(cfg, sd, dev), CodecEncoderEngine(cfg, sd, dev)
codes = torch.randint(0, 1024, (B, 9, 220), device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    wav = dec.decode(codes)
    enc.encode(wav)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(5):
        wav = dec.decode(codes)
    ev[1].record()
    for _ in range(5):
        back = enc.encode(wav)
    ev[2].record()
torch.cuda.synchronize()
w8 = CodecEngine(cfg, sd, dev, precision="f16pair_w8")
with torch.cuda.stream(s):
    w8.decode(codes)
    e8 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e8[0].record()
    for _ in range(5):
        w8.decode(codes)
    e8[1].record()
torch.cuda.synchronize()
print(f"B={B}: decode with fp8-quantised conv weights (single fp16 weight plane) {e8[0].elapsed_time(e8[1]) / 5:.2f} ms")
h16 = CodecEngine(cfg, sd, dev, precision="f16")
with torch.cuda.stream(s):
    wav16 = h16.decode(codes)
    e16 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e16[0].record()
    for _ in range(5):
        h16.decode(codes)
    e16[1].record()
torch.cuda.synchronize()
print(f"B={B}: decode with plain fp16 operands (\"f16\": one MFMA per product) {e16[0].elapsed_time(e16[1]) / 5:.2f} ms; rms vs the f16-pair decode "
      f"{float(((wav16 - wav).float() ** 2).mean().sqrt()):.3e}")
mx = CodecEngine(cfg, sd, dev, precision="mx8")
with torch.cuda.stream(s):
    wav_mx = mx.decode(codes)
    em = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    em[0].record()
    for _ in range(5):
        mx.decode(codes)
    em[1].record()
torch.cuda.synchronize()
d = (wav_mx - wav).float()
print(f"B={B}: decode on the block-scaled fp8 MFMA (mx8) {em[0].elapsed_time(em[1]) / 5:.2f} ms; rms vs the f16-pair decode "
      f"{float((d ** 2).mean().sqrt()):.3e} (signal rms {float((wav ** 2).mean().sqrt()):.3e})")
print(f"B={B} clips of 2.56 s: decode {ev[0].elapsed_time(ev[1]) / 5:.2f} ms, encode {ev[1].elapsed_time(ev[2]) / 5:.2f} ms")
