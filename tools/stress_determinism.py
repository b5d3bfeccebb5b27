"""Race screen: the same inputs through every HIP path many times must give bit-identical outputs (no atomics with an order
that matters anywhere in the product path).  python tools/stress_determinism.py [repeats]      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import AvclipEngine, CodecEncoderEngine, CodecEngine, DecoderEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
bad = 0


def screen(name, fn, n=N):
    global bad
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else (ref,))]
    diff = 0
    for _ in range(n - 1):
        out = fn()
        out = out if isinstance(out, (tuple, list)) else (out,)
        diff += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, out))
    print(f"{name:58s} {n} runs, {diff} differing outputs")
    bad += diff


cfg = synth.FULL_SAMPLER
feats = synth.video_features(8, seed=0).to(dev)
for wd, sd in (("h1", synth.sampler_state_dict(cfg, seed=0, round_bf16=True)), ("h2", synth.sampler_state_dict(cfg, seed=0, round_bf16=False))):
    eng = DecoderEngine(cfg, sd, dev, wdtype=wd)
    kw = dict(use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, seed=7)
    screen(f"decode loop, {wd} storage, 8 clips, top-k 250, cfg 6", lambda: eng.generate_codes(feats, 220, **kw), max(4, N // 4))
    prompt = torch.randint(0, 1024, (8, 9, 166), generator=torch.Generator().manual_seed(1)).to(dev)
    screen(f"prompt prefill + loop, {wd} storage", lambda: eng.generate_codes(feats, 221, prompt=prompt, **kw), max(4, N // 4))
    feats16 = synth.video_features(16, seed=0).to(dev)      # 32 decoder rows: the two-row-block GEMVs and one-launch MLP (round 5)
    screen(f"decode loop, {wd} storage, 16 clips (32 rows), top-k 250, cfg 6", lambda: eng.generate_codes(feats16, 220, **kw), max(4, N // 4))
    eng.check_status()
    del eng
    torch.cuda.empty_cache()
# round 6: configs[4]'s engine (fp8 weights against the hi plane, fp16 K/V cache), 32 rows, with and without a prompt
eng = DecoderEngine(cfg, synth.sampler_state_dict(cfg, seed=0, round_bf16=False), dev, wdtype="fp8h", kv_dtype="f16")
feats16 = synth.video_features(16, seed=0).to(dev)
screen("decode loop, fp8h + fp16 K/V, 16 clips (32 rows), top-k 250, cfg 6", lambda: eng.generate_codes(feats16, 220, use_sampling=True, top_k=250, cfg_scale=6.0, seed=7),
       max(4, N // 4))
prompt16 = torch.randint(0, 1024, (16, 9, 166), generator=torch.Generator().manual_seed(2)).to(dev)
screen("prompt prefill + loop, fp8h + fp16 K/V, 32 rows", lambda: eng.generate_codes(feats16, 221, prompt=prompt16, use_sampling=True, top_k=250, cfg_scale=6.0, seed=7),
       max(4, N // 4))
eng.check_status()
del eng
torch.cuda.empty_cache()
ccfg = synth.FULL_CODEC
csd = dict(synth.codec_state_dict(ccfg, seed=0))
csd.update(synth.codec_encoder_state_dict(ccfg, seed=0))
codes = torch.randint(0, 1024, (8, 9, 220), device=dev)
for prec in ("f16pair", "f32", "f16pair_w8", "mx8"):
    dec = CodecEngine(ccfg, csd, dev, precision=prec)
    screen(f"codec decode, precision {prec}", lambda: dec.decode(codes))
    if prec == "f16pair":
        wav = dec.decode(codes).clone()
    del dec
enc = CodecEncoderEngine(ccfg, csd, dev)
screen("codec encode", lambda: enc.encode(wav))
av = AvclipEngine(synth.FULL_AVCLIP, synth.avclip_state_dict(seed=0), dev)
frames = torch.randn(4, 4, 3, 16, 224, 224, device=dev)
screen("Segment-AVCLIP features, 4 clips x 4 segments", lambda: av.forward(frames))
print("RACE SCREEN:", "clean" if bad == 0 else f"{bad} DIFFERENCES")
sys.exit(1 if bad else 0)
