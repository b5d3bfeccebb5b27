"""Row f1 measurement: a 10.24 s clip generated the way the reference's script does it for the released 2.56 s model
(scripts/generate.py:327-369): 2.56 s window, 0.64 s stride, prompt carry-over, one codec decode at the end.
Engines driven directly (as bench.py does); synthetic weights / features; B clips, cfg 6.0, top-k 250.
    python tools/bench_longform.py [clips] [duration_s]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import CodecEngine, DecoderEngine  # noqa: E402
from vaura_amd.longform import COMPRESSION_MODEL_FRAME_RATE, chunk_schedule  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
duration = float(sys.argv[2]) if len(sys.argv) > 2 else 10.24
dev = "cuda:0"
cfg, ccfg = synth.FULL_SAMPLER, synth.FULL_CODEC
eng = DecoderEngine(cfg, synth.sampler_state_dict(cfg, seed=0), dev, wdtype="bf16")
codec = CodecEngine(ccfg, synth.codec_state_dict(ccfg, seed=0), dev)
n_seg = 16                                                     # 16 segments of 8 feature tokens = 10.24 s of video
feats = synth.video_features(B, n_seg * 8, seed=0).reshape(B, n_seg, 8, 768).to(dev)
sched = chunk_schedule(duration, 2.56, 0.64, 25)
stride_tokens = int(COMPRESSION_MODEL_FRAME_RATE * 0.64)
kw = dict(use_sampling=True, top_k=250, cfg_scale=6.0, seed=3)


def run():
    all_tokens, prompt = [], None
    for ch in sched:
        lo, hi = ch["positions"]
        pos = torch.arange(lo, hi, device=dev)
        sel = feats[:, pos % n_seg].reshape(B, -1, 768)
        tok = eng.generate_codes(sel, ch["max_gen_len"], prompt=prompt, **kw)
        all_tokens.append(tok if prompt is None else tok[:, :, prompt.shape[-1]:])
        prompt = tok[:, :, stride_tokens:]
    codes = torch.cat(all_tokens, dim=-1)
    return codes, codec.decode(codes)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        codes, wav = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
eng.check_status()            # a broken hand-off / non-finite logits would make this time meaningless: fail instead
n_tok = codes.shape[-1]
print(json.dumps({"workload": f"{duration} s clips via the sliding-window caller (2.56 s window, 0.64 s stride), batch {B}, cfg 6.0, top-k 250",
                  "chunks": len(sched), "tokens_per_clip_and_codebook": n_tok, "seconds_per_batch": round(dt, 4),
                  "codec_tokens_per_s": round(B * 9 * n_tok / dt, 1), "sec_audio_per_sec": round(B * n_tok * 512 / 44100 / dt, 2)}))
