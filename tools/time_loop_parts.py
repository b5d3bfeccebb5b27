"""Where does a configs[1] batch spend its time outside the kernels?  Times, with HIP events on the engine's stream:
(a) vaura_generate_loop alone (228 graph replays), (b) generate_codes (condition MLP + pattern + loop + revert), (c) codec."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import CodecEngine, DecoderEngine  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.FULL_SAMPLER
eng = DecoderEngine(cfg, synth.sampler_state_dict(cfg, seed=0, round_bf16=True), dev, wdtype="bf16")
feats = synth.video_features(8, seed=0).to(dev)
kw = dict(use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, seed=1)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(2):
        eng.generate_codes(feats, 220, **kw)
    torch.cuda.synchronize()
    sp = eng._sampling(True, 1.0, 250, 0.0, 6.0, 1, 0)
    res = {}
    for name, fn in (("generate_codes", lambda: eng.generate_codes(feats, 220, **kw)),
                     ("loop only", lambda: (eng.start_sequence(None), eng.run(0, 228, sp))[0])):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 5, 1e3 * th / 5)
    for k, (g, h) in res.items():
        print(f"{k:16s} GPU {g:8.3f} ms per batch, host enqueue {h:8.3f} ms")
