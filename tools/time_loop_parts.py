"""Where does a configs[1] batch spend its time outside the kernels?  HIP events on the engine's stream + host clocks:
(a) generate_codes (condition MLP + pattern + loop + revert) and (b) vaura_generate_loop alone (228 graph replays), each
  - back to back (5 batches enqueued without waiting: once the host is a full AQL ring ahead of the GPU, hipGraphLaunch BLOCKS
    until the queue drains, so this "host time" converges to the GPU time minus the ring's depth - it is back-pressure,
    not enqueue cost), and
  - one batch enqueued on an idle, synchronised stream (the real cost of enqueueing 228 graph replays)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import CodecEngine, DecoderEngine  # noqa: E402

dev = torch.device("cuda:0")
_v = ctypes.c_int(0)
torch.cuda.init()
ctypes.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(ctypes.byref(_v))
print("HIP runtime in this process:", _v.value)
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
        print(f"{name:16s} 5 batches back to back: GPU {e0.elapsed_time(e1) / 5:8.3f} ms per batch, host {1e3 * th / 5:8.3f} ms per batch (back-pressure included)")
        hs = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            hs.append(1e3 * (time.perf_counter() - t0))
            torch.cuda.synchronize()
        print(f"{name:16s} one batch on an idle stream: host enqueue {sorted(hs)[2]:8.3f} ms (median of 5)")
    eng.check_status()        # a broken hand-off / non-finite logits would make these times meaningless: fail instead
