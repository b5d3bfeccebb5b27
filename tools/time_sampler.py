"""Time of one sampler launch per mode (op-level vaura_sample, back to back on one stream): python tools/time_sampler.py"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import _lib as L
import ctypes as C
lib = L.lib()
dev = "cuda:0"
B, K, V = 8, 9, 1024
logits = torch.randn(2 * B, K * V, device=dev) * 3
tokens = torch.zeros(B * K, dtype=torch.int32, device=dev)
def run(use_sampling, top_k, top_p, cfg, tie_eps=0.0):
    sp = L.Sampling()
    sp.use_sampling, sp.top_k, sp.temp, sp.top_p, sp.cfg_scale, sp.seed, sp.clip_base, sp.input_is_probs = use_sampling, top_k, 1.0, top_p, cfg, 1, 0, 0
    sp.tie_eps = tie_eps          # near-tie screen (round 6): 0 = off
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(50):
            lib.vaura_sample(L.ptr(logits), B, K, V, C.byref(sp), None, 0, L.ptr(tokens), L.current_stream(torch.device(dev)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(2000):
            lib.vaura_sample(L.ptr(logits), B, K, V, C.byref(sp), None, i, L.ptr(tokens), L.current_stream(torch.device(dev)))
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 2000 * 1000
for name, a in (("greedy cfg6", (0, 0, 0.0, 6.0)), ("plain sampling cfg6", (1, 0, 0.0, 6.0)), ("top-k 250 cfg6", (1, 250, 0.0, 6.0)), ("top-p 0.9 cfg6", (1, 0, 0.9, 6.0)), ("top-k 250 cfg1", (1, 250, 0.0, 1.0))):
    print(f"{name}: {run(*a):.2f} us per launch (back to back, includes the dispatch gap); with the near-tie screen (tie_eps 1.5e-6): {run(*a, 1.5e-6):.2f}")
