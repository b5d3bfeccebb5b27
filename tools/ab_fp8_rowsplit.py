"""A/B of the fp8 row-split pair instances (wo / w2 on gemv3h_kernel<.., WT = 1>, round 5) against round 4's one-workgroup-per-tile
kernels (second flag word, bit 3), whole 228-step loops, alternating in one process.  python tools/ab_fp8_rowsplit.py [clips]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import _lib as L, synth
from vaura_amd.engine import DecoderEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
cfg = synth.FULL_SAMPLER
eng = DecoderEngine(cfg, synth.sampler_state_dict(cfg, seed=0, round_bf16=False), dev, wdtype="fp8")
feats = synth.video_features(B, seed=0).to(dev)
kw = dict(use_sampling=True, top_k=250, cfg_scale=6.0, seed=3)
toks = {}
with torch.cuda.stream(torch.cuda.Stream(dev)):
    for rep in range(3):
        for f2 in (0, 8):
            L.lib().vaura_set_debug_flags2(f2)
            eng._free_graph()
            toks[f2] = eng.generate_codes(feats, 220, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.generate_codes(feats, 220, **kw)
            torch.cuda.synchronize()
            print(f"rep {rep} flags2 {f2} ({'one workgroup per tile' if f2 else 'row-split pair instances'}): {1e3 * (time.perf_counter() - t0) / 3:.2f} ms per 228-step loop, {B} clips ({eng.rows} rows)")
    eng.check_status()
L.lib().vaura_set_debug_flags2(0)
print("token agreement between the two forms:", float((toks[0] == toks[8]).float().mean()))
