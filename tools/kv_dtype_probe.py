"""K/V cache types at configs[4]s shape (16 clips, cfg 6, 32 rows, full depth): loop time and teacher-forced logits against the fp32 cache.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from vaura_amd import synth
from vaura_amd.engine import DecoderEngine
cfg=synth.FULL_SAMPLER; sd=synth.sampler_state_dict(cfg, seed=0, round_bf16=False)
f=synth.video_features(16, seed=0).cuda()
kw=dict(use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, seed=1234)
s=torch.cuda.Stream()
idx=torch.randint(0,1024,(4,9,32)).cuda()
ref=None
for wd,kv in (("fp8h","f32"),("fp8h","f16"),("fp8h","f8"),("h2","f32")):
    e=DecoderEngine(cfg, sd, "cuda:0", wdtype=wd, kv_dtype=kv)
    with torch.cuda.stream(s):
        e.generate_codes(f,220,**kw); torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(3): e.generate_codes(f,220,**kw)
        torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/3*1e3
    e.check_status()
    lg=e.logits_all_positions(idx, f[:4]).float().cpu()
    if ref is None: ref=lg
    print(wd, kv, "loop ms", round(dt,2), "logits rel-rms vs fp8h/f32-KV", float((lg-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()), flush=True)
    del e; torch.cuda.empty_cache()
