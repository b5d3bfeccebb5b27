cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generate.py -x -q -m gpu -k "one_launch_mlp_is_bit or fp8 or configs4 or two_row" > gpurun_out/t7.log 2>&1
echo "pytest rc $?" >> gpurun_out/t7.log
tail -6 gpurun_out/t7.log
cat > /tmp/ab_fp8e.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from vaura_amd import _lib as L, synth
from vaura_amd.engine import DecoderEngine
cfg = synth.FULL_SAMPLER
eng = DecoderEngine(cfg, synth.sampler_state_dict(cfg, seed=0, round_bf16=False), "cuda:0", wdtype="fp8")
feats = synth.video_features(16, seed=0).to("cuda:0")
kw = dict(use_sampling=True, top_k=250, cfg_scale=6.0, seed=3)
toks = {}
with torch.cuda.stream(torch.cuda.Stream("cuda:0")):
    for rep in range(3):
        for f2 in (0, 16):
            L.lib().vaura_set_debug_flags2(f2); eng._free_graph()
            toks[f2] = eng.generate_codes(feats, 220, **kw); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): eng.generate_codes(feats, 220, **kw)
            torch.cuda.synchronize()
            print(f"rep {rep} flags2 {f2} ({'separate launches' if f2 else 'one-launch MLP on fp8 tile pairs'}): {1e3 * (time.perf_counter() - t0) / 3:.2f} ms per 228-step loop, 16 clips (32 rows)")
    eng.check_status()
print("tokens identical:", bool(torch.equal(toks[0], toks[16])))
PY
timeout 600 python /tmp/ab_fp8e.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_fp8_engine.log
