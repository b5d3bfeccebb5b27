"""Row f2 timing: Segment-AVCLIP features for one configs[1] batch (8 clips x 4 segments of 16 x 224 x 224 frames).
    python tools/time_avclip.py [n_clips]      (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import AvclipEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
FLAGS = int(os.environ.get("VAURA_DEBUG_FLAGS", "0"))    # A/B: 64 = linears on the 128 x 96 conv tile, 128 = one-thread-per-query space attention
FLAGS2 = int(os.environ.get("VAURA_DEBUG_FLAGS2", "0"))  # second flag word: 32 = linears on one LDS stage (two barriers per k-step)
if FLAGS or FLAGS2:
    from vaura_amd import _lib as L  # noqa: E402
    L.lib().vaura_set_debug_flags(FLAGS)
    L.lib().vaura_set_debug_flags2(FLAGS2)
eng = AvclipEngine(synth.FULL_AVCLIP, synth.avclip_state_dict(seed=0), dev)
frames = torch.randn(B, 4, 3, 16, 224, 224, device=dev)
for _ in range(2):
    eng.forward(frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    out = eng.forward(frames)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
c = synth.FULL_AVCLIP
L = 1 + c.t * c.n
D, H = c.embed_dim, c.embed_dim * c.mlp_ratio
flop_seg = 2 * (c.t * c.n * 1536 * D + c.depth * L * (2 * (3 * D * D + D * D) + 2 * D * H))      # linears only
print(f"avclip[flags {FLAGS}:{FLAGS2}]: {B} clips x 4 segments: {dt * 1e3:.2f} ms per batch = {B * 4 / dt:.1f} segments/s; linear layers "
      f"{B * 4 * flop_seg / dt / 1e12:.1f} TFLOP/s-equivalent (x3 raw fp16 MFMA); out {tuple(out.shape)}")
