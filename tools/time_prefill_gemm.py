"""Time the four GEMMs of one teacher-forced prompt pass (sliding-window caller's later chunk: 166 positions x 16 rows) at the
C ABI (`vaura_gemv_pair`, preallocated buffers, HIP events), per weight storage.  TF-equiv = 2 M N K / t (the flops of the
fp32 product the fp16-pair arithmetic reproduces; the MFMA work is 2x that with one weight plane, 3x with two).

    python tools/time_prefill_gemm.py [rows] [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import _lib as L, ops  # noqa: E402

dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 166 * 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
FLAGS = int(os.environ.get("VAURA_DEBUG_FLAGS", "0"))
L.lib().vaura_set_debug_flags(FLAGS)
D, F = 1536, 4096
SHAPES = [("qkv", 3 * D, D, L.EPI_STORE, True), ("wo", D, D, L.EPI_RESID, False), ("w13", 2 * F, D, L.EPI_SWIGLU, True),
          ("w2", D, F, L.EPI_RESID, False)]
rp = (rows + 15) // 16 * 16
g = torch.Generator().manual_seed(0)
print(f"rows {rows} ({rp // 16} row blocks), debug flags {FLAGS}")
total = {}
for wname, wd in (("h1", L.W_H1), ("h2", L.W_H2), ("fp8", L.W_FP8)):
    tot = 0.0
    for name, N, K, epi, norm in SHAPES:
        w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
        wp = ops.pack_weight(w, wd)
        x = torch.randn(rows, K, generator=g).to(dev)
        gain = (torch.rand(K, generator=g) + 0.5).to(dev)
        xs, ss = ops.split_rows(ops.pack_rows(x), rows, K, gain if norm else None, want_ss=norm)
        n_out = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.zeros(rp * n_out, dtype=torch.float32, device=dev)
        osp = torch.zeros(rp * 2 * n_out, dtype=torch.int16, device=dev)
        oss = torch.zeros((rp // 16) * (n_out // 16) * 16, dtype=torch.float32, device=dev)
        res = torch.randn(rp * n_out, generator=g).to(dev) if epi == L.EPI_RESID else None
        gout = torch.ones(n_out, device=dev)
        st = L.current_stream()

        def run():
            L.check(L.lib().vaura_gemv_pair(L.ptr(wp), wd, L.ptr(xs), L.ptr(ss) if norm else None, K // 16 if norm else 0, L.ptr(res),
                                            L.ptr(out) if epi != L.EPI_SWIGLU else None, None, L.ptr(osp), L.ptr(gout),
                                            L.ptr(oss) if epi == L.EPI_RESID else None, rows, N, K, epi, 1e-5, st), "vaura_gemv_pair")

        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot += ms
        print(f"  {wname:3s} {name:4s} N={N:5d} K={K:4d}: {ms * 1e3:8.1f} us  {2.0 * rows * N * K / ms / 1e9:7.1f} TF-equiv")
    total[wname] = tot
    flops = sum(2.0 * rows * N * K for _, N, K, _, _ in SHAPES)
    print(f"  {wname}: layer {tot * 1e3:.1f} us -> 24 layers {tot * 24:.2f} ms, {flops / tot / 1e9:.1f} TF-equiv")
