#!/usr/bin/env python
"""Which workgroups publish late?  Per workgroup id (and per blockIdx % 8 = XCD under round-robin placement) the mean time of wave 0's stamp `idx`
(default 3 = published) over the launches of a stamps file (tools/pmc_driver <libvaura_hip_stamps.so> --stamps out.bin), relative to the launch's first wave.

    python tools/engine_stamps_by_block.py out.bin [idx]
"""
import sys

import numpy as np

rec = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = (rec[:, 0] & np.uint64(0xFF)).astype(np.int64)
blk = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
wave = ((rec[:, 0] >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64)
t = rec[:, 1:9].astype(np.int64)
sel = kind == 11
blk, wave, t = blk[sel], wave[sel], t[sel]
order = np.argsort(t[:, 0], kind="stable")
blk, wave, t = blk[order], wave[order], t[order]
cuts = [0] + [i for i in range(1, len(t)) if t[i, 0] - t[i - 1, 0] > 300] + [len(t)]
acc, n, late = np.zeros(256), np.zeros(256), np.zeros(256)
for a, b in zip(cuts[:-1], cuts[1:]):
    if b - a < 256:
        continue
    tt, ww, bb = t[a:b] - t[a:b, 0].min(), wave[a:b], blk[a:b]
    m = (ww == 0) & (tt[:, idx] > 0)
    v, ids = tt[m, idx] * 0.01, bb[m]
    acc[ids] += v
    n[ids] += 1
    thr = np.sort(v)[-16] if len(v) >= 16 else v.max()
    late[ids[v >= thr]] += 1
mean = acc / np.maximum(n, 1)
print(f"stamp {idx}: mean over {int(n.max())} launches; all workgroups: median {np.median(mean[n > 0]):.2f} us, max {mean.max():.2f}")
print("by blockIdx % 8:", " ".join(f"{mean[(np.arange(256) % 8 == x) & (n > 0)].mean():.2f}" for x in range(8)))
print("by blockIdx // 32:", " ".join(f"{mean[(np.arange(256) // 32 == x) & (n > 0)].mean():.2f}" for x in range(8)))
worst = np.argsort(-late)[:24]
print("most often among the 16 latest:", " ".join(f"{int(w)}({int(late[w])})" for w in worst))
