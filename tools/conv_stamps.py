#!/usr/bin/env python
"""Timeline of the codec's conv kernels from the diagnostic build's in-kernel stamps (csrc/dac.hip VA_STAMP; collected by
`MFMA_STAMPS=out.bin tools/mfma_driver <libvaura_hip_stamps.so> codec 8`): per launch (kind, grid), how long wave 0 of a
workgroup spends between consecutive stamps (us, median / p90 over the launch's workgroups) and its whole lifetime.

    kind 20 conv_pair_kernel            0 start | 1 first tiles staged | 2 main loop done | 4 first 128-row pass staged | 5 ... stored | 6 end
    kind 21 conv_pair_kernel<.., FUSE>  0 | 1 | 2 | 3 pass 0: Snake image in LDS | 4 pass 0: 1 x 1 products | 5 pass 0: stored | 6 end (4 passes)
    kind 22 conv_unit_kernel            0 | 1 | 2 | 3 Snake in registers | 4 column tile 0: 1 x 1 products | 5 column tile 0: stored | 6 end
"""
import sys

import numpy as np

rec = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
kind = (rec[:, 0] & np.uint64(0xFF)).astype(np.int64)
wave = ((rec[:, 0] >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64)
t = rec[:, 1:9].astype(np.int64)
order = np.argsort(t[:, 0], kind="stable")
kind, wave, t = kind[order], wave[order], t[order]
# launches: a change of kind, or > 20 us without a wave start
cuts = [0] + [i for i in range(1, len(t)) if kind[i] != kind[i - 1] or t[i, 0] - t[i - 1, 0] > 2000] + [len(t)]
print("kind   WGs   span   life(med/p90) | " + "  ".join(f"{a}->{b}" for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6))) + "   (us; medians over workgroups, wave 0)")
for a, b in zip(cuts[:-1], cuts[1:]):
    m = wave[a:b] == 0
    if m.sum() < 64:
        continue
    tt = t[a:b][m]
    k = int(kind[a])
    span = (tt[:, 6].max() - tt[:, 0].min()) * 0.01
    life = (tt[:, 6] - tt[:, 0]) * 0.01
    segs = []
    prev = 0
    for i in range(1, 7):
        if (tt[:, i] == 0).all():
            segs.append("    -")
            continue
        d = (tt[:, i] - tt[:, prev]) * 0.01
        segs.append(f"{np.median(d):5.1f}")
        prev = i
    print(f"{k:4d} {m.sum():5d} {span:6.1f}   {np.median(life):5.1f} / {np.percentile(life, 90):5.1f}    | " + "  ".join(segs))
