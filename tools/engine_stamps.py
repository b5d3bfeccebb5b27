#!/usr/bin/env python
"""Timeline of the one-launch MLP / layer-tail kernels (csrc/mlp_engine.h) from the diagnostic build's in-kernel stamps
(tools/pmc_driver <libvaura_hip_stamps.so> --stamps out.bin with PMC_STAMP_ALL_WAVES=1): per stamp, when the first / median /
last WAVE of the launch reached it, relative to the launch's first wave start (us; medians over launches), wave 0 and waves 1..7
separately.

    python tools/engine_stamps.py gpurun_out/r04/stamps_mlp_h2.bin [kind]      kind 11 = mlp_engine_kernel, 12 = tail_engine_kernel
"""
import sys

import numpy as np

NAMES = {11: ["start", "p1 requested", "p1 products done", "w0: published | w1-7: w2 requested", "hand-off passed", "weights+planes landed",
              "done", "flushed"],
         12: ["start", "p0 products done", "w0: p0 published", "hand-off 0 passed", "p1 products done", "w0: p1 published",
              "hand-off 1 passed", "done"]}


def main():
    rec = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
    kind = (rec[:, 0] & np.uint64(0xFF)).astype(np.int64)
    want = int(sys.argv[2]) if len(sys.argv) > 2 else (11 if (kind == 11).any() else 12)
    blk = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
    wave = ((rec[:, 0] >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64)
    t = rec[:, 1:9].astype(np.int64)
    sel = kind == want
    blk, wave, t = blk[sel], wave[sel], t[sel]
    order = np.argsort(t[:, 0], kind="stable")
    blk, wave, t = blk[order], wave[order], t[order]
    # launches: gaps of > 3 us between consecutive wave starts
    cuts = [0] + [i for i in range(1, len(t)) if t[i, 0] - t[i - 1, 0] > 300] + [len(t)]
    rows = {}
    spans = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b - a < 256:
            continue
        tt, ww, bb = t[a:b] - t[a:b, 0].min(), wave[a:b], blk[a:b]
        spans.append(tt[:, 7].max())
        for grp, m in (("wave 0", ww == 0), ("waves 1-7", ww != 0), ("wave 0, blocks >= 192", (ww == 0) & (bb >= 192))):
            if not m.any():
                continue
            for i in range(8):
                v = tt[m, i]
                v = v[v > 0] if i else v
                if len(v):
                    rows.setdefault((grp, i), []).append((v.min(), np.median(v), v.max()))
    print(f"kind {want}: {len(spans)} launches, median launch span (first wave start -> last record flushed) {np.median(spans) * 0.01:.2f} us")
    for grp in ("wave 0", "waves 1-7", "wave 0, blocks >= 192"):
        print(f"-- {grp}: first / median / last wave to reach each stamp (us from the launch's first wave start)")
        for i in range(8):
            if (grp, i) in rows:
                a = np.median(np.array(rows[(grp, i)]), axis=0) * 0.01
                print(f"   t{i} {NAMES[want][i]:38s} {a[0]:7.2f} {a[1]:7.2f} {a[2]:7.2f}")


if __name__ == "__main__":
    main()
