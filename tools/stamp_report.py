"""Per-phase decomposition of the decode-step stages from the in-kernel s_memrealtime stamps of the DIAGNOSTIC build
(csrc/common.h VAURA_STAMPS; collected by `tools/pmc_driver <libvaura_hip_stamps.so> --stamps <file>`).

    python tools/stamp_report.py gpurun_out/r03/stamps_bf16.bin [out.json]

A record = one WAVE: {kind | block << 8 | xcc << 40 | wave << 48, t0..t6, t7} (128 bytes) in 10 ns ticks of the chip-wide 100 MHz
counter; t7 = after the record slot's atomic returned (the flush's own round trip, excluded from the gap below).
Launches do not overlap (in-order stream), so sorting by t0 and cutting where the kind changes recovers them.  Per launch:
  ramp      first wave start -> last wave start of the grid
  issue     t1 - t0   wave start -> every request of the first batch issued
  weights   t2 - t1   -> weight tiles (HBM) landed            [attention: the new q/k/v + rope entry, written by the previous kernel]
  operands  t3 - t2   -> activation planes / partials landed  [attention: first barrier, rotated q/k/v parked in LDS]
  compute   t4 - t3   MFMAs + LDS write                       [attention: every cached K/V row landed]
  barrier   t5 - t4   workgroup barrier                       [attention: scores, softmax, P.V, second barrier]
  epilogue  t6 - t5   reduction over waves, epilogue, stores acknowledged
  skew      last wave's t4 - first wave's t4 inside a workgroup (how long the fastest wave waits at the reduction barrier)
  span      last t6 - first t0 of the launch;  gap = next launch's first t0 - this launch's last t7 (store-ack of the record
            and the kernel boundary; the flush's atomic round trip is not in it)
The stamps drain the memory pipeline where they wait: read SHARES, not the absolute span (the product kernels overlap more)."""
import json
import sys

import numpy as np

KINDS = {1: "qkv", 2: "attn", 3: "wo", 4: "w13", 5: "w2", 6: "heads", 9: "other"}
PHASES = ["issue", "weights", "operands", "compute", "barrier", "epilogue"]


def main():
    path = sys.argv[1]
    rec = np.fromfile(path, dtype=np.uint64).reshape(-1, 16)
    kind = (rec[:, 0] & 0xFF).astype(np.int64)
    xcc = ((rec[:, 0] >> np.uint64(40)) & np.uint64(15)).astype(np.int64)
    blk = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
    t = rec[:, 1:9].astype(np.int64)
    order = np.argsort(t[:, 0], kind="stable")
    kind, xcc, t, blk = kind[order], xcc[order], t[order], blk[order]
    cuts = [0] + [i for i in range(1, len(kind)) if kind[i] != kind[i - 1]] + [len(kind)]
    launches = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        tt = t[a:b]
        bb = blk[a:b]
        ub = np.unique(bb)
        skew = np.median([tt[bb == u, 4].max() - tt[bb == u, 4].min() for u in ub[:64]])
        launches.append({"kind": int(kind[a]), "n": len(ub), "first": int(tt[:, 0].min()), "last_start": int(tt[:, 0].max()),
                         "end": int(tt[:, 6].max()), "end7": int(tt[:, 7].max()), "phases": np.median(np.diff(tt[:, :7], axis=1), axis=0),
                         "skew": float(skew), "wg_span": float(np.median(tt[:, 6] - tt[:, 0])), "xccs": len(set(xcc[a:b].tolist()))})
    out = {"source": path, "tick_ns": 10, "records": int(len(kind)), "launches": len(launches), "per_kind": {}}
    print(f"{len(kind)} records, {len(launches)} launches")
    print(f"{'kind':6s} {'n':>4s} {'WGs':>5s} {'ramp':>6s} " + " ".join(f"{p:>9s}" for p in PHASES) + f" {'skew':>6s} {'wv span':>8s} {'span':>7s} {'gap>':>6s}   (us, medians over waves / launches)")
    for k, name in KINDS.items():
        idx = [i for i, l in enumerate(launches) if l["kind"] == k]
        if not idx:
            continue
        ls = [launches[i] for i in idx]
        ph = np.median(np.stack([l["phases"] for l in ls]), axis=0) * 0.01
        ramp = float(np.median([l["last_start"] - l["first"] for l in ls])) * 0.01
        span = float(np.median([l["end"] - l["first"] for l in ls])) * 0.01
        wgs = float(np.median([l["wg_span"] for l in ls])) * 0.01
        gaps = [launches[i + 1]["first"] - launches[i]["end7"] for i in idx if i + 1 < len(launches)
                and launches[i + 1]["first"] - launches[i]["end7"] < 2000]        # < 20 us: skip the unstamped embed / sampler
        gap = float(np.median(gaps)) * 0.01 if gaps else None
        skew = float(np.median([l["skew"] for l in ls])) * 0.01
        out["per_kind"][name] = {"launches": len(ls), "workgroups": int(np.median([l["n"] for l in ls])), "ramp_us": round(ramp, 2),
                                 **{f"{p}_us": round(float(v), 2) for p, v in zip(PHASES, ph)}, "wave_skew_at_barrier_us": round(skew, 2),
                                 "wave_span_us": round(wgs, 2),
                                 "launch_span_us": round(span, 2), "gap_to_next_us": None if gap is None else round(gap, 2)}
        print(f"{name:6s} {len(ls):4d} {int(np.median([l['n'] for l in ls])):5d} {ramp:6.2f} " + " ".join(f"{v:9.2f}" for v in ph)
              + f" {skew:6.2f} {wgs:8.2f} {span:7.2f} " + (f"{gap:6.2f}" if gap is not None else "   n/a"))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
