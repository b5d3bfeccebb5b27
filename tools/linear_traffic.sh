#!/bin/bash
# Fabric traffic of row f2's linear layers per launch SHAPE: FETCH_SIZE / WRITE_SIZE (separate --pmc passes) and durations of
# `mfma_driver <lib> avclip 8`, bucketed by kernel and grid size.   gpurun -- 'bash tools/linear_traffic.sh [lib-tag ..]'
cd ${GRAFT_REPO_ROOT:-.}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/linear_traffic; mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_driver.cpp -o /tmp/mfma_driver -ldl || exit 1
cd /tmp && export TMPDIR=/tmp
for t in "" "$@"; do
  LIB=$ROOT/vaura_amd/csrc/libvaura_hip$t.so
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p_${C}$t -- /tmp/mfma_driver $LIB avclip 8 1 1 > $OUT/p_${C}$t.log 2>&1
  done
  rocprofv3 --kernel-trace --output-format csv -d $OUT/k$t -- /tmp/mfma_driver $LIB avclip 8 1 1 > $OUT/k$t.log 2>&1
  python3 - "$OUT" "$t" <<'PY'
import csv, glob, sys, collections
out, t = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/p_{C}{t}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "linear" not in r["Kernel_Name"]: continue
            key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if C == "FETCH_SIZE": n[key] += 1
dur = collections.defaultdict(float); nd = collections.Counter()
for f in glob.glob(f"{out}/k{t}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; nd[key] += 1
print(f"== lib{t}")
tot = 0
for key in sorted(agg, key=lambda k: -dur.get(k, 0)):
    c = n[key]; fe = agg[key]["FETCH_SIZE"] / c * 1024 * 2; wr = agg[key]["WRITE_SIZE"] / c * 1024      # guide: FETCH_SIZE x2 on gfx950
    d = dur[key] / max(nd[key], 1); tot += dur[key]
    if d == 0: continue
    print(f"{key[0]:42s} grid {key[1]:>9s} x{c:3d}: fetch {fe / 1e6:8.1f} MB  write {wr / 1e6:8.1f} MB  {d:8.1f} us  -> {(fe + wr) / d / 1e6:6.2f} TB/s")
print(f"linears total {tot / 1e3:.2f} ms")
PY
done 2>&1 | tee $OUT/summary.txt
