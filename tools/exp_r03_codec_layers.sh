#!/bin/bash
# per-dispatch durations of one codec decode (kernel trace of tools/mfma_driver codec 8): which layers hold the time
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/codec_layers; mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_driver.cpp -o /tmp/mfma_driver -ldl || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o c -- /tmp/mfma_driver $GRAFT_REPO_ROOT/vaura_amd/csrc/libvaura_hip.so codec 8 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = open("$OUT/dispatches.txt", "w")
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.write(f'{r["Kernel_Name"][:60]:60s} grid {r["Grid_Size_X"]:>8s} {r["Grid_Size_Y"]:>5s} {r["Grid_Size_Z"]:>4s} wg {r["Workgroup_Size_X"]:>4s} {d:9.1f} us\n')
out.close()
PY
rm -rf $OUT/trace; tail -45 $OUT/dispatches.txt
