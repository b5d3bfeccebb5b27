#!/bin/bash
# per-kernel averages of the extractor forward (mfma_driver avclip 8 under rocprofv3 --stats); linear_pair_kernel is the box-speed reference
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/avclip_stats; rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_driver.cpp -o /tmp/mfma_driver -ldl || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- /tmp/mfma_driver $GRAFT_REPO_ROOT/vaura_amd/csrc/libvaura_hip.so avclip 8 > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f'{r["Name"][:60]:60s} {r["Calls"]:>5s} {float(r["AverageNs"])/1e3:9.1f} us {r["Percentage"]:>6s} %')
PY
rm -rf $OUT/t
