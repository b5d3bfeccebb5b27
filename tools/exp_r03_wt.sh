#!/bin/bash
# write-through stores (product) against ordinary stores (libvaura_hip_plain.so), same box, alternating; then the GPU suite
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/wt; mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread || exit 1
for w in h2 h1; do for rep in 1 2; do for lib in libvaura_hip.so libvaura_hip_plain.so; do
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/$lib --time 5 --weights $w > $OUT/t.log 2>&1; echo "$w $lib: $(grep 'loop of 228' $OUT/t.log | cut -c1-110)"
done; done; done
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
