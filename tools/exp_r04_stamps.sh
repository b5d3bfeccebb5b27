#!/bin/bash
# in-kernel stamps of the one-launch MLP (flags 0) and the layer-tail experiment (flags 8), every wave recording
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread || exit 1
[ -n "$ALLWAVES" ] && export PMC_STAMP_ALL_WAVES=1
for w in h2 h1; do for f in ${STAMP_FLAGS:-0 8}; do
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip_stamps.so --stamps gpurun_out/r04/stamps_${w}_f${f}.bin --flags $f --weights $w --steps 2 --pos0 100
  python tools/engine_stamps.py gpurun_out/r04/stamps_${w}_f${f}.bin > gpurun_out/r04/stamps_${w}_f${f}.txt 2>&1
  cat gpurun_out/r04/stamps_${w}_f${f}.txt
  rm -f gpurun_out/r04/stamps_${w}_f${f}.bin
done; done
