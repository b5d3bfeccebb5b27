#!/bin/bash
# A/B of the one-launch MLP (csrc/mlp_engine.h, debug flag 4) against the two-launch path on the product library:
# interleaved rounds of the 228-step loop in one process per storage, then per-stage averages
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread || exit 1
for w in h2 h1; do
  echo "== weights $w" 
  timeout 600 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time ${ROUNDS:-5} --flags ${FLAGS:-0,4} --weights $w 2>&1 | tail -12
done
