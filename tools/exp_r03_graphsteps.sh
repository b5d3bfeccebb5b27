#!/bin/bash
# decode steps per graph launch (VAURA_GRAPH_STEPS): loop time of the C++ driver, alternating
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread || exit 1
for w in h2 h1; do for k in 1 4 12 1 4 12; do
  echo "$w steps/graph $k: $(VAURA_GRAPH_STEPS=$k timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --weights $w 2>&1 | grep 'loop of 228' | cut -c1-100)"
done; done
