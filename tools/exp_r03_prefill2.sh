#!/bin/bash
# later chunk of the sliding-window caller: LDS-DMA GEMM (flags 0) against the register-staged one (131072), both storages, same box
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/prefill2; mkdir -p $OUT
for w in h1 h2; do for f in 0 131072 0 131072; do
  VAURA_DEBUG_FLAGS=$f VAURA_WEIGHTS=$w VAURA_PREFILL_PASSES=192 timeout 600 python tools/time_sliding_window.py > $OUT/sw_${w}_$f.log 2>&1; echo "$w flags $f: $(grep 'ms per chunk' $OUT/sw_${w}_$f.log)"
done; done
