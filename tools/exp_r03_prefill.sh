#!/bin/bash
# prefill GEMM variants (debug flags: 0x4000 round-2 tile rule, 0x8000 no XCD remap, 0x10000 one weight k-group in flight, 32 64-row tiles)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/prefill; mkdir -p $OUT
for f in ${FLAGSETS:-0 114688 32768 65536 16384 32}; do
  VAURA_DEBUG_FLAGS=$f timeout 600 python tools/time_prefill_gemm.py > $OUT/gemm_$f.log 2>&1; grep -v "amdgpu.ids" $OUT/gemm_$f.log
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generate.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
VAURA_PREFILL_PASSES=192 timeout 600 python tools/time_sliding_window.py > $OUT/sw.log 2>&1; tail -2 $OUT/sw.log
