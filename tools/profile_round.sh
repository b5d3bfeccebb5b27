#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root.  Usage: tools/profile_round.sh r02
#   1. rocprofv3 kernel stats of the bench command itself (python) -> gpurun_out/prof_<tag>/stats
#   2. the same kernels through the PRODUCT library driven by a plain C++ program (tools/pmc_driver.cpp dlopen()s
#      vaura_amd/csrc/libvaura_hip.so and calls vaura_decode_step): kernel stats + separate --pmc passes for FETCH_SIZE and
#      WRITE_SIZE (rocprofv3 --pmc crashes at start-up under python on this image; the counters need their own passes,
#      MI355X_MICROARCH.md §rocprofv3 PMC slots).  The driver is rebuilt every time: no stale binary.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
LIB=$ROOT/vaura_amd/csrc/libvaura_hip.so
DRV=$ROOT/tools/pmc_driver
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 $ROOT/tools/pmc_driver.cpp -o $DRV -ldl || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-plugin > $OUT/bench_stats.log 2>&1
tail -n 1 $OUT/bench_stats.log | cut -c1-300
for W in bf16 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_stats_$W -- $DRV $LIB --weights $W --steps 24 --pos0 100 > $OUT/drv_stats_$W.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/drv_${C}_$W -- $DRV $LIB --weights $W --steps 24 --pos0 100 > $OUT/drv_${C}_$W.log 2>&1
  done
  tail -n 1 $OUT/drv_stats_$W.log
done
find $OUT -name "*.csv" | head -30
