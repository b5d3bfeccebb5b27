#!/bin/bash
# HBM-traffic counters for the decode GEMV kernels, collected on the stand-alone harness
# (tools/microbench/gemv_bench runs the product's own kernel templates on cycling weight sets):
# rocprofv3 --pmc segfaults at start-up when the profiled program is the python bench on this image
# (gpurun_out/prof_*/bench_fetch.log), while it works for a plain HIP binary.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$ROOT/tools/microbench/gemv_bench
[ -x $B ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I$ROOT/include -I$ROOT/vaura_amd/csrc $B.hip -o $B
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/mb_$C -- $B "g3 " > $OUT/mb_$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mb_stats -- $B "g3 " > $OUT/mb_stats.log 2>&1
tail -n 8 $OUT/mb_stats.log
