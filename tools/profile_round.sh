#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel stats of the bench command.
# (The PMC passes for HBM traffic are in tools/profile_pmc_microbench.sh: rocprofv3 --pmc crashes at start-up
# under the python bench on this image.)
# Usage: tools/profile_round.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
tail -n 1 $OUT/bench_stats.log
find $OUT -name "*.csv" | head -20
