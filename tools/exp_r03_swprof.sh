#!/bin/bash
# kernel-level profile of one later chunk of the sliding-window caller (prompt pass + 63 decode steps), per debug-flag set
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/swprof; mkdir -p $OUT
export VAURA_PREFILL_PASSES=192
cd /tmp && export TMPDIR=/tmp
for f in ${FLAGSETS:-0 131072}; do
  export VAURA_DEBUG_FLAGS=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$f -o sw -- python3 $GRAFT_REPO_ROOT/tools/time_sliding_window.py > $OUT/sw_$f.log 2>&1
  grep "ms per chunk" $OUT/sw_$f.log
  cp $(find $OUT/trace_$f -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$f.csv; grep "gemm\|prefill\|rope_append" $OUT/kernel_stats_$f.csv | cut -c1-60,100-220
  rm -rf $OUT/trace_$f
done
